"""Rehearsal plumbing: N ranks of the product path on FEWER GPUs than ranks (a one-GPU box), over the `gloo` backend.

RCCL refuses two ranks on one device, and gloo has no device all-to-all.  `stage_all_to_all_through_host()` replaces
`torch.distributed.all_to_all_single` for device tensors by: finish the current stream, copy to the host, gloo's
host all-to-all, copy back.  Everything else a rank executes is the product path: the HIP lookup / exchange kernels,
HIP graphs, the explicit step, the flat-gradient all-reduce (gloo reduces device tensors itself).  Used by
`bench.py` with TORCHREC_AMD_BENCH_BACKEND=gloo and by tests/test_multirank_gpu.py; never on a real multi-GPU run
(`nccl` = RCCL there, untouched).  The timings of such a run are NOT multi-GPU timings: all ranks share one GPU's
CUs and HBM and the exchange crosses the host."""
import os

import torch
import torch.distributed as dist
from fbgemm_gpu._streams import side_stream

_real_all_to_all_single = None


class _Done:
    def wait(self):
        return True

    def is_completed(self):
        return True


def stage_all_to_all_through_host() -> None:
    global _real_all_to_all_single
    if _real_all_to_all_single is not None:
        return
    real = _real_all_to_all_single = dist.all_to_all_single

    def all_to_all_single(output, input, output_split_sizes=None, input_split_sizes=None, group=None, async_op=False):
        if not input.is_cuda:
            return real(output, input, output_split_sizes, input_split_sizes, group=group, async_op=async_op)
        torch.cuda.current_stream(input.device).synchronize()
        host_out = torch.empty(output.shape, dtype=output.dtype)
        real(host_out, input.cpu().contiguous(), output_split_sizes, input_split_sizes, group=group)
        output.copy_(host_out)
        return _Done() if async_op else None

    dist.all_to_all_single = all_to_all_single
    real_list = dist.all_to_all

    def all_to_all(output_tensor_list, input_tensor_list, group=None, async_op=False):
        """The list form (pieces that are views, e.g. the half-batch exchange) through ONE host all_to_all_single."""
        if not input_tensor_list:
            return real_list(output_tensor_list, input_tensor_list, group=group, async_op=async_op)
        if input_tensor_list[0].is_cuda:  # (host tensors take the same route: gloo has no list-form all-to-all at all)
            torch.cuda.current_stream(input_tensor_list[0].device).synchronize()
        dtype = input_tensor_list[0].dtype
        host_in = torch.cat([t.reshape(-1).cpu() for t in input_tensor_list])
        host_out = torch.empty(sum(t.numel() for t in output_tensor_list), dtype=dtype)
        real(host_out, host_in, [t.numel() for t in output_tensor_list], [t.numel() for t in input_tensor_list], group=group)
        for t, piece in zip(output_tensor_list, host_out.split([t.numel() for t in output_tensor_list])):
            t.copy_(piece.view(t.shape))
        return _Done() if async_op else None

    dist.all_to_all = all_to_all


_link_state = {}


def set_link_full_bytes(nbytes: int) -> None:
    """The message size the emulated link time stands for (see emulate_link_time)."""
    _link_state["full_bytes"] = int(nbytes)


def emulate_link_time(fwd_us: float, full_bytes: int = 0) -> None:
    """One-rank rehearsal: make every device `all_to_all_single` take at least `fwd_us` microseconds on the GPU timeline,
    as a transfer over xGMI would (the one-rank RCCL all-to-all is a local copy at HBM speed: 109 MB in 44 us, where 7
    links move a rank's 25 MB per peer in ~250 us).  A spin kernel (`torch.cuda._sleep`, one wavefront) runs on a
    dedicated stream in front of the collective, which is issued from that stream: the collective's own stream then waits
    for the spin, consumers wait for the collective as always, the compute stream is never touched.  Calibrated once.
    `full_bytes`: the message size `fwd_us` stands for; a smaller message (one half of a half-batch exchange) spins
    proportionally shorter.  0 = every message takes `fwd_us`.
    Measurement plumbing for bench.py (TORCHREC_AMD_REHEARSAL_LINK_US); never on a real multi-GPU run."""
    global _real_all_to_all_single
    if fwd_us <= 0 or "stream" in _link_state:
        return
    dev = torch.device("cuda", torch.cuda.current_device())
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(1000)
    torch.cuda.synchronize()
    a.record()
    torch.cuda._sleep(10_000_000)
    b.record()
    b.synchronize()
    cycles_per_us = 10_000_000 / (a.elapsed_time(b) * 1e3)
    _link_state.update(full_bytes=int(full_bytes or _link_state.get("full_bytes", 0)), stream=side_stream(dev), cycles=int(fwd_us * cycles_per_us), us=fwd_us)
    real = dist.all_to_all_single

    def all_to_all_single(output, input, output_split_sizes=None, input_split_sizes=None, group=None, async_op=False):
        if not input.is_cuda or input.numel() * input.element_size() < (1 << 20):  # ids: small, latency only
            return real(output, input, output_split_sizes, input_split_sizes, group=group, async_op=async_op)
        return behind_link(lambda: real(output, input, output_split_sizes, input_split_sizes, group=group, async_op=async_op),
                           [output, input], input.numel() * input.element_size(), async_op)

    def behind_link(issue, tensors, nbytes, async_op):
        link = _link_state["stream"]
        link.wait_stream(torch.cuda.current_stream())
        fb = _link_state.get("full_bytes", 0)
        frac = min(1.0, nbytes / fb) if fb > 0 else 1.0
        with torch.cuda.stream(link):
            torch.cuda._sleep(max(int(_link_state["cycles"] * frac), 1))
            work = issue()
        for t in tensors:
            t.record_stream(link)
        if not async_op:
            torch.cuda.current_stream().wait_stream(link)
        return work

    real_list = dist.all_to_all

    def all_to_all(output_tensor_list, input_tensor_list, group=None, async_op=False):
        nbytes = sum(t.numel() * t.element_size() for t in input_tensor_list)
        if not input_tensor_list or not input_tensor_list[0].is_cuda or nbytes < (1 << 20):
            return real_list(output_tensor_list, input_tensor_list, group=group, async_op=async_op)
        return behind_link(lambda: real_list(output_tensor_list, input_tensor_list, group=group, async_op=async_op),
                           list(output_tensor_list) + list(input_tensor_list), nbytes, async_op)

    dist.all_to_all = all_to_all

    # dense all-reduces (TORCHREC_AMD_FORCE_DENSE_REDUCE issues them on one rank): 20 us of latency + bytes at an algorithm
    # bandwidth, on a stream of their own (they use their own communicator in the product: models/dlrm.py)
    ar_gbs = float(os.environ.get("TORCHREC_AMD_REHEARSAL_AR_GBS", "0"))
    if ar_gbs > 0:
        real_ar = dist.all_reduce
        ar_stream = side_stream(dev)
        cyc_us = _link_state["cycles"] / fwd_us

        def all_reduce(tensor, op=dist.ReduceOp.SUM, group=None, async_op=False):
            if not tensor.is_cuda:
                return real_ar(tensor, op=op, group=group, async_op=async_op)
            us = 20.0 + tensor.numel() * tensor.element_size() / (ar_gbs * 1e3)
            ar_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(ar_stream):
                torch.cuda._sleep(max(int(us * cyc_us), 1))
                work = real_ar(tensor, op=op, group=group, async_op=async_op)
            tensor.record_stream(ar_stream)
            if not async_op:
                torch.cuda.current_stream().wait_stream(ar_stream)
            return work

        dist.all_reduce = all_reduce

    dist.all_to_all_single = all_to_all_single
