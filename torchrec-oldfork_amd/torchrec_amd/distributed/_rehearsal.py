"""Rehearsal plumbing: N ranks of the product path on FEWER GPUs than ranks (a one-GPU box), over the `gloo` backend.

RCCL refuses two ranks on one device, and gloo has no device all-to-all.  `stage_all_to_all_through_host()` replaces
`torch.distributed.all_to_all_single` for device tensors by: finish the current stream, copy to the host, gloo's
host all-to-all, copy back.  Everything else a rank executes is the product path: the HIP lookup / exchange kernels,
HIP graphs, the explicit step, the flat-gradient all-reduce (gloo reduces device tensors itself).  Used by
`bench.py` with TORCHREC_AMD_BENCH_BACKEND=gloo and by tests/test_multirank_gpu.py; never on a real multi-GPU run
(`nccl` = RCCL there, untouched).  The timings of such a run are NOT multi-GPU timings: all ranks share one GPU's
CUs and HBM and the exchange crosses the host."""
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


_link_state = {}


def emulate_link_time(fwd_us: float) -> None:
    """One-rank rehearsal: make every device `all_to_all_single` take at least `fwd_us` microseconds on the GPU timeline,
    as a transfer over xGMI would (the one-rank RCCL all-to-all is a local copy at HBM speed: 109 MB in 44 us, where 7
    links move a rank's 25 MB per peer in ~250 us).  A spin kernel (`torch.cuda._sleep`, one wavefront) runs on a
    dedicated stream in front of the collective, which is issued from that stream: the collective's own stream then waits
    for the spin, consumers wait for the collective as always, the compute stream is never touched.  Calibrated once.
    Measurement plumbing for bench.py (TORCHREC_AMD_REHEARSAL_LINK_US); never on a real multi-GPU run."""
    global _real_all_to_all_single
    if fwd_us <= 0 or _link_state:
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
    _link_state.update(stream=side_stream(dev), cycles=int(fwd_us * cycles_per_us), us=fwd_us)
    real = dist.all_to_all_single

    def all_to_all_single(output, input, output_split_sizes=None, input_split_sizes=None, group=None, async_op=False):
        if not input.is_cuda or input.numel() * input.element_size() < (1 << 20):  # ids: small, latency only
            return real(output, input, output_split_sizes, input_split_sizes, group=group, async_op=async_op)
        link = _link_state["stream"]
        link.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(link):
            torch.cuda._sleep(_link_state["cycles"])
            work = real(output, input, output_split_sizes, input_split_sizes, group=group, async_op=async_op)
        for t in (output, input):
            t.record_stream(link)
        if not async_op:
            torch.cuda.current_stream().wait_stream(link)
        return work

    dist.all_to_all_single = all_to_all_single
