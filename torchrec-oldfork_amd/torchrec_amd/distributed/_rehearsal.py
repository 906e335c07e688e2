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
