"""Development probe: wall-clock (perf_counter) host time per train step spent inside chosen functions of the sparse
path (forward pieces, autograd backward functions on the engine's device thread, optimizer), without cProfile's
overhead.  Usage: python tools/step_timers.py <out.txt> [bench args]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
out = sys.argv[1]
sys.argv = [sys.argv[0]] + sys.argv[2:]
import bench  # noqa: E402

sys.path.insert(0, bench.PKG)
import torch  # noqa: E402
from fbgemm_gpu import split_table_batched_embeddings_ops as tbe  # noqa: E402
from torchrec_amd.distributed import embeddingbag as eb, hip_graph, train_pipeline as tp  # noqa: E402
from torchrec_amd.models import dlrm  # noqa: E402
from torchrec_amd.optim import keyed  # noqa: E402

acc = {}
lock = threading.Lock()


def wrap(owner, name, label=None, static=False):
    fn = getattr(owner, name)
    label = label or f"{getattr(owner, '__name__', owner)}.{name}"

    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            d = time.perf_counter() - t0
            with lock:
                c = acc.setdefault(label, [0, 0.0])
                c[0] += 1
                c[1] += d
    setattr(owner, name, staticmethod(w) if static else w)


base = tbe.SplitTableBatchedEmbeddingBagsCodegen.__mro__[1]
for n in ("_real_layout", "_get_layout", "_pooled_layout", "_forward_impl", "_prepare_backward", "_ensure_pinned"):
    wrap(base, n)
for cls in (tbe._FusedLookupInto, tbe._DenseLookupInto, tbe._FusedLookup, tbe._DenseLookup, eb._ExchangeReq, eb._ExchangeWait,
            hip_graph._Replay, dlrm._FusedDotInteraction):
    wrap(cls, "forward", static=True)
    wrap(cls, "backward", static=True)
for n in ("input_dist", "compute_and_output_dist", "_dp_inputs"):
    wrap(eb.ShardedEmbeddingBagCollection, n)
for n in ("start_forward", "finish_forward", "start_backward", "finish_backward"):
    wrap(eb._ExchangeState, n)
for n in ("finish", "start_backward", "finish_backward"):
    wrap(eb.ExplicitLookupStep, n)
wrap(dlrm.DLRMTrain, "finish_dense_grads")
wrap(dlrm.DLRMTrain, "_explicit_step")
wrap(torch, "empty", "torch.empty")
wrap(torch.cuda.Event, "record", "torch.cuda.Event.record")
wrap(torch.cuda.CUDAGraph, "replay", "CUDAGraph.replay")
wrap(tp.TrainPipelineSparseDist, "progress")
wrap(tp.TrainPipelineSparseDist, "_start_data_dist")
wrap(torch.Tensor, "backward", "loss.backward")
wrap(keyed.CombinedOptimizer, "step")
wrap(keyed.CombinedOptimizer, "zero_grad")
wrap(dlrm.DLRMTrain, "forward")
wrap(torch.distributed, "all_to_all_single", "dist.all_to_all_single")
wrap(torch.distributed, "all_reduce", "dist.all_reduce")

_args = bench.parse()
_orig_progress = tp.TrainPipelineSparseDist.progress
_seen = [0]
_SPIN = float(os.environ.get("STEP_SPIN_US", "0")) * 1e-6


_last_end = [None]
_between = [0, 0.0]


def _progress(self, it):
    if _last_end[0] is not None:
        _between[0] += 1
        _between[1] += time.perf_counter() - _last_end[0]
    try:
        return _progress2(self, it)
    finally:
        _last_end[0] = time.perf_counter()


def _progress2(self, it):
    _seen[0] += 1
    if _seen[0] == _args.warmup + 1:  # first timed step: forget warm-up (first-use builds, graph capture)
        with lock:
            acc.clear()
        _between[0], _between[1] = 0, 0.0
    if _SPIN > 0:  # host-bound or GPU-bound?  burn host time per step and see whether the step gets longer
        t_end = time.perf_counter() + _SPIN
        while time.perf_counter() < t_end:
            pass
    return _orig_progress(self, it)


tp.TrainPipelineSparseDist.progress = _progress
bench.main(_args)
ms = torch.cuda.memory_stats()
print("[step_timers] device allocs", ms.get("num_device_alloc"), "device frees", ms.get("num_device_free"), "alloc retries",
      ms.get("num_alloc_retries"), "sync_all_streams", ms.get("num_sync_all_streams"), file=sys.stderr)
steps = acc["TrainPipelineSparseDist.progress"][0]
with open(out, "w") as f:
    f.write(f"{steps} steps (after warm-up); us per step, calls per step\n")
    f.write(f"{_between[1] / max(_between[0], 1) * 1e6:9.1f} us         between two progress() calls (the caller's loop)\n")
    for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        f.write(f"{t / steps * 1e6:9.1f} us  {n / steps:5.2f}x  {k}\n")
