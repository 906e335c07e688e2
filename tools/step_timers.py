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

sys.path.insert(0, os.path.join(bench.ROOT, "tests"))
import _paths  # noqa: E402,F401
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


def _progress(self, it):
    _seen[0] += 1
    if _seen[0] == _args.warmup + 1:  # first timed step: forget warm-up (first-use builds, graph capture)
        with lock:
            acc.clear()
    if _SPIN > 0:  # host-bound or GPU-bound?  burn host time per step and see whether the step gets longer
        t_end = time.perf_counter() + _SPIN
        while time.perf_counter() < t_end:
            pass
    return _orig_progress(self, it)


tp.TrainPipelineSparseDist.progress = _progress
bench.main(_args)
steps = acc["TrainPipelineSparseDist.progress"][0]
with open(out, "w") as f:
    f.write(f"{steps} steps (after warm-up); us per step, calls per step\n")
    for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        f.write(f"{t / steps * 1e6:9.1f} us  {n / steps:5.2f}x  {k}\n")
