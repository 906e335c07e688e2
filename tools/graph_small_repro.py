"""Development repro: the tiny e2e model of tests/test_multirank_gpu.py, ONE process, HIP-graph segments."""
import faulthandler
import os
import sys

faulthandler.enable()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import _paths  # noqa: E402,F401
import torch  # noqa: E402

import test_multirank_gpu as T  # noqa: E402
from torchrec_amd.distributed.types import ShardingEnv  # noqa: E402

import torch.distributed as dist  # noqa: E402

dev = torch.device("cuda", 0)
backend = os.environ.get("REPRO_BACKEND", "")
if backend:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    torch.cuda.set_device(0)
    dist.init_process_group(backend, rank=0, world_size=1, **({"device_id": dev} if backend == "nccl" else {}))
    env = ShardingEnv.from_process_group(dist.group.WORLD)
else:
    env = ShardingEnv.from_local(1, 0)
keys, model, opt = T._e2e_model(env, dev, dp_max_rows=0, graph_batch=int(os.environ.get('GRAPH_BATCH', '0')))
print("ddp:", type(model._dmp_wrapped_module).__name__, flush=True)
import torchrec_amd.distributed.train_pipeline as tp  # noqa: E402
_orig = tp.TrainPipelineSparseDist.__init__


def _init(self, m, o, d, hip_graphs=False):
    _orig(self, m, o, d, hip_graphs=False)
    self._hip_graphs = hip_graphs


tp.TrainPipelineSparseDist.__init__ = _init
T._e2e_init_tables(model)
out = T._e2e_run(model, opt, keys, T._e2e_batches(2), 0, 1, dev, not int(os.environ.get('GRAPH_BATCH', '0')))
print("graphs:", model.module._graphs is not None, "losses", out[0])
