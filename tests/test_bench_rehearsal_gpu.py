"""bench.py end to end on the GPU box, small tables: the default N = 1 path, and the one-rank rehearsal of
everything a rank of an N > 1 run executes (RCCL group, id / pooled all-to-all + exchange kernels,
replicated tiny tables, DistributedDataParallel, HIP graphs captured before the DDP wrap)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *flags):
    env = dict(os.environ, **extra_env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--row-cap", "200000", "--steps", "4", "--warmup", "3",
                          "--no-cpu-baseline", "--num-batches", "3", *flags], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_bench_default_path():
    d = _run({}, "--global-batch", "4096")
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["roofline"]["bound"] == "hbm" and d["hip_graphs"] is True
    assert d["metric"].startswith("samples/sec") and d["unit"] == "samples/s" and d["scaling"] == "strong"
    assert d["launcher"] == "direct" and d["rccl_ranks"] == 0
    e2e = d["roofline"]["end_to_end"]
    assert 0 < e2e["hbm_frac"] < 1 and 0 < e2e["mfma_f32_frac"] < 1 and e2e["hbm_bytes_per_sample"] == 67392


def test_bench_one_rank_rehearsal_of_the_multi_gpu_path():
    env = {"TORCHREC_AMD_FORCE_EXCHANGE": "1", "TORCHREC_AMD_FORCE_DDP": "1", "TORCHREC_AMD_FORCE_DP": "1",
           "MASTER_PORT": "29561"}
    d = _run(env, "--global-batch", "4096", "--spawn")  # through bench.py's own launcher, as `--gpus N` goes
    assert "11 replicated" in d["config"]["parallelism"] and d["hip_graphs"] is True and d["value"] > 0
    assert d["launcher"] == "bench.py" and d["rccl_ranks"] == 1 and d["backend"] == "nccl"
    assert d["config"]["plan"]["data_parallel"] == 11 and d["hip_graphs_note"] == "on"
    assert d["binding"]["resource"] in ("mfma_f32", "hbm_embedding", "xgmi_busiest_link")
    e = _run(env, "--global-batch", "4096", "--hip-graphs", "off", "--tuned-gemms", "off")
    assert e["hip_graphs"] is False and e["tuned_gemms"] is False and e["value"] > 0
