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


def _bench(extra_env, *flags, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", *flags], env=env,
                          capture_output=True, text=True, timeout=timeout)


def _run(extra_env, *flags):
    out = _bench(extra_env, "--row-cap", "200000", "--steps", "4", "--warmup", "3", "--num-batches", "3", *flags)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout  # ONE line on stdout
    return json.loads(lines[0])


def _close(a, b, rel):
    """checksums [sum, sum |x|]: equal to `rel` of the absolute mass."""
    return abs(a[0] - b[0]) <= rel * max(a[1], b[1]) and abs(a[1] - b[1]) <= rel * max(a[1], b[1])


def test_bench_default_path():
    d = _run({}, "--global-batch", "4096")
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["roofline"]["bound"] == "hbm" and d["hip_graphs"] is True
    assert d["metric"].startswith("samples/sec") and d["unit"] == "samples/s" and d["scaling"] == "strong"
    assert d["launcher"] == "direct" and d["rccl_ranks"] == 0
    e2e = d["roofline"]["end_to_end"]
    assert 0 < e2e["hbm_frac"] < 1 and 0 < e2e["mfma_f32_frac"] < 1 and e2e["hbm_bytes_per_sample"] == 67392
    # per-step HIP-event times with a median, the on-box copy rate next to the spec peak
    assert 0 < d["min_max_ms_per_step"][0] <= d["median_ms_per_step"] <= d["min_max_ms_per_step"][1]
    assert 2000 < d["roofline"]["peak_measured_copy_GBs"] < 8000 and d["roofline"]["peak"] == 8000.0
    c = d["checks"]
    assert c["sort_giveups"] == 0 and c["bounds_check_errors"] == 0 and c["dense_replicas_identical"] is True
    assert 0.3 < c["loss_first"] < 1.5 and 0.3 < c["loss_last"] < 1.5
    # "no work skipped": the graph + explicit-step run (what the line above timed) against an eager, autograd-driven
    # run of the same program on the same batches and the same initial model
    e = _run({}, "--global-batch", "4096", "--hip-graphs", "off")
    assert e["hip_graphs"] is False and e["explicit_backward_steps"] == 0 and d["explicit_backward_steps"] == 7
    for k in ("loss_first", "loss_last"):
        assert abs(c[k] - e["checks"][k]) <= 2e-4 * abs(c[k]), (k, c[k], e["checks"][k])
    for k in ("dense", "embedding"):
        assert _close(c["param_checksum"][k], e["checks"]["param_checksum"][k], 2e-6), (k, c["param_checksum"], e["checks"])


def test_bench_run_with_a_kernel_fault_is_invalid():
    """A sort give-up during the run (played by the host-side test hook) must not produce a result line."""
    out = _bench({"TORCHREC_AMD_BENCH_INJECT_FAULT": "1"}, "--row-cap", "200000", "--steps", "2", "--warmup", "1",
                 "--num-batches", "2", "--global-batch", "4096")
    assert out.returncode == 4, out.stderr[-2000:]
    assert not [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert "INVALID RUN: sort give-ups 1" in out.stderr


def test_bench_main_at_world_2_on_one_gpu_matches_world_1():
    """bench.main() with world > 1 — the pinned mixed plan (4 row-wise), the all-rank graph
    decision, init_data_parallel after capture, the flat-gradient all-reduce, per-rank accounting — executed for real:
    two ranks started by bench.py's own launcher share GPU 0 over gloo (TORCHREC_AMD_BENCH_BACKEND; RCCL refuses two
    ranks on one device), FULL-SIZE tables, default plan, graphs auto.  The losses and the parameter checksums must
    equal a world-size-1 run on the same global batches (--data-ranks 2) from the same initial model.
    Launch line: examples/dlrm/README.MD:17-28, dlrm_main.py:469-478."""
    common = ("--steps", "6", "--warmup", "2", "--num-batches", "4")
    out = _bench({"TORCHREC_AMD_BENCH_BACKEND": "gloo"}, "--gpus", "2", *common)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["launcher"] == "bench.py" and d["backend"] == "gloo" and "NOT a multi-GPU" in d["rehearsal"]
    assert d["vs_baseline"] is None and d["scaling"] == "strong" and d["value"] > 0
    plan = d["config"]["plan"]
    assert plan["row_wise"] == 4 and plan["data_parallel"] > 0 and sum(plan["table_wise_per_rank"]) == plan["table_wise"]
    assert plan["source"].startswith("pinned mixed plan") and d["config"]["local_batch"] == 32768
    assert d["binding"]["xgmi_bytes_per_rank_per_step"] > 0
    assert d["hip_graphs"] is True and d["hip_graphs_note"] == "on" and d["explicit_backward_steps"] == 8
    # the N = 2 default at the global batch of 65 536: the exchange in two half-batches (auto from 32 768 samples per rank)
    assert d["half_batch_steps"] == 8 and d["prefetched_lookups"] == 0  # (no late weight gradients in half-batch mode)
    allk = d["roofline"]["all"]
    assert allk["tbe_fwd_short_kernel"]["launches"] >= 6 and allk["bwd_update_kernel"]["avg_us"] > 0
    c = d["checks"]
    assert c["sort_giveups"] == 0 and c["bounds_check_errors"] == 0 and c["dense_replicas_identical"] is True
    one = _bench({}, "--gpus", "1", "--data-ranks", "2", *common)
    assert one.returncode == 0, one.stderr[-3000:]
    o = json.loads(one.stdout.strip().splitlines()[-1])
    assert o["n_gpus"] == 1 and o["hip_graphs"] is False and o["config"]["data_ranks"] == 2
    for k in ("loss_first", "loss_last"):
        assert abs(c[k] - o["checks"][k]) <= 2e-4 * abs(c[k]), (k, c[k], o["checks"][k])
    for k in ("dense", "embedding"):
        assert _close(c["param_checksum"][k], o["checks"]["param_checksum"][k], 2e-6), (k, c["param_checksum"], o["checks"])


def test_bench_one_rank_rehearsal_of_the_multi_gpu_path():
    env = {"TORCHREC_AMD_FORCE_EXCHANGE": "1", "TORCHREC_AMD_FORCE_DDP": "1", "TORCHREC_AMD_FORCE_DP": "1",
           "MASTER_PORT": "29561"}
    d = _run(env, "--global-batch", "4096", "--spawn")  # through bench.py's own launcher, as `--gpus N` goes
    assert "11 replicated" in d["config"]["parallelism"] and d["hip_graphs"] is True and d["value"] > 0
    assert d["launcher"] == "bench.py" and d["rccl_ranks"] == 1 and d["backend"] == "nccl"
    assert d["config"]["plan"]["data_parallel"] == 11 and d["hip_graphs_note"] == "on"
    assert d["binding"]["resource"] in ("mfma_f32", "hbm_embedding", "xgmi_busiest_link")
    assert d["checks"]["sort_giveups"] == 0 and d["checks"]["dense_replicas_identical"] is True
    e = _run(env, "--global-batch", "4096", "--hip-graphs", "off", "--tuned-gemms", "off")
    assert e["hip_graphs"] is False and e["tuned_gemms"] is False and e["value"] > 0


def test_bench_rehearsal_with_the_exchange_in_two_half_batches():
    """TORCHREC_AMD_HALF_BATCHES=1: two half-batch all-to-alls each way and two head segments per step, over the one-rank
    RCCL group (list-form all_to_all on row-range views), with an emulated link time — same training as whole batches up
    to the summation order (loss = mean of the halves' means, weight gradients = sum of the halves')."""
    env = {"TORCHREC_AMD_FORCE_EXCHANGE": "1", "TORCHREC_AMD_FORCE_DP": "1", "MASTER_PORT": "29563",
           "TORCHREC_AMD_REHEARSAL_LINK_US": "50"}
    whole = _run(dict(env, TORCHREC_AMD_HALF_BATCHES="0"), "--global-batch", "4096")
    # (with two late weight-gradient layers and the prefetch behind them: the half-batch step's own version of both)
    halves = _run(dict(env, TORCHREC_AMD_HALF_BATCHES="1", TORCHREC_AMD_WGRAD_LATE_LAYERS="2"), "--global-batch", "4096")
    assert halves["prefetched_lookups"] == 6
    assert whole["half_batch_steps"] == 0 and halves["half_batch_steps"] == 7 == halves["explicit_backward_steps"]
    assert halves["checks"]["sort_giveups"] == 0 and halves["checks"]["bounds_check_errors"] == 0
    for k in ("loss_first", "loss_last"):
        assert abs(whole["checks"][k] - halves["checks"][k]) <= 2e-4 * abs(whole["checks"][k]), (k, whole["checks"], halves["checks"])
    for k in ("dense", "embedding"):
        assert _close(whole["checks"]["param_checksum"][k], halves["checks"]["param_checksum"][k], 2e-6), k


def test_late_weight_gradients_behind_the_prefetched_lookup_change_no_bit():
    """Default explicit step (part of the head's weight gradients replayed behind the NEXT step's prefetched lookup +
    all-to-all) against the same run with everything in its old place: a reordering of independent work, so losses and
    parameter checksums must be IDENTICAL — in particular the prefetched step must not read the replicated tables before
    the dense optimizer has updated them."""
    env = {"TORCHREC_AMD_FORCE_EXCHANGE": "1", "TORCHREC_AMD_FORCE_DP": "1", "MASTER_PORT": "29564"}
    plain = _run(dict(env, TORCHREC_AMD_WGRAD_LATE_LAYERS="0", TORCHREC_AMD_PREFETCH_LOOKUP="0"), "--global-batch", "4096")
    env = dict(env, TORCHREC_AMD_WGRAD_LATE_LAYERS="3")  # (auto keeps the late graph for per-rank batches of 16 384+)
    late = _run(env, "--global-batch", "4096")
    assert late["explicit_backward_steps"] == plain["explicit_backward_steps"] == 7
    assert late["prefetched_lookups"] > 0 and plain["prefetched_lookups"] == 0
    assert late["checks"]["loss_first"] == plain["checks"]["loss_first"]
    assert late["checks"]["loss_last"] == plain["checks"]["loss_last"]
    assert late["checks"]["param_checksum"] == plain["checks"]["param_checksum"]
    # the same again with the split-off weight gradients FIRST in the backward window, their all-reduce behind them (opt-in)
    early = _run(dict(env, TORCHREC_AMD_WGRAD_SPLIT_MODE="early", TORCHREC_AMD_FORCE_DENSE_REDUCE="1"), "--global-batch", "4096")
    assert early["prefetched_lookups"] == 0
    assert early["checks"]["loss_last"] == plain["checks"]["loss_last"]
    assert early["checks"]["param_checksum"] == plain["checks"]["param_checksum"]
    # the same again with the dense gradient all-reduces of an N > 1 run ISSUED (head slice without the late part, the rest,
    # the late part; on their own RCCL communicator) although one rank needs none: the call sequence of a real rank
    forced = _run(dict(env, TORCHREC_AMD_FORCE_DENSE_REDUCE="1"), "--global-batch", "4096")
    assert forced["checks"]["loss_last"] == plain["checks"]["loss_last"]
    assert forced["checks"]["param_checksum"] == plain["checks"]["param_checksum"]
    # the same again with the exchange's unpack / pack captured into the head segment's graphs (persistent receive /
    # send buffers; opt-in): the same kernels on the same data
    graphed = _run(dict(env, TORCHREC_AMD_GRAPH_EXCHANGE="1"), "--global-batch", "4096")
    assert graphed["hip_graphs_note"] == "on" and graphed["explicit_backward_steps"] == 7
    assert graphed["checks"]["loss_last"] == plain["checks"]["loss_last"]
    assert graphed["checks"]["param_checksum"] == plain["checks"]["param_checksum"]
