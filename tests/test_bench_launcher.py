"""bench.py's built-in launcher (`python bench.py --gpus N` with no WORLD_SIZE): it must start N fresh rank
processes before anything touches the GPU, relay rank 0's ONE JSON line and return the ranks' exit code.
Driven here on CPU over gloo through TORCHREC_AMD_BENCH_DRYRUN (no model, no GPU); the GPU rehearsal
(tests/test_bench_rehearsal_gpu.py) goes through the same launcher with --spawn.
Reference launch line: examples/dlrm/README.MD:17-28, examples/dlrm/dlrm_main.py:469-478."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(env_extra, *flags, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_launcher_starts_two_ranks_and_relays_one_json_line():
    out = _bench({"TORCHREC_AMD_BENCH_DRYRUN": "1"}, "--gpus", "2", "--steps", "3", "--warmup", "1")
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout  # ONE line on stdout, whatever the ranks or torchrun print
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == 2 and d["ranks"] == 2 and d["backend"] == "gloo"
    assert d["steps"] == 3 and d["warmup"] == 1 and d["launcher"] == "bench.py" and d["value"] > 0


def test_launcher_parent_never_imports_torch():
    """The parent must not initialise the GPU: it does not even import torch (checked by poisoning the import
    for the parent only: the children get a clean environment variable back)."""
    code = (
        "import sys, os, runpy\n"
        "class Block:\n"
        "    def find_spec(self, name, path=None, target=None):\n"
        "        if name == 'torch' or name.startswith('torch.'):\n"
        "            raise ImportError('the launcher parent imported torch')\n"
        "sys.meta_path.insert(0, Block())\n"
        f"sys.argv = [{os.path.join(ROOT, 'bench.py')!r}, '--gpus', '2', '--steps', '2', '--warmup', '1']\n"
        f"runpy.run_path({os.path.join(ROOT, 'bench.py')!r}, run_name='__main__')\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["TORCHREC_AMD_BENCH_DRYRUN"] = "1"
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert json.loads(out.stdout.strip().splitlines()[-1])["ranks"] == 2


def test_launcher_propagates_a_rank_failure():
    out = _bench({"TORCHREC_AMD_BENCH_DRYRUN": "fail"}, "--gpus", "2", "--steps", "2", "--warmup", "1")
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]  # no result line from a failed run


def test_external_torchrun_is_still_a_rank():
    """The driver's own launch line (torch.distributed.run sets WORLD_SIZE): bench.py must not spawn again."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["TORCHREC_AMD_BENCH_DRYRUN"] = "1"
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29577", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")][-1])
    assert d["ranks"] == 2 and d["launcher"] == "external"
