"""Development probe: cProfile of the host side of bench.py's train step (after `skip` pipeline steps), to find
where the Python / launch time of a step goes.  Usage: python tools/step_cprofile.py <out.txt> <skip> [bench args]"""
import cProfile
import io
import os
import pstats
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
out, skip = sys.argv[1], int(sys.argv[2])
sys.argv = [sys.argv[0]] + sys.argv[3:]
import bench  # noqa: E402

sys.path.insert(0, os.path.join(bench.ROOT, "tests"))
import _paths  # noqa: E402,F401
from torchrec_amd.distributed.train_pipeline import TrainPipelineSparseDist  # noqa: E402

prof = cProfile.Profile()
calls = [0]
orig = TrainPipelineSparseDist.progress


def progress(self, it):
    calls[0] += 1
    if calls[0] <= skip:
        return orig(self, it)
    prof.enable()
    try:
        return orig(self, it)
    finally:
        prof.disable()


TrainPipelineSparseDist.progress = progress
bench.main(bench.parse())
steps = max(1, calls[0] - skip)
for key in ("tottime", "cumulative"):
    buf = io.StringIO()
    pstats.Stats(prof, stream=buf).sort_stats(key).print_stats(45)
    with open(out, "a") as f:
        f.write(f"==== {steps} profiled steps, sorted by {key}\n" + buf.getvalue())
