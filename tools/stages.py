"""Per-stage attribution of a train step from one `rocprofv3 --kernel-trace --marker-trace --hip-trace` run.

The path's host code brackets its stages with roctx ranges carrying the reference's label strings
(torchrec_amd/profiling.py, TORCHREC_AMD_PROFILE_LABELS=roctx; reference: train_pipeline.py:504-550, dist_data.py:190-388,
comm_ops.py:489-921).  A kernel belongs to the innermost range that was open on the launching thread when its launch call
(hipLaunchKernel / hipGraphLaunch / hipMemcpyAsync ..., matched by correlation id) was made; work launched from other
threads (RCCL's progress thread) or outside every range is listed as such.  Output: GPU time per step and label, averaged
over the last `steps` "## forward ##" ranges.  Tracing slows the HOST (the wall time per step under the tracer is not the
un-traced step time — that is bench.py's own line); kernel durations and their attribution are what this is for.

Usage: python tools/stages.py <kernel_trace.csv> <marker_api_trace.csv> <hip_api_trace.csv> [steps]"""
import collections
import csv
import sys


def main():
    kpath, mpath, hpath = sys.argv[1:4]
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 30
    K = list(csv.DictReader(open(kpath)))
    M = list(csv.DictReader(open(mpath)))
    H = {r["Correlation_Id"]: r for r in csv.DictReader(open(hpath))}
    by_thread = collections.defaultdict(list)
    for r in M:
        by_thread[r["Thread_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]))
    for v in by_thread.values():
        v.sort()
    fw = sorted((int(r["Start_Timestamp"]), r["Thread_Id"]) for r in M if r["Function"] == "## forward ##")
    if len(fw) < steps + 1:
        steps = len(fw) - 1
    lo, hi = fw[-steps - 1][0], fw[-1][0]
    main_thread = fw[-1][1]

    def open_ranges(thread, t):
        return sorted((s, name) for s, e, name in by_thread.get(thread, ()) if s <= t <= e)

    per, per_path, launches = collections.Counter(), collections.Counter(), collections.Counter()
    kernels_by = collections.defaultdict(collections.Counter)
    for k in K:
        h = H.get(k["Correlation_Id"])
        if h is None:
            continue
        t = int(h["Start_Timestamp"])
        if not (lo <= t < hi):
            continue
        dur = int(k["End_Timestamp"]) - int(k["Start_Timestamp"])
        if h["Thread_Id"] != main_thread:
            lab = p = f"(thread {h['Thread_Id']}: not the training thread)"
        else:
            rs = open_ranges(h["Thread_Id"], t)
            lab = rs[-1][1] if rs else "(outside every range)"
            p = " > ".join(n for _, n in rs) or lab
        per[lab] += dur
        per_path[p] += dur
        launches[lab] += 1
        kernels_by[lab][k["Kernel_Name"].split("(")[0][-70:]] += dur
    tot = sum(per.values())
    print(f"{steps} steps; GPU kernel time per step {tot / steps / 1e3:.1f} us; host wall per step under the tracer "
          f"{(hi - lo) / steps / 1e3:.1f} us")
    print(f"{'innermost label':52s} {'us/step':>9s} {'%':>6s} {'launches/step':>14s}")
    for lab, v in per.most_common():
        print(f"{lab:52s} {v / steps / 1e3:9.1f} {100.0 * v / tot:6.1f} {launches[lab] / steps:14.1f}")
        for kn, kv in kernels_by[lab].most_common(5):
            print(f"      {kv / steps / 1e3:8.1f}  {kn}")
    print("\nby full range path:")
    for p, v in per_path.most_common(20):
        print(f"  {v / steps / 1e3:9.1f} us/step  {p}")


if __name__ == "__main__":
    main()
