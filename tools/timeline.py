"""Development tool: one train step of a rocprofv3 kernel_trace.csv as a timeline — start offset, duration, gap to the
previous kernel END on any stream, stream id, kernel.  Usage: python tools/timeline.py <kernel_trace.csv> <marker> [step_from_end]"""
import csv
import sys

path, marker = sys.argv[1], sys.argv[2]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", "?"), r.get("Queue_Id", "?"))
              for r in csv.DictReader(open(path)))
marks = [s for s, e, n, q, qq in rows if marker in n]
# the marker kernel may run more than once per step: a step = span between every k-th occurrence
per = int(sys.argv[4]) if len(sys.argv) > 4 else 1
lo, hi = marks[-(back + 1) * per - 1], marks[-back * per - 1]
win = [r for r in rows if lo <= r[0] < hi]
print(f"step wall {(hi - lo) / 1e3:.1f} us, {len(win)} kernels")
last_end = lo
for s, e, n, q, qq in win:
    short = n.split("(")[0].replace("void ", "")[-60:]
    print(f"{(s - lo) / 1e3:8.1f} +{(e - s) / 1e3:7.1f}  gap {max(0, s - last_end) / 1e3:6.1f}  s{q:>3s}  {short}")
    last_end = max(last_end, e)
