"""Development tool: GPU-busy time per step from a rocprofv3 kernel_trace.csv — sums kernel durations
(and the union of their intervals) inside the window spanned by the last `steps` launches of a marker
kernel.  Usage: python tools/busy.py <kernel_trace.csv> <marker substring> <steps>"""
import csv
import sys

path, marker, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
rows.sort()
marks = [s for s, e, n in rows if marker in n]
lo, hi = marks[-steps - 1], marks[-1]
win = [(s, e, n) for s, e, n in rows if lo <= s < hi]
total = sum(e - s for s, e, _ in win)
union, cur_s, cur_e = 0, None, None
for s, e, _ in win:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print(f"{steps} steps: wall {(hi - lo) / steps / 1e3:.1f} us/step, kernel sum {total / steps / 1e3:.1f} us/step, "
      f"GPU busy (union) {union / steps / 1e3:.1f} us/step, launches/step {len(win) / steps:.1f}")
by = {}
for s, e, n in win:
    k = n[:70]
    by[k] = by.get(k, 0) + (e - s)
for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:14]:
    print(f"  {v / steps / 1e3:8.1f} us/step  {k}")
