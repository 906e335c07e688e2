"""Development tool: idle gaps of the GPU inside one train step, from a rocprofv3 kernel_trace.csv — for the last
`steps` windows between launches of a marker kernel, every interval > `min_us` in which no kernel ran, with the
kernels that ended before / started after it (averaged by (before, after) pair).
Usage: python tools/gaps.py <kernel_trace.csv> <marker substring> <steps> [min_us]"""
import csv
import sys

path, marker, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
min_ns = float(sys.argv[4]) * 1e3 if len(sys.argv) > 4 else 3e3
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", "?"))
        for r in csv.DictReader(open(path))]
rows.sort()
marks = [s for s, e, n, q in rows if marker in n]
lo, hi = marks[-steps - 1], marks[-1]
win = [r for r in rows if lo <= r[0] < hi]
gaps, cur_e, cur_n = {}, None, None
total = 0
for s, e, n, q in win:
    if cur_e is not None and s - cur_e > min_ns:
        k = (cur_n[:48], n[:48])
        g = gaps.setdefault(k, [0, 0])
        g[0] += 1
        g[1] += s - cur_e
        total += s - cur_e
    if cur_e is None or e > cur_e:
        cur_e, cur_n = e, n
print(f"{steps} steps: wall {(hi - lo) / steps / 1e3:.1f} us/step, idle in gaps > {min_ns / 1e3:.0f} us: {total / steps / 1e3:.1f} us/step")
for (a, b), (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"  {t / steps / 1e3:7.1f} us/step  {c / steps:4.1f}x  {a}  ->  {b}")
