"""Development tool: list launches of kernels matching a substring from a rocprofv3 kernel_trace.csv
(grid size, duration, start offset) to find out what a hot anonymous kernel (FillFunctor, ...) belongs to."""
import csv
import sys

path, pat = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(path)))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
idx = {id(r): i for i, r in enumerate(rows)}
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for i, r in enumerate(rows):
    if pat in r["Kernel_Name"]:
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        prev = rows[i - 1]["Kernel_Name"][:60] if i else ""
        nxt = rows[i + 1]["Kernel_Name"][:60] if i + 1 < len(rows) else ""
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e6:10.3f} ms  {dur:8.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>10}  "
              f"prev={prev}  next={nxt}")
