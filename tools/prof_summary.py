"""Prints a rocprofv3 kernel_stats.csv with kernel names shortened (they can be kilobytes long).
Usage: python tools/prof_summary.py <kernel_stats.csv> [max_rows]"""
import csv
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.search(r"(radix_sort_onesweep_\w+|radix_sort_\w+)", name)
    if "rocprim" in name and m:
        return "rocprim::" + m.group(1)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(<[^<>]{0,60}>)?)", name)
    out = m.group(1) if m else name
    if out.startswith("at::native") and "elementwise" in name:
        inner = re.search(r"(\w+Functor|\w+_kernel_cuda|uniform|normal|random_from_to|arange)", name)
        out += "[" + (inner.group(1) if inner else "?") + "]"
    return out[:110]


def main():
    path = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rows = list(csv.DictReader(open(path)))
    print(f"{'kernel':110s} {'calls':>6s} {'avg_us':>10s} {'total_ms':>10s} {'pct':>6s}")
    for r in rows[:n]:
        print(f"{short(r['Name']):110s} {int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:10.1f} "
              f"{float(r['TotalDurationNs']) / 1e6:10.3f} {float(r['Percentage']):6.2f}")


if __name__ == "__main__":
    main()
