"""Per-kernel HBM traffic from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection.csv files.
FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE counts
128-B requests as 64 B for wide (16 B/lane) coalesced reads -> doubled; WRITE_SIZE is exact for
16-B-per-lane stores.  Usage: pmc_summary.py <fetch_csv> <write_csv> [name-substring ...]"""
import csv
import json
import sys
from collections import defaultdict


def load(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    pats = sys.argv[3:] or ["tbe::"]
    out = {}
    for name in sorted(set(fetch) | set(write)):
        if not any(p in name for p in pats):
            continue
        short = name.split("(")[0].replace("void ", "")
        f = fetch.get(name, [])
        w = write.get(name, [])
        # skip the first launches (warm-up) when there are enough samples
        f = f[len(f) // 3:] if len(f) > 3 else f
        w = w[len(w) // 3:] if len(w) > 3 else w
        fk = sum(f) / len(f) if f else 0.0
        wk = sum(w) / len(w) if w else 0.0
        out[short] = {"launches": max(len(f), len(w)), "FETCH_SIZE_KiB_raw": round(fk, 1),
                      "WRITE_SIZE_KiB_raw": round(wk, 1),
                      "hbm_read_MB_corrected_x2": round(2 * fk * 1024 / 1e6, 1),
                      "hbm_write_MB": round(wk * 1024 / 1e6, 1),
                      "hbm_total_MB": round((2 * fk + wk) * 1024 / 1e6, 1)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
