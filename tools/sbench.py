"""Micro-benchmark of the pair sort (tbe_sort_pairs) on Criteo-shaped row keys: µs per sort, HIP events on the
launch stream.  Usage: python tools/sbench.py [--iters 50]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import _paths  # noqa: E402,F401
from fbgemm_gpu import _lib  # noqa: E402
from fbgemm_gpu._lib import check, ptr, stream_ptr  # noqa: E402
from test_sort_gpu import make_keys  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--case", type=int, default=-1, help="run only this case index")
    args = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(0)
    cases = ((1_703_936, 28, 4), (1_703_936, 28, 8), (212_992, 26, 4), (106_496, 28, 4), (106_496, 28, 8),
             (8_000_000, 28, 4))
    for ci, (n, bits, pbytes) in enumerate(cases):
        if args.case >= 0 and ci != args.case:
            continue
        keys = torch.from_numpy(make_keys(rng, n, bits, np.uint32, "criteo").view(np.int32)).to(dev)
        pay = torch.arange(n, dtype=torch.int32 if pbytes == 4 else torch.int64, device=dev)
        k, p = keys.clone(), pay.clone()
        kt, pt = torch.empty_like(k), torch.empty_like(p)
        nbytes = lib.tbe_sort_pairs_workspace_bytes(n, bits)
        ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
        off = (-ws.data_ptr()) % 256
        ts = []
        for it in range(args.iters + 5):
            k.copy_(keys)
            p.copy_(pay)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            check(lib.tbe_sort_pairs(ptr(k), ptr(kt), ptr(p), ptr(pt), n, bits, 4, pbytes, ws.data_ptr() + off, nbytes,
                                     stream_ptr(dev)), "tbe_sort_pairs")
            b.record()
            torch.cuda.synchronize()
            if it >= 5:
                ts.append(a.elapsed_time(b) * 1e3)
        ts.sort()
        print(f"n={n:>9} key_bits={bits} payload={pbytes}B  median {ts[len(ts) // 2]:7.1f} us  min {ts[0]:7.1f} us "
              f"(incl. copy-back when the pass count is odd)", flush=True)


if __name__ == "__main__":
    main()
