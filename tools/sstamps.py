"""Where a sort pass spends its time: per-segment wall-clock stamps written by radix_pass_kernel
(tbe_debug_set_sort_stamps).  Prints, per pass, the median / max over segments of each phase's END time
relative to the first segment's start.  Usage: python tools/sstamps.py [n] [key_bits]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import _paths  # noqa: E402,F401
from fbgemm_gpu import _lib  # noqa: E402
from fbgemm_gpu._lib import check, ptr, stream_ptr  # noqa: E402
from test_sort_gpu import make_keys  # noqa: E402

PHASES = ["ticket", "count+hist", "level-0 rows", "level-1 rows", "base scan", "place (LDS)", "write out"]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_703_936
    bits = int(sys.argv[2]) if len(sys.argv) > 2 else 28
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(0)
    keys = torch.from_numpy(make_keys(rng, n, bits, np.uint32, "criteo").view(np.int32)).to(dev)
    pay = torch.arange(n, dtype=torch.int32, device=dev)
    kt, pt = torch.empty_like(keys), torch.empty_like(pay)
    nbytes = lib.tbe_sort_pairs_workspace_bytes(n, bits)
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
    off = (-ws.data_ptr()) % 256
    stamps = torch.zeros(7 * 256 * 8 + 8, dtype=torch.int64, device=dev)
    for it in range(4):
        k, p = keys.clone(), pay.clone()
        if it == 3:
            lib.tbe_debug_set_sort_stamps(stamps.data_ptr())
        check(lib.tbe_sort_pairs(ptr(k), ptr(kt), ptr(p), ptr(pt), n, bits, 4, 4, ws.data_ptr() + off, nbytes,
                                 stream_ptr(dev)), "tbe_sort_pairs")
        torch.cuda.synchronize()
    lib.tbe_debug_set_sort_stamps(None)
    st = stamps.cpu().numpy()[:7 * 256 * 8].reshape(7, 256, 8)
    for p in range(7):
        used = st[p, :, 7] != 0
        if not used.any():
            continue
        s = st[p][used].astype(np.float64)
        t0 = s[:, 7].min()
        print(f"pass {p}: {int(used.sum())} segments; starts spread {(s[:, 7].max() - t0) / 100:.2f} us")
        for i, name in enumerate(PHASES):
            rel = (s[:, i] - t0) / 100.0  # 100 MHz -> us
            dur = (s[:, i] - (s[:, i - 1] if i else s[:, 7])) / 100.0
            print(f"   {name:14s} ends at median {np.median(rel):6.2f} max {rel.max():6.2f} us | phase takes median "
                  f"{np.median(dur):5.2f} max {dur.max():5.2f} us")


if __name__ == "__main__":
    main()
