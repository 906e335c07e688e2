"""Timing of the pooled exchange receiver/sender kernels at the 8-GPU shape (no communication:
the exchange buffer is synthetic).  Development tool; numbers quoted in DESIGN.md."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import _paths  # noqa: E402,F401
import fbgemm_gpu  # noqa: E402,F401
import torchrec_amd.distributed._device_ops  # noqa: E402,F401


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def main():
    W, Bl, D, F = 8, 8192, 128, 26
    dev = "cuda"
    # 2 row-wise features (on every rank) + 24 table-wise (3 per rank), as the planner lays Criteo out
    n_rw = 2
    feat_src = [-1] * n_rw + [r for r in range(W) for _ in range(3)]
    feat_slab_col = [0, D] + [(n_rw + i) * D for _ in range(W) for i in range(3)]
    D_loc = (n_rw + 3) * D
    out_col = [i * D for i in range(F + 1)]
    t = lambda v, dt: torch.tensor(v, dtype=dt, device=dev)  # noqa: E731
    feat_out_col, fsrc, fcol = t(out_col, torch.int32), t(feat_src, torch.int32), t(feat_slab_col, torch.int32)
    slab_off = t([r * Bl * D_loc for r in range(W)], torch.int64)
    slab_stride = t([D_loc] * W, torch.int32)
    recv = torch.randn(W * Bl * D_loc, device=dev)
    grad = torch.randn(Bl, F * D, device=dev)
    us = timeit(lambda: torch.ops.tbe_hip.pooled_exchange_unpack(recv, feat_out_col, fsrc, fcol, slab_off, slab_stride,
                                                                 Bl, F * D, True, 1.0))
    rd = (24 + n_rw * W) * D * 4 * Bl
    wr = F * D * 4 * Bl
    print(f"pooled_exchange_unpack W={W} B_local={Bl}: {us:.1f} us  ({(rd + wr) / us / 1e3:.0f} GB/s, {(rd + wr) / 1e6:.0f} MB)")
    us = timeit(lambda: torch.ops.tbe_hip.pooled_exchange_pack(grad, feat_out_col, fsrc, fcol, slab_off, slab_stride,
                                                               W * Bl * D_loc, True, 0.125))
    print(f"pooled_exchange_pack   W={W} B_local={Bl}: {us:.1f} us  ({(rd + wr) / us / 1e3:.0f} GB/s)")


if __name__ == "__main__":
    main()
