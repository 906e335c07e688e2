"""Timing of the KJT index ops at the BASELINE shape (26 features x batch 65 536, pooling factor 1):
development tool, numbers quoted in DESIGN.md."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import _paths  # noqa: E402,F401
import fbgemm_gpu  # noqa: E402,F401


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
    F, B, W = 26, 65536, 8
    dev = "cuda"
    lengths = torch.ones(F * B, dtype=torch.int32, device=dev)
    values = torch.randint(0, 1 << 25, (F * B,), device=dev, dtype=torch.int64)
    perm = torch.randperm(F, device=dev).to(torch.int32)
    blocks = torch.full((F,), (1 << 25) // W, dtype=torch.int64, device=dev)
    us = timeit(lambda: torch.ops.fbgemm.asynchronous_complete_cumsum(lengths))
    print(f"asynchronous_complete_cumsum  n={F * B}: {us:.1f} us  ({F * B * 8 / us / 1e3:.1f} GB/s r+w)")
    us = timeit(lambda: torch.ops.fbgemm.permute_2D_sparse_data(perm, lengths.view(F, B), values, None, F * B))
    print(f"permute_2D_sparse_data        {F}x{B}: {us:.1f} us  ({F * B * 24 / us / 1e3:.1f} GB/s r+w incl. lengths)")
    us = timeit(lambda: torch.ops.fbgemm.block_bucketize_sparse_features(lengths, values, False, False, blocks, W, None))
    print(f"block_bucketize_sparse_features W={W}: {us:.1f} us")


if __name__ == "__main__":
    main()
