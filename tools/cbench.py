"""Host-offloaded tables (BASELINE config 4 shape, scaled to one GPU): forward + backward + fused row-wise
Adagrad of F features over host-resident tables, with EmbeddingLocation.MANAGED (every row over the
host link) vs MANAGED_CACHING (64-way HBM row cache, csrc/tbe_cache.hip) vs DEVICE, under
Zipf-distributed ids.  Development tool; numbers quoted in DESIGN.md.
Usage: python tools/cbench.py [--rows 20000000] [--features 4] [--batch 65536] [--zipf 1.05] [--load-factor 0.2]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import _paths  # noqa: E402,F401
from fbgemm_gpu.split_embedding_configs import EmbOptimType  # noqa: E402
from fbgemm_gpu.split_table_batched_embeddings_ops import (  # noqa: E402
    ComputeDevice, EmbeddingLocation, SplitTableBatchedEmbeddingBagsCodegen)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=20_000_000)
    ap.add_argument("--features", type=int, default=4)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--zipf", type=float, default=1.05)
    ap.add_argument("--load-factor", type=float, default=0.2)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--locations", default="DEVICE,MANAGED,MANAGED_CACHING")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    F, B, D, R = args.features, args.batch, args.dim, args.rows
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    a = 1.0 - args.zipf
    batches = []
    for _ in range(16):
        u = torch.rand(F * B, generator=g, device=dev, dtype=torch.float64)
        x = ((u * (float(R) ** a - 1.0) + 1.0) ** (1.0 / a)).floor().clamp_(1, R).to(torch.int64) - 1
        # scatter the hot ids over the table (real ids are hashed, not sorted by popularity)
        batches.append((x * 2654435761) % R)
    offsets = torch.arange(F * B + 1, dtype=torch.int64, device=dev)
    grad = torch.randn(B, F * D, device=dev)
    uniq = sum(int(torch.unique(b).numel()) for b in batches) / len(batches)
    print(f"{F} features x {R} rows x {D} (fp32: {F * R * D * 4 / 2**30:.1f} GiB), batch {B}, zipf {args.zipf}: "
          f"{uniq:.0f} distinct rows of {F * B} ids per batch", flush=True)
    for name in args.locations.split(","):
        loc = getattr(EmbeddingLocation, name)
        t0 = time.time()
        mod = SplitTableBatchedEmbeddingBagsCodegen(
            [(R, D, loc, ComputeDevice.CUDA) for _ in range(F)], device=dev, optimizer=EmbOptimType.EXACT_ROWWISE_ADAGRAD,
            learning_rate=0.01, cache_load_factor=args.load_factor)
        build = time.time() - t0

        def step(i):
            out = mod(batches[i % len(batches)], offsets)
            out.backward(grad)

        for i in range(len(batches) + 4):  # warm the cache over the whole batch pool
            step(i)
        torch.cuda.synchronize()
        if mod.cache_stats() is not None:
            mod._cache.counters.zero_()
        t0 = time.perf_counter()
        for i in range(args.iters):
            step(i)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.iters * 1e3
        st = mod.cache_stats()
        extra = ""
        if st is not None:
            extra = (f"  hit rate {st['hits'] / max(1, st['hits'] + st['misses']):.3f}, evictions/step "
                     f"{st['evictions'] / args.iters:.0f}, staged last batch {st['staged_last_batch']}, slots {st['slots']}")
        print(f"{name:16s} {ms:8.3f} ms/step  {F * B / ms / 1e3:8.2f} M lookups/s  (built in {build:.1f}s){extra}", flush=True)
        del mod
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
