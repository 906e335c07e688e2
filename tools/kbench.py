"""Kernel-level timing of the TBE hot path at the Criteo-1TB shape (development tool;
bench.py is the judged harness).  Usage: python tools/kbench.py [--batch 65536] [--cap ROWS]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import _paths  # noqa: E402,F401
from fbgemm_gpu.split_embedding_configs import EmbOptimType  # noqa: E402
from fbgemm_gpu.split_table_batched_embeddings_ops import (  # noqa: E402
    ComputeDevice, EmbeddingLocation, SplitTableBatchedEmbeddingBagsCodegen)

CRITEO_ROWS = [45833188, 36746, 17245, 7413, 20243, 3, 7114, 1441, 62, 29275261, 1572176, 345138, 10, 2209,
               11267, 128, 4, 974, 14, 48937457, 11316796, 40094537, 452104, 12606, 104, 35]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--cap", type=int, default=0, help="cap rows per table (0 = full 85 GiB)")
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--opt", default="EXACT_SGD")
    ap.add_argument("--nbatches", type=int, default=8)
    ap.add_argument("--pooling", type=int, default=1, help="ids per bag (fixed pooling factor)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    rows = [min(r, args.cap) if args.cap else r for r in CRITEO_ROWS]
    D, B, F = args.dim, args.batch, len(rows)
    t0 = time.time()
    mod = SplitTableBatchedEmbeddingBagsCodegen(
        [(r, D, EmbeddingLocation.DEVICE, ComputeDevice.CUDA) for r in rows], device=dev,
        optimizer=getattr(EmbOptimType, args.opt), learning_rate=0.01)
    for w, r in zip(mod.split_embedding_weights(), rows):
        w.uniform_(-(1.0 / r) ** 0.5, (1.0 / r) ** 0.5)
    torch.cuda.synchronize()
    print(f"tables: {sum(rows)} rows, {sum(rows) * D * 4 / 2**30:.1f} GiB, built in {time.time() - t0:.1f}s", flush=True)
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    batches = []
    for _ in range(args.nbatches):
        idx = torch.cat([torch.randint(0, r, (B * args.pooling,), generator=g, device=dev, dtype=torch.int64) for r in rows])
        batches.append(idx)
    L = args.pooling
    offsets = torch.arange(F * B + 1, dtype=torch.int64, device=dev) * L
    grad = torch.randn(B, F * D, device=dev)

    def timeit(fn, n):
        for i in range(3):
            fn(i)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(n):
            fn(i)
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / n  # ms

    outs = {}

    def fwd(i):
        with torch.no_grad():  # forward only: do not start the backward's side-stream sort
            outs["o"] = mod(batches[i % len(batches)], offsets)

    ms_f = timeit(fwd, args.iters)
    fwd_bytes = B * (F * L * (D * 4 + 8) + F * 8 + F * D * 4)
    print(f"fwd: {ms_f * 1e3:.1f} us  {fwd_bytes / ms_f / 1e6:.1f} GB/s (algorithmic {fwd_bytes / 1e6:.1f} MB)", flush=True)

    def fwdbwd(i):
        o = mod(batches[i % len(batches)], offsets)
        o.backward(grad)

    import ctypes

    from fbgemm_gpu import _lib
    lib = _lib.load()
    for i in range(3):
        fwdbwd(i)
    torch.cuda.synchronize()
    lib.tbe_profile_enable(1)
    tot, n = ctypes.c_double(0.0), ctypes.c_int64(0)
    for slot in range(4):
        lib.tbe_profile_read(slot, ctypes.byref(tot), ctypes.byref(n))
    for i in range(args.iters):
        fwdbwd(i)
    torch.cuda.synchronize()
    names = ["fwd kernel", "bwd_update kernel", "bwd apply (update+fixup)", "bwd prepare (linearize+sort)"]
    for slot in range(4):
        lib.tbe_profile_read(slot, ctypes.byref(tot), ctypes.byref(n))
        if n.value:
            print(f"  [events] {names[slot]}: {tot.value / n.value * 1e3:.1f} us avg over {n.value}", flush=True)
    lib.tbe_profile_enable(0)
    ms_fb = timeit(fwdbwd, args.iters)
    bwd_bytes = B * (F * D * 4 + F * 16 + 2 * F * D * 4)
    ms_b = ms_fb - ms_f
    print(f"fwd+bwd: {ms_fb * 1e3:.1f} us ; bwd ~ {ms_b * 1e3:.1f} us  {bwd_bytes / ms_b / 1e6:.1f} GB/s "
          f"(algorithmic {bwd_bytes / 1e6:.1f} MB)", flush=True)
    print(f"train-step TBE samples/s: {B / ms_fb * 1e3:.3e}", flush=True)


if __name__ == "__main__":
    main()
