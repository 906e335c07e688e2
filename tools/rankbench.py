"""What ONE rank of an N-rank run of BASELINE config 3 executes on the embedding side, on one GPU, at its real shape.

The one-GPU rehearsals of bench.py put all 26 features on one rank at the per-rank batch; a real rank r of the pinned
plan (the 4 largest tables row-wise, the rest table-wise / replicated) holds ITS table-wise tables, a 1/W
row window of the 4 row-wise tables and the replicated tiny tables, and receives B_global ids per held feature
(B_local from each of the W sources).  This tool builds exactly that — through the product's
ShardedEmbeddingBagCollection with ShardingEnv.from_local(W, r), so the TBE carries the all-to-all output layout and the
row windows of a real rank — feeds it synthetic "received" ids (no collective runs: what is measured is every KERNEL the
exchange brackets), and times per step:

  lookup (tbe_fwd_*), backward prepare (linearize + sort), update + fix-up, pooled_exchange unpack / pack, the
  replicated tables' dense-gradient lookup + backward, [bucketized mode: block_bucketize on the sender side]

with algorithmic bytes next to each (zero rows of foreign ids named separately).  `--mode bucketized` feeds the
row-wise features what a bucketized input dist delivers instead (only the ids of this rank's row block, as LOCAL rows,
empty bags elsewhere; torchrec/distributed/embedding_sharding.py:121-184, sharding/rw_sharding.py:229-236).

Under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) the per-kernel HBM bytes come from
tools/pmc_summary.py.  Usage: python tools/rankbench.py --world 8 --rank 0 [--mode windows|bucketized] [--json out.json]
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torchrec-oldfork_amd"))

from fbgemm_gpu import _lib  # noqa: E402
from torchrec_amd.datasets.random import CRITEO_1TB_ROWS, DEFAULT_CAT_NAMES  # noqa: E402
from torchrec_amd.distributed.embeddingbag import ShardedEmbeddingBagCollection  # noqa: E402
from torchrec_amd.distributed.planner import EmbeddingShardingPlanner, Topology, rw_block_size  # noqa: E402
from torchrec_amd.distributed.types import ShardingEnv  # noqa: E402
from torchrec_amd.modules.embedding_configs import EmbeddingBagConfig  # noqa: E402
from torchrec_amd.modules.embedding_modules import EmbeddingBagCollection  # noqa: E402

D = 128


def events(fn, iters, warm=3):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--global-batch", type=int, default=65536)
    ap.add_argument("--row-wise", type=int, default=4)
    ap.add_argument("--mode", choices=["windows", "bucketized"], default="windows")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--nbatches", type=int, default=8)
    ap.add_argument("--row-cap", type=int, default=0)
    ap.add_argument("--pooling", type=int, default=1, help="ids per bag (fixed pooling factor)")
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    W, me, Bg, L = args.world, args.rank, args.global_batch, args.pooling
    Bl = Bg // W
    dev = torch.device("cuda", 0)
    rows = [min(r, args.row_cap) if args.row_cap else r for r in CRITEO_1TB_ROWS]
    tables = [EmbeddingBagConfig(name=f"t_{n}", embedding_dim=D, num_embeddings=rows[i], feature_names=[n])
              for i, n in enumerate(DEFAULT_CAT_NAMES)]
    ebc = EmbeddingBagCollection(tables=tables, device=torch.device("meta"))
    plan = EmbeddingShardingPlanner(Topology(W), num_row_wise=args.row_wise or None).plan_tables(tables)
    t0 = time.time()
    sebc = ShardedEmbeddingBagCollection(ebc, plan, ShardingEnv.from_local(W, me), {"learning_rate": 0.01}, dev)
    torch.cuda.synchronize()
    kind = sebc._table_kind
    feats = sebc._local_feats[me]  # global feature numbers this rank's fused lookup holds, row-wise first
    Fl = len(feats)
    n_rw = sum(1 for g in feats if kind[g] == -1)
    n_tw = Fl - n_rw
    n_dp = len(sebc._dp_feats)
    held_gib = sum(w.numel() for w, _ in sebc.local_shards().values()) * 4 / 2**30
    print(f"rank {me}/{W}: {n_tw} table-wise + {n_rw} row-wise shards ({held_gib:.1f} GiB) + {n_dp} replicated tables, "
          f"B_local {Bl}; built in {time.time() - t0:.1f} s", flush=True)
    mod, dpm = sebc._emb_module, sebc._dp_module
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + me)

    # ---- what the id all-to-all delivers: [src rank][local feature][sample] ----------------------------------------
    batches = []
    for _ in range(args.nbatches):
        if args.mode == "windows":
            v = torch.stack([torch.stack([torch.randint(0, rows[g], (Bl * L,), generator=gen, device=dev) for g in feats])
                             for _ in range(W)]).reshape(-1)
            off = torch.arange(W * Fl * Bl + 1, dtype=torch.int64, device=dev) * L
            batches.append((v, off))
        else:
            vals, lens = [], []
            for _ in range(W):
                for g in feats:
                    ids = torch.randint(0, rows[g], (Bl * L,), generator=gen, device=dev)
                    if kind[g] == -1:
                        blk = rw_block_size(rows[g], W)
                        mine = (ids // blk) == me
                        vals.append(ids[mine] - me * blk)  # bag-major order is kept: stable inside a bag
                        lens.append(mine.view(Bl, L).sum(dim=1).to(torch.int64))
                    else:
                        vals.append(ids)
                        lens.append(torch.full((Bl,), L, dtype=torch.int64, device=dev))
            lens = torch.cat(lens)
            off = torch.zeros(lens.numel() + 1, dtype=torch.int64, device=dev)
            off[1:] = torch.cumsum(lens, 0)
            batches.append((torch.cat(vals), off))
    if args.mode == "bucketized":
        mod.set_row_windows(None)  # ids arrive as local rows of the shard
    n_ids = [int(b[0].numel()) for b in batches]
    grad = torch.randn(W * Bl, sebc._D_local, device=dev)
    lib = _lib.load()

    def read(slot):
        tot, n = ctypes.c_double(0.0), ctypes.c_int64(0)
        lib.tbe_profile_read(slot, ctypes.byref(tot), ctypes.byref(n))
        return (tot.value / n.value * 1e3) if n.value else 0.0

    def fwdbwd(i):
        v, off = batches[i % len(batches)]
        out, rec = mod.lookup_no_autograd(v, off)
        mod.backward_no_autograd(rec, grad)

    for i in range(3):
        fwdbwd(i)
    torch.cuda.synchronize()
    lib.tbe_profile_enable(1)
    for s in range(4):
        read(s)
    for i in range(args.iters):
        fwdbwd(i)
    torch.cuda.synchronize()
    us = {"tbe_fwd_kernel": read(0), "bwd_update_kernel": read(1), "bwd_apply (update + fix-up)": read(2),
          "bwd_prepare (linearize + sort, side stream)": read(3)}
    rows_upd = ctypes.c_int64(0)
    lib.tbe_profile_read_rows(ctypes.byref(rows_upd))
    U = rows_upd.value / args.iters
    lib.tbe_profile_enable(0)
    us["fwd + bwd, back to back"] = events(fwdbwd, args.iters)

    # ---- exchange kernels on this rank's receive / send buffers ----------------------------------------------------
    lay = sebc._exchange_layout(Bl)
    recv = torch.randn(lay["recv_numel"], device=dev)
    gout = torch.randn(Bl, sebc._D_total, device=dev)

    def unpack(i):
        torch.ops.tbe_hip.pooled_exchange_unpack(recv, lay["feat_out_col"], lay["feat_src"], lay["feat_slab_col"],
                                                 lay["slab_offset"], lay["slab_stride"], Bl, sebc._D_total, sebc._vec_ok, 1.0)

    def pack(i):
        torch.ops.tbe_hip.pooled_exchange_pack(gout, lay["feat_out_col"], lay["feat_src"], lay["feat_slab_col"],
                                               lay["slab_offset"], lay["slab_stride"], lay["recv_numel"], sebc._vec_ok, 1.0 / W)

    us["pooled_exchange_unpack"] = events(unpack, args.iters)
    us["pooled_exchange_pack"] = events(pack, args.iters)

    # ---- replicated tables: dense-gradient lookup over the LOCAL batch ---------------------------------------------
    if dpm is not None:
        dp_rows = [rows[g] for g in sebc._dp_feats]
        dpv = [torch.cat([torch.randint(0, r, (Bl,), generator=gen, device=dev) for r in dp_rows]) for _ in range(4)]
        dpo = torch.arange(n_dp * Bl + 1, dtype=torch.int64, device=dev)
        buf = torch.empty(Bl, sebc._D_total, device=dev)

        def dp_step(i):
            _, rec = dpm.lookup_no_autograd(dpv[i % 4], dpo, None, into=(buf, sebc._dp_out_off, sebc._D_total))
            dpm.backward_no_autograd(rec, gout)

        us["replicated tables: lookup + dense-gradient backward"] = events(dp_step, args.iters)

    # ---- sender side of the bucketized mode: block_bucketize of the row-wise features' LOCAL ids -------------------
    n_rw_all = sum(1 for k in kind if k == -1)
    if n_rw_all:
        rw_tabs = [t for t, k in enumerate(kind) if k == -1]
        ids = torch.cat([torch.randint(0, rows[t], (Bl * L,), generator=gen, device=dev) for t in rw_tabs])
        lens = torch.full((n_rw_all * Bl,), L, dtype=torch.int32, device=dev)
        blocks = torch.tensor([rw_block_size(rows[t], W) for t in rw_tabs], dtype=torch.int64, device=dev)

        def bucketize(i):
            torch.ops.fbgemm.block_bucketize_sparse_features(lengths=lens, indices=ids, bucketize_pos=False, sequence=False,
                                                             block_sizes=blocks, my_size=W, weights=None)

        us["sender: block_bucketize of the row-wise features (bucketized mode only)"] = events(bucketize, args.iters)

    # ---- algorithmic bytes (SURVEY.md §8d terms, per step on this rank) -------------------------------------------
    N = sum(n_ids) / len(n_ids)
    bags = W * Fl * Bl
    local_rw = n_rw * Bg * L / W  # ids of row-wise features that fall into this rank's window (expected)
    rows_read = n_tw * Bg * L + local_rw
    nonzero_out = n_tw * Bg + n_rw * Bg * (1.0 - (1.0 - 1.0 / W) ** L)  # bags with at least one local id
    zero_rows = (n_tw + n_rw) * Bg - nonzero_out  # output rows written as zeros (bags without a local id)
    fwd_useful = rows_read * D * 4 + N * 8 + bags * 8 + nonzero_out * D * 4
    fwd_zero_write = zero_rows * D * 4
    bwd_useful = rows_read * D * 4 + N * 8 + bags * 8 + U * 2 * D * 4  # one gradient row read per local id
    out = {
        "world": W, "rank": me, "mode": args.mode, "global_batch": Bg, "local_batch": Bl, "pooling_factor": L,
        "held": {"table_wise": n_tw, "row_wise_shards": n_rw, "replicated": n_dp, "GiB": round(held_gib, 2)},
        "ids_per_step": N, "bags_per_step": bags, "distinct_rows_updated_per_step": U,
        "us_per_step": {k: round(v, 1) for k, v in us.items()},
        "algorithmic_MB": {"forward useful (rows + ids + offsets + non-zero output)": round(fwd_useful / 1e6, 2),
                           "forward zero rows written (foreign row-wise ids)": round(fwd_zero_write / 1e6, 2),
                           "backward (grad rows + ids + offsets + 2 x distinct rows)": round(bwd_useful / 1e6, 2),
                           "exchange unpack (read slabs + write [B_local, sum D])": round(
                               (lay["recv_numel"] + Bl * (sebc._D_total - n_dp * D)) * 4 / 1e6, 2),
                           "exchange pack": round((lay["recv_numel"] + Bl * (sebc._D_total - n_dp * D)) * 4 / 1e6, 2)},
        "GBps": {"tbe_fwd_kernel (useful bytes)": round(fwd_useful / max(us["tbe_fwd_kernel"], 1e-9) / 1e3, 1),
                 "tbe_fwd_kernel (incl. zero rows)": round((fwd_useful + fwd_zero_write) / max(us["tbe_fwd_kernel"], 1e-9) / 1e3, 1),
                 "bwd_update_kernel": round(bwd_useful / max(us["bwd_update_kernel"], 1e-9) / 1e3, 1)},
    }
    print(json.dumps(out, indent=1), flush=True)
    if args.json:
        os.makedirs(os.path.dirname(os.path.abspath(args.json)), exist_ok=True)
        json.dump(out, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
