"""Checkpoint surface of the sharded collections and the fused optimizer (reference layout:
torchrec/distributed/embeddingbag.py:405-416 `embedding_bags.<table>.weight` for every table a rank holds,
batched_embedding_kernel.py:241-249 optimizer state nested under the parameter key, optim/keyed.py:69-186 keyed
state_dict / in-place load_state_dict): save -> perturb -> load -> identical.  CPU: two gloo ranks (table-wise +
row-wise + replicated tables, oracle compute); GPU: the real kernels with row-wise Adagrad and a table behind the HBM
row cache (state read through the cache write-back)."""
import copy
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _paths  # noqa: F401
from _results import ResultStore
from test_sharded_gloo import _free_port

ROWS = [40, 7, 23, 90, 5]
DIMS = [8, 8, 8, 8, 8]


def _local(v):
    """The rank's own tensor of a state value: a sharded table / state comes as a torch ShardedTensor when a process
    group exists (as in the reference), everything else as a plain tensor."""
    from torchrec_amd.distributed.embeddingbag import unwrap_local

    return unwrap_local(v)


def _clone(sd):
    return {k: _local(v).detach().clone() for k, v in sd.items()}


def _clone_opt(osd):
    return {"state": {k: {kk: _local(vv).detach().clone() for kk, vv in v.items()} for k, v in osd["state"].items()}}


def _cpu_worker(rank, W, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    try:
        import _cpu_ops
        _cpu_ops.register()
        from _oracle_tbe import oracle_dp_tbe_factory, oracle_tbe_factory
        from fbgemm_gpu.split_embedding_configs import EmbOptimType
        from torch import nn
        from torchrec_amd.distributed.embeddingbag import EmbeddingBagCollectionSharder
        from torchrec_amd.distributed.model_parallel import DistributedModelParallel
        from torchrec_amd.distributed.planner import EmbeddingShardingPlanner, Topology
        from torchrec_amd.distributed.types import ShardingEnv
        from torchrec_amd.modules.embedding_configs import EmbeddingBagConfig
        from torchrec_amd.modules.embedding_modules import EmbeddingBagCollection
        from torchrec_amd.sparse.jagged_tensor import KeyedJaggedTensor

        class Holder(nn.Module):
            def __init__(self, ebc):
                super().__init__()
                self.sparse = ebc
                self.head = nn.Linear(sum(DIMS), 1)

            def forward(self, kjt):
                return self.head(self.sparse(kjt).wait().values())

        torch.manual_seed(0)
        keys = [f"f{i}" for i in range(len(ROWS))]
        tables = [EmbeddingBagConfig(name=f"t{i}", embedding_dim=DIMS[i], num_embeddings=ROWS[i], feature_names=[keys[i]])
                  for i in range(len(ROWS))]
        dmp = DistributedModelParallel(
            Holder(EmbeddingBagCollection(tables, device=torch.device("meta"))), env=ShardingEnv.from_process_group(dist.group.WORLD),
            device=torch.device("cpu"),
            sharders=[EmbeddingBagCollectionSharder({"learning_rate": 0.1, "optimizer": EmbOptimType.EXACT_ROWWISE_ADAGRAD},
                                                    tbe_factory=oracle_tbe_factory, dp_tbe_factory=oracle_dp_tbe_factory)],
            planner=EmbeddingShardingPlanner(Topology(W, "cpu"), num_row_wise=2, dp_max_rows=10))
        kinds = {n: p.sharding_type for n, p in dmp.plan.plan["sparse"].items()}
        rng = np.random.default_rng(7 + rank)
        kjt = KeyedJaggedTensor.from_fixed_lengths(keys, torch.from_numpy(np.concatenate(
            [rng.integers(0, ROWS[f], size=6) for f in range(len(ROWS))]).astype(np.int64)), [1] * len(keys))
        dmp(kjt).sum().backward()  # moves the fused tables (oracle SGD) and leaves a gradient on the replicated ones
        sd = dmp.state_dict()
        fo = dmp.fused_optimizer
        for st in fo.state_dict()["state"].values():  # give the state non-trivial values
            for v in st.values():
                _local(v).copy_(torch.rand_like(_local(v)))
        saved, saved_opt = _clone(sd), _clone_opt(fo.state_dict())
        # perturb everything the checkpoint covers
        with torch.no_grad():
            for v in dmp.state_dict().values():
                _local(v).add_(1.0)
            for st in fo.state_dict()["state"].values():
                for v in st.values():
                    _local(v).mul_(3.0)
        missing, unexpected = dmp.load_state_dict(saved)
        fo.load_state_dict(saved_opt)
        after, after_opt = dmp.state_dict(), fo.state_dict()
        ok = (not missing and not unexpected and set(after) == set(saved) and all(torch.equal(_local(after[k]), saved[k]) for k in saved)
              and all(torch.equal(_local(after_opt["state"][k][kk]), vv) for k, v in saved_opt["state"].items() for kk, vv in v.items()))
        # loading the ShardedTensor objects themselves works too (what a reference-written checkpoint holds per rank)
        dmp.load_state_dict(dmp.state_dict())
        fo.load_state_dict(fo.state_dict())
        from torch.distributed._shard.sharded_tensor import ShardedTensor
        sharded_meta = {k: {"size": list(v.size()), "shards": [(list(m.shard_offsets), list(m.shard_sizes), m.placement.rank())
                                                               for m in v.metadata().shards_metadata]}
                        for k, v in after.items() if isinstance(v, ShardedTensor)}
        opt_sharded = {kk: list(vv.size()) for v in after_opt["state"].values() for kk, vv in v.items() if isinstance(vv, ShardedTensor)}
        ret[rank] = {"ok": ok, "keys": sorted(saved.keys()), "kinds": kinds, "opt_keys": {k: sorted(v) for k, v in saved_opt["state"].items()},
                     "shapes": {k: list(v.shape) for k, v in saved.items()}, "sharded_meta": sharded_meta, "opt_sharded": opt_sharded}
    finally:
        dist.destroy_process_group()


def test_save_perturb_load_two_ranks_cpu():
    W = 2
    ret = ResultStore()
    mp.spawn(_cpu_worker, args=(W, _free_port(), ret), nprocs=W, join=True)
    for r in range(W):
        got = ret[r]
        assert got["ok"]
        kinds = got["kinds"]
        assert sorted(kinds.values()).count("row_wise") == 2 and "data_parallel" in kinds.values() and "table_wise" in kinds.values()
        for t, kind in kinds.items():
            key = f"sparse.embedding_bags.{t}.weight"
            rows = ROWS[int(t[1:])]
            if kind == "data_parallel":
                assert got["shapes"][key] == [rows, 8]          # replicated tables are saved whole, on every rank
            elif kind == "row_wise":
                assert got["shapes"][key] == [(rows + 1) // 2 if r == 0 else rows - (rows + 1) // 2, 8]
        assert "head.weight" in got["keys"] and "head.bias" in got["keys"]
        fused = [t for t, k in kinds.items() if k != "data_parallel" and f"sparse.embedding_bags.{t}.weight" in got["keys"]]
        assert got["opt_keys"] == {f"sparse.embedding_bags.{t}.weight": [f"{t}.momentum1"] for t in fused}
        # sharded tables (and their row-wise optimizer state) are ShardedTensors carrying the GLOBAL layout
        # (embedding_kernel.py:63-122, batched_embedding_kernel.py:166-246); replicated tables are plain tensors
        for t in fused:
            m = got["sharded_meta"][f"sparse.embedding_bags.{t}.weight"]
            rows = ROWS[int(t[1:])]
            assert m["size"] == [rows, 8]
            if kinds[t] == "row_wise":
                assert m["shards"] == [([0, 0], [(rows + 1) // 2, 8], 0), ([(rows + 1) // 2, 0], [rows - (rows + 1) // 2, 8], 1)]
            else:
                assert len(m["shards"]) == 1 and m["shards"][0][:2] == ([0, 0], [rows, 8]) and m["shards"][0][2] == r
            assert got["opt_sharded"][f"{t}.momentum1"] == [rows]
        assert not any(f".{t}." in k for k in got["sharded_meta"] for t, kk in kinds.items() if kk == "data_parallel")


@pytest.mark.gpu
def test_save_perturb_load_with_row_cache_gpu():
    from fbgemm_gpu.split_embedding_configs import EmbOptimType
    from torchrec_amd.distributed.embeddingbag import ShardedEmbeddingBagCollection
    from torchrec_amd.distributed.planner import EmbeddingShardingPlanner, ParameterConstraints, Topology
    from torchrec_amd.distributed.types import ShardingEnv
    from torchrec_amd.modules.embedding_configs import EmbeddingBagConfig
    from torchrec_amd.modules.embedding_modules import EmbeddingBagCollection
    from torchrec_amd.sparse.jagged_tensor import KeyedJaggedTensor

    dev = torch.device("cuda", 0)
    rows, D, B = [3000, 50, 700], 64, 64
    keys = [f"f{i}" for i in range(3)]
    tables = [EmbeddingBagConfig(name=f"t{i}", embedding_dim=D, num_embeddings=rows[i], feature_names=[keys[i]]) for i in range(3)]
    cons = {"t0": ParameterConstraints(["table_wise"], ["batched_fused_uvm_caching"])}
    plan = EmbeddingShardingPlanner(Topology(1), constraints=cons).plan_tables(tables)
    sebc = ShardedEmbeddingBagCollection(EmbeddingBagCollection(tables, device=torch.device("meta")), plan, ShardingEnv.from_local(1, 0),
                                         {"learning_rate": 0.1, "optimizer": EmbOptimType.EXACT_ROWWISE_ADAGRAD, "eps": 1e-3, "cache_sets": 2},
                                         dev)
    assert sebc._emb_module._cache is not None
    rng = np.random.default_rng(3)

    def step():
        ids = np.concatenate([rng.integers(0, rows[f], size=B) for f in range(3)]).astype(np.int64)
        out = sebc(KeyedJaggedTensor.from_fixed_lengths(keys, torch.from_numpy(ids).to(dev), [1] * 3)).wait().values()
        out.backward(torch.from_numpy(rng.standard_normal(tuple(out.shape)).astype(np.float32)).to(dev))

    for _ in range(3):
        step()
    fo = sebc.fused_optimizer
    saved, saved_opt = _clone(sebc.state_dict()), _clone_opt(fo.state_dict())  # no process group: plain tensors
    assert sorted(saved) == ["embedding_bags.t0.weight", "embedding_bags.t1.weight", "embedding_bags.t2.weight"]
    assert {k: sorted(v) for k, v in saved_opt["state"].items()} == {f"embedding_bags.t{i}.weight": [f"t{i}.momentum1"] for i in range(3)}
    assert float(saved_opt["state"]["embedding_bags.t0.weight"]["t0.momentum1"].abs().sum()) > 0  # read through the cache write-back
    for _ in range(3):  # perturb by training on
        step()
    torch.cuda.synchronize()
    assert not torch.equal(sebc.state_dict()["embedding_bags.t0.weight"].cpu(), saved["embedding_bags.t0.weight"].cpu())
    holder = torch.nn.Module()
    holder.add_module("m", sebc)
    res = holder.load_state_dict({f"m.{k}": v for k, v in saved.items()})
    assert not res.missing_keys and not res.unexpected_keys
    fo.load_state_dict(saved_opt)
    after, after_opt = sebc.state_dict(), fo.state_dict()
    for k in saved:
        assert torch.equal(after[k].cpu(), saved[k].cpu())
    for k, v in saved_opt["state"].items():
        for kk, vv in v.items():
            assert torch.equal(after_opt["state"][k][kk].cpu(), vv.cpu())
    # and the restored module trains on: same ids and gradient twice from the same checkpoint give the same weights
    rng = np.random.default_rng(11)
    step()
    torch.cuda.synchronize()
    w1 = _clone(sebc.state_dict())
    holder.load_state_dict({f"m.{k}": v for k, v in saved.items()})
    fo.load_state_dict(saved_opt)
    rng = np.random.default_rng(11)
    step()
    torch.cuda.synchronize()
    w2 = sebc.state_dict()
    for k in w1:
        assert torch.equal(w1[k].cpu(), w2[k].cpu())
