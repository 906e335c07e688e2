"""CPU, world_size 2 (gloo): the distributed host logic of the N > 1 path — planner, id input
dist, a2a-ready TBE layout, ONE pooled all-to-all with table-wise copy + row-wise reduce, gradient
exchange with 1/W division — against the unsharded oracle.  Follows the reference's core test
pattern (sharded-vs-unsharded equivalence after one train step,
torchrec/distributed/test_utils/test_model_parallel_base.py:148-294); compute is the oracle's
(tests/_oracle_tbe.py) because the product has no CPU kernels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _paths  # noqa: F401


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


ROWS = [40, 7, 23, 90, 5]
DIMS = [8, 8, 8, 8, 8]
LR = 0.25


def _global_data(W, B_local, fixed_len, weighted, seed=3):
    rng = np.random.default_rng(seed)
    F = len(ROWS)
    per_rank = []
    for r in range(W):
        lengths = (np.full(F * B_local, fixed_len) if fixed_len else rng.integers(0, 4, size=F * B_local)).astype(np.int32)
        vals = np.concatenate([rng.integers(0, ROWS[f], size=int(lengths[f * B_local:(f + 1) * B_local].sum()))
                               for f in range(F)]).astype(np.int64)
        wts = (rng.random(vals.size).astype(np.float32) + 0.5) if weighted else None
        grad = rng.standard_normal((B_local, sum(DIMS))).astype(np.float32)
        per_rank.append((lengths, vals, wts, grad))
    init = [rng.standard_normal((r, d)).astype(np.float32) for r, d in zip(ROWS, DIMS)]
    return per_rank, init


def _worker(rank, W, port, fixed_len, weighted, n_rw, ret, dp_max_rows=0, mean=False, rw_mode=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    try:
        import _cpu_ops
        _cpu_ops.register()
        from _oracle_tbe import oracle_dp_tbe_factory, oracle_tbe_factory
        from torchrec_amd.distributed.embeddingbag import ShardedEmbeddingBagCollection
        from torchrec_amd.distributed.planner import EmbeddingShardingPlanner, Topology
        from torchrec_amd.distributed.types import ShardingEnv
        from torchrec_amd.modules.embedding_configs import EmbeddingBagConfig
        from torchrec_amd.modules.embedding_modules import EmbeddingBagCollection
        from torchrec_amd.sparse.jagged_tensor import KeyedJaggedTensor

        B_local = 6
        per_rank, init = _global_data(W, B_local, fixed_len, weighted)
        keys = [f"f{i}" for i in range(len(ROWS))]
        from torchrec_amd.modules.embedding_configs import PoolingType
        # mean: False = all SUM, True = all MEAN, "mixed" = odd tables MEAN (both pooling types in ONE collection)
        is_mean = [(i % 2 == 1) if mean == "mixed" else bool(mean) for i in range(len(ROWS))]
        tables = [EmbeddingBagConfig(name=f"t{i}", embedding_dim=DIMS[i], num_embeddings=ROWS[i], feature_names=[keys[i]],
                                     pooling=PoolingType.MEAN if is_mean[i] else PoolingType.SUM)
                  for i in range(len(ROWS))]
        ebc = EmbeddingBagCollection(tables, is_weighted=weighted, device=torch.device("meta"))
        plan = EmbeddingShardingPlanner(Topology(W, "cpu"), num_row_wise=n_rw, dp_max_rows=dp_max_rows).plan_tables(tables)
        env = ShardingEnv.from_process_group(dist.group.WORLD)
        sebc = ShardedEmbeddingBagCollection(ebc, plan, env, {"learning_rate": LR}, torch.device("cpu"),
                                             tbe_factory=oracle_tbe_factory, dp_tbe_factory=oracle_dp_tbe_factory,
                                             rw_input_dist=rw_mode)
        # load the global initial weights into the local shards / replicas
        for name, (w, row0) in sebc.local_shards().items():
            t = int(name[1:])
            w.copy_(torch.from_numpy(init[t][row0:row0 + w.shape[0]]))
        with torch.no_grad():
            for name, w in sebc.dp_tables().items():
                w.copy_(torch.from_numpy(init[int(name[1:])]))
        lengths, vals, wts, grad = per_rank[rank]
        if fixed_len:
            kjt = KeyedJaggedTensor.from_fixed_lengths(keys, torch.from_numpy(vals), [fixed_len] * len(keys),
                                                       weights=torch.from_numpy(wts) if weighted else None)
        else:
            kjt = KeyedJaggedTensor.from_lengths_sync(keys, torch.from_numpy(vals), torch.from_numpy(lengths),
                                                      weights=torch.from_numpy(wts) if weighted else None)
        out = sebc(kjt).wait()
        assert out.keys() == keys
        vals_out = out.values()
        vals_out.backward(torch.from_numpy(grad))
        shards = {n: (w.clone().numpy(), r0) for n, (w, r0) in sebc.local_shards().items()}
        # replicated tables: what DDP + a dense SGD would do = all-reduce(mean) of the dense grad, then step
        if sebc._dp_module is not None:
            g = sebc._dp_module.weights.grad.clone()
            dist.all_reduce(g)
            g /= W
            with torch.no_grad():
                sebc._dp_module.weights -= LR * g
            for n, w in sebc.dp_tables().items():
                shards[n] = (w.clone().numpy(), 0)
        ret[rank] = (vals_out.detach().numpy().copy(), shards, {n: p.sharding_type for n, p in plan.items()},
                     sebc._rw_mode_active)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fixed_len,weighted,n_rw", [(1, False, 2), (1, True, 5), (0, False, 1), (0, True, 3), (4, False, 2),
                                                     (6, True, 1)])
@pytest.mark.parametrize("rw_mode", ["windows", "bucketize", "auto"])
def test_row_wise_input_dist_modes_world2(fixed_len, weighted, n_rw, rw_mode):
    """the bucketized row-wise input dist of the POOLED path (block_bucketize + lengths / ids
    exchange, embedding_sharding.py:121-184) next to the row-window one; both must give the unsharded result.  auto:
    windows for a host-known pooling factor <= 2, bucketize for longer or data-dependent bags."""
    ret = test_sharded_equals_unsharded_world2(fixed_len, weighted, n_rw, 10 if n_rw <= 3 else 0, False, rw_mode=rw_mode)
    want = rw_mode if rw_mode != "auto" else ("windows" if 0 < fixed_len <= 2 else "bucketize")
    assert ret[0][3] == want and ret[1][3] == want


def test_bucketized_input_dist_refuses_mean_pooled_row_wise_tables():
    import _cpu_ops
    _cpu_ops.register()
    from _oracle_tbe import oracle_tbe_factory
    from torchrec_amd.distributed.embeddingbag import ShardedEmbeddingBagCollection
    from torchrec_amd.distributed.planner import EmbeddingShardingPlanner, Topology
    from torchrec_amd.distributed.types import ShardingEnv
    from torchrec_amd.modules.embedding_configs import EmbeddingBagConfig, PoolingType
    from torchrec_amd.modules.embedding_modules import EmbeddingBagCollection

    tables = [EmbeddingBagConfig(name="t0", embedding_dim=8, num_embeddings=40, feature_names=["f0"], pooling=PoolingType.MEAN)]
    plan = EmbeddingShardingPlanner(Topology(2, "cpu"), num_row_wise=1, dp_max_rows=0).plan_tables(tables)
    assert plan["t0"].sharding_type == "row_wise"
    ebc = EmbeddingBagCollection(tables, device=torch.device("meta"))
    with pytest.raises(NotImplementedError, match="MEAN"):
        ShardedEmbeddingBagCollection(ebc, plan, ShardingEnv.from_local(2, 0), {}, torch.device("cpu"),
                                      tbe_factory=oracle_tbe_factory, rw_input_dist="bucketize")
    auto = ShardedEmbeddingBagCollection(ebc, plan, ShardingEnv.from_local(2, 0), {}, torch.device("cpu"),
                                         tbe_factory=oracle_tbe_factory, rw_input_dist="auto")
    assert auto._rw_mean is True  # auto keeps such collections on row windows


@pytest.mark.parametrize("fixed_len,weighted,n_rw,dp_max_rows,mean", [
    (1, False, 1, 0, False), (2, True, 2, 0, False), (0, False, 1, 0, False), (0, True, 0, 0, False),
    (1, False, 5, 0, False), (1, False, 0, 10, False), (0, True, 1, 25, False), (2, False, 0, 100, False),
    (0, False, 2, 10, True), (3, False, 5, 0, True), (0, False, 2, 10, "mixed"), (0, True, 1, 0, "mixed"), (2, False, 0, 25, "mixed")])
def test_sharded_equals_unsharded_world2(fixed_len, weighted, n_rw, dp_max_rows, mean, rw_mode=None):
    """mean=True: MEAN pooling over row-wise shards — every rank divides its partial sum by the FULL bag length
    (all ids travel to every rank, rows outside its block are masked), so the partial pools still add up."""
    from oracle import oracle

    from _util import oracle_backward_mixed, oracle_forward_mixed

    W = 2
    feat_mean = [(i % 2 == 1) if mean == "mixed" else bool(mean) for i in range(len(ROWS))]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(W, _free_port(), fixed_len, weighted, n_rw, ret, dp_max_rows, mean, rw_mode), nprocs=W, join=True)
    per_rank, init = _global_data(W, 6, fixed_len, weighted)
    # unsharded oracle on each rank's batch (forward), then ONE backward over the global batch with
    # grads / W (GRADIENT_DIVISION, comm_ops.py:527-528)
    tabs = oracle.Tables(ROWS, DIMS)
    for t in range(len(ROWS)):
        tabs.weights[t][...] = init[t]
    F = len(ROWS)
    for r in range(W):
        lengths, vals, wts, grad = per_rank[r]
        offs = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
        ref = oracle_forward_mixed(tabs, vals, offs, wts, feat_mean)
        np.testing.assert_allclose(ret[r][0], ref, rtol=1e-5, atol=1e-5)
    kinds = ret[0][2]
    assert sum(1 for k in kinds.values() if k == "row_wise") == n_rw
    n_dp = sum(1 for r in ROWS if r <= dp_max_rows)
    assert sum(1 for k in kinds.values() if k == "data_parallel") == n_dp
    # global batch = rank-major concatenation per feature
    B = 6
    g_len = np.concatenate([np.concatenate([per_rank[r][0][f * B:(f + 1) * B] for r in range(W)]) for f in range(F)])
    pos = [np.concatenate([[0], np.cumsum(per_rank[r][0])]) for r in range(W)]
    g_vals = np.concatenate([np.concatenate([per_rank[r][1][pos[r][f * B]:pos[r][(f + 1) * B]] for r in range(W)]) for f in range(F)])
    g_w = (np.concatenate([np.concatenate([per_rank[r][2][pos[r][f * B]:pos[r][(f + 1) * B]] for r in range(W)]) for f in range(F)])
           if weighted else None)
    g_grad = np.concatenate([per_rank[r][3] for r in range(W)], axis=0) / W
    g_offs = np.concatenate([[0], np.cumsum(g_len)]).astype(np.int64)
    oracle_backward_mixed(tabs, g_vals, g_offs, g_grad, oracle.OPT_EXACT_SGD, LR, g_w, feat_mean)
    seen_rows = {t: 0 for t in range(F)}
    for r in range(W):
        for name, (w, row0) in ret[r][1].items():
            t = int(name[1:])
            np.testing.assert_allclose(w, tabs.weights[t][row0:row0 + w.shape[0]], rtol=1e-5, atol=1e-5)
            seen_rows[t] += w.shape[0]
    for t in range(F):
        replicas = W if kinds[f"t{t}"] == "data_parallel" else 1
        assert seen_rows[t] == ROWS[t] * replicas, "sharded rows live on exactly one rank, replicated tables on all"
    return dict(ret)


def test_planner_criteo_plans():
    """26 Criteo tables: the 11 tables with <= 2500 rows are replicated (data-parallel), the other 15 are
    placed table-wise with the smallest possible maximum per rank; nothing is row-wise (all fit)."""
    from torchrec_amd.datasets.random import CRITEO_1TB_ROWS
    from torchrec_amd.distributed.planner import EmbeddingShardingPlanner, Topology, rw_shard_rows
    from torchrec_amd.modules.embedding_configs import EmbeddingBagConfig

    tables = [EmbeddingBagConfig(name=f"t{i}", embedding_dim=128, num_embeddings=r, feature_names=[f"c{i}"])
              for i, r in enumerate(CRITEO_1TB_ROWS)]
    for W in (1, 2, 4, 8):
        plan = EmbeddingShardingPlanner(Topology(W)).plan_tables(tables)
        kinds = [p.sharding_type for p in plan.values()]
        if W == 1:
            assert kinds.count("table_wise") == 26
            continue
        assert kinds.count("data_parallel") == 11 and kinds.count("row_wise") == 0
        per_rank = [sum(1 for p in plan.values() if p.sharding_type == "table_wise" and p.ranks == [r]) for r in range(W)]
        assert sum(per_rank) == 15 and max(per_rank) == -(-15 // W), per_rank
        mem = [0] * W
        for t in tables:
            p = plan[t.name]
            if p.sharding_type == "table_wise":
                mem[p.ranks[0]] += t.num_embeddings * 512
        assert max(mem) < 288e9 * 0.85
    # capacity forces row-wise: one table larger than a GPU
    big = [EmbeddingBagConfig(name="huge", embedding_dim=128, num_embeddings=700_000_000, feature_names=["h"])]
    plan = EmbeddingShardingPlanner(Topology(8)).plan_tables(big)
    assert plan["huge"].sharding_type == "row_wise"
    # rw_shard_rows examples of planner/enumerators.py:277-312
    assert rw_shard_rows(10, 3) == [4, 4, 2] and rw_shard_rows(5, 4) == [2, 2, 1, 0]


def test_planner_host_offload_for_tables_beyond_hbm():
    """BASELINE config 4 shape: > 2 TB of rows on 8 x 288 GB.  Row-wise shards alone exceed HBM, so the
    largest tables move to host memory behind the HBM row cache (batched_fused_uvm_caching,
    torchrec/distributed/embedding_types.py:57-76) until the plan fits; constraints can force a kernel."""
    from torchrec_amd.distributed.planner import EmbeddingShardingPlanner, ParameterConstraints, Topology
    from torchrec_amd.modules.embedding_configs import EmbeddingBagConfig

    rows = [2_000_000_000, 1_500_000_000, 900_000_000, 40_000_000, 1000]  # 2.27 TB at D = 128 fp32
    tables = [EmbeddingBagConfig(name=f"t{i}", embedding_dim=128, num_embeddings=r, feature_names=[f"c{i}"])
              for i, r in enumerate(rows)]
    topo = Topology(8)
    plan = EmbeddingShardingPlanner(topo).plan_tables(tables)
    assert [plan[f"t{i}"].sharding_type for i in range(5)] == ["row_wise"] * 3 + ["table_wise", "data_parallel"]
    kernels = [plan[f"t{i}"].compute_kernel for i in range(3)]
    assert kernels[0] == "batched_fused_uvm_caching" and kernels[2] == "batched_fused"
    cap = topo.hbm_cap * (1 - topo.hbm_reserve_fraction)
    hbm = 0.0
    for i in range(3):
        shard = -(-rows[i] // 8) * 512
        hbm += shard * (topo.caching_ratio if kernels[i].endswith("caching") else 1.0)
    assert hbm <= cap
    # one GPU: a table larger than HBM is offloaded instead of failing
    plan1 = EmbeddingShardingPlanner(Topology(1)).plan_tables(tables[:1])
    assert plan1["t0"].sharding_type == "table_wise" and plan1["t0"].compute_kernel == "batched_fused_uvm_caching"
    # forced kernel
    forced = EmbeddingShardingPlanner(Topology(2), constraints={
        "t3": ParameterConstraints(sharding_types=["row_wise"], compute_kernels=["batched_fused_uvm"])}).plan_tables(tables[3:])
    assert forced["t3"].sharding_type == "row_wise" and forced["t3"].compute_kernel == "batched_fused_uvm"


def _e2e_worker(rank, W, port, ret):
    """Full DLRM train steps through DistributedModelParallel + DDP + TrainPipelineSparseDist on
    gloo/CPU with the oracle TBE (tests/_oracle_tbe.py)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    try:
        import _cpu_ops
        _cpu_ops.register()
        from _oracle_tbe import oracle_dp_tbe_factory, oracle_tbe_factory
        from torchrec_amd.datasets.random import RandomRecDataset
        from torchrec_amd.distributed.embeddingbag import EmbeddingBagCollectionSharder
        from torchrec_amd.distributed.model_parallel import DistributedModelParallel
        from torchrec_amd.distributed.planner import EmbeddingShardingPlanner, Topology
        from torchrec_amd.distributed.train_pipeline import TrainPipelineSparseDist
        from torchrec_amd.distributed.types import ShardingEnv
        from torchrec_amd.models.dlrm import DLRMTrain
        from torchrec_amd.modules.embedding_configs import EmbeddingBagConfig
        from torchrec_amd.modules.embedding_modules import EmbeddingBagCollection
        from torchrec_amd.optim.keyed import CombinedOptimizer, KeyedOptimizerWrapper

        torch.manual_seed(0)  # same dense init on every rank (DDP also broadcasts rank 0's)
        rows = [50, 9, 31, 17, 8]
        D = 8
        keys = [f"c{i}" for i in range(len(rows))]
        tables = [EmbeddingBagConfig(name=f"t{i}", embedding_dim=D, num_embeddings=rows[i], feature_names=[keys[i]])
                  for i in range(len(rows))]
        ebc = EmbeddingBagCollection(tables, device=torch.device("meta"))
        dev = torch.device("cpu")
        train_model = DLRMTrain(ebc, dense_in_features=13, dense_arch_layer_sizes=[16, D], over_arch_layer_sizes=[12, 1],
                                dense_device=dev)
        env = ShardingEnv.from_process_group(dist.group.WORLD)
        model = DistributedModelParallel(train_model, env=env, device=dev,
                                         sharders=[EmbeddingBagCollectionSharder({"learning_rate": 0.05}, oracle_tbe_factory,
                                                                                 oracle_dp_tbe_factory)],
                                         planner=EmbeddingShardingPlanner(Topology(W, "cpu"), dp_max_rows=10))
        opt = CombinedOptimizer([model.fused_optimizer,
                                 KeyedOptimizerWrapper(dict(model.named_parameters()), lambda p: torch.optim.SGD(p, lr=0.05))])
        data = RandomRecDataset(keys, 4, rows, manual_seed=100 + rank, num_generated_batches=3, num_batches=5, device=dev)
        keys_before = (sorted(model.state_dict().keys()), sorted(k for k, _ in model.named_parameters()))
        pipe = TrainPipelineSparseDist(model, opt, dev)
        model.train()
        it = iter(data)
        losses = []
        for _ in range(3):
            loss = pipe.progress(it)[0]
            losses.append(float(loss))
        # the pipeline rewrites the sharded module's forward on the instance (train_pipeline.py:193-243): the module tree,
        # the parameter names and the state_dict keys are what they were, and a checkpoint taken before loads after
        assert (sorted(model.state_dict().keys()), sorted(k for k, _ in model.named_parameters())) == keys_before
        model.load_state_dict({k: (v.clone() if torch.is_tensor(v) else v) for k, v in model.state_dict().items()})
        # the 4th step under a torch profiler: the reference's range labels must show up (and only then: without a
        # profiler `label()` hands out a shared no-op context, torchrec_amd/profiling.py)
        from torchrec_amd import profiling
        assert profiling.label("## forward ##") is profiling.label("## backward ##")  # no profiler: the no-op singleton
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
            loss = pipe.progress(it)[0]
        losses.append(float(loss))
        seen_labels = sorted({e.key for e in prof.key_averages() if e.key.startswith("## ")})
        dense_sd = {k: v.detach().clone() for k, v in model.named_parameters()}
        shards = {n: (w.clone().numpy(), r0) for n, (w, r0) in model.sharded_modules()[0].local_shards().items()}
        assert set(model.sharded_modules()[0].dp_tables()) == {"t1", "t4"}  # 9 and 8 rows: replicated
        ret[rank] = (losses, {k: v.numpy() for k, v in dense_sd.items()}, shards, seen_labels)
    finally:
        dist.destroy_process_group()


def test_dlrm_e2e_dmp_ddp_pipeline_world2():
    W = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_e2e_worker, args=(W, _free_port(), ret), nprocs=W, join=True)
    l0, d0, s0, labels0 = ret[0]
    l1, d1, s1, _ = ret[1]
    assert all(np.isfinite(l0)) and all(np.isfinite(l1))
    # pipeline stages + exchange req / wait pairs, under the reference's label strings (train_pipeline.py:504-550,
    # dist_data.py:190, comm_ops.py:489, 591)
    for want in ("## zero_grad ##", "## forward ##", "## backward ##", "## optimizer ##", "## all2all_data:indices ##",
                 "## alltoall_fwd_single ##", "## alltoall_bwd_single ##", "## tbe_lookup ##"):
        assert want in labels0, (want, labels0)
    # DDP keeps the dense replicas identical
    for k in d0:
        np.testing.assert_allclose(d0[k], d1[k], rtol=0, atol=0)
    # every table row lives on exactly one rank
    assert any("_dp_module.weights" in k for k in d0), "replicated tables must be dense parameters under DDP"
    rows = {"t0": 50, "t2": 31, "t3": 17}  # the sharded ones
    seen = {k: 0 for k in rows}
    for s in (s0, s1):
        for n, (w, _) in s.items():
            seen[n] += w.shape[0]
    assert seen == rows


def _seq_worker(rank, W, port, ret, row_wise=()):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    try:
        import _cpu_ops
        _cpu_ops.register()
        from _oracle_tbe import oracle_seq_tbe_factory
        from torchrec_amd.distributed.embedding import ShardedEmbeddingCollection
        from torchrec_amd.distributed.types import ParameterSharding, ShardingEnv
        from torchrec_amd.modules.embedding_configs import EmbeddingConfig
        from torchrec_amd.sparse.jagged_tensor import KeyedJaggedTensor

        rows, D, B = [30, 11, 19], 4, 5
        keys = ["a", "b", "c"]
        cfgs = [EmbeddingConfig(name=f"t{i}", embedding_dim=D, num_embeddings=rows[i], feature_names=[keys[i]]) for i in range(3)]
        plan = {"t0": ParameterSharding("table_wise", "batched_fused", [1]),
                "t1": ParameterSharding("table_wise", "batched_fused", [0]),
                "t2": ParameterSharding("table_wise", "batched_fused", [1])}
        for n in row_wise:
            plan[n] = ParameterSharding("row_wise", "batched_fused", list(range(W)))
        sec = ShardedEmbeddingCollection(cfgs, plan, ShardingEnv.from_process_group(dist.group.WORLD),
                                         {"learning_rate": 0.5}, torch.device("cpu"), oracle_seq_tbe_factory)
        init = [np.random.default_rng(100 + t).standard_normal((rows[t], D)).astype(np.float32) for t in range(3)]
        r0 = sec.local_shard_row_offsets()
        for name, w in sec.local_shards().items():
            w.copy_(torch.from_numpy(init[int(name[1:])][r0[name]:r0[name] + w.shape[0]]))
        rng = np.random.default_rng(7 + rank)
        lengths = rng.integers(0, 4, size=3 * B).astype(np.int32)
        vals = np.concatenate([rng.integers(0, rows[f], size=int(lengths[f * B:(f + 1) * B].sum())) for f in range(3)]).astype(np.int64)
        kjt = KeyedJaggedTensor.from_lengths_sync(keys, torch.from_numpy(vals), torch.from_numpy(lengths))
        out = sec(kjt).wait()
        embs = {k: out[k].values() for k in keys}
        cat = torch.cat([embs[k] for k in keys])
        g = np.random.default_rng(70 + rank).standard_normal(tuple(cat.shape)).astype(np.float32)
        cat.backward(torch.from_numpy(g))
        row0 = sec.local_shard_row_offsets()
        ret[rank] = ({k: embs[k].detach().numpy().copy() for k in keys}, lengths, vals, g,
                     {n: (w.clone().numpy(), row0[n]) for n, w in sec.local_shards().items()})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("row_wise", [(), ("t0",), ("t0", "t2"), ("t0", "t1", "t2")])
def test_sharded_sequence_embedding_world2(row_wise):
    """Table-wise, mixed and all row-wise (bucketize + unbucketize_permute) sequence sharding."""
    from oracle import oracle

    W = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_seq_worker, args=(W, _free_port(), ret, row_wise), nprocs=W, join=True)
    _check_seq(ret, W)


def _check_seq(ret, W, D=4):
    from oracle import oracle

    rows, B = [30, 11, 19], 5
    init = [np.random.default_rng(100 + t).standard_normal((rows[t], D)).astype(np.float32) for t in range(3)]
    keys = ["a", "b", "c"]
    tabs = oracle.Tables(rows, [D] * 3)
    for t in range(3):
        tabs.weights[t][...] = init[t]
    # forward: each rank's per-feature rows are plain gathers
    for r in range(W):
        embs, lengths, vals, g, _ = ret[r]
        pos = np.concatenate([[0], np.cumsum([lengths[f * B:(f + 1) * B].sum() for f in range(3)])])
        for f, k in enumerate(keys):
            np.testing.assert_array_equal(embs[k], init[f][vals[pos[f]:pos[f + 1]]])
    # backward: global batch = concat over ranks per feature, exact SGD (the reference does not divide the
    # sequence gradient by the world size, unlike the pooled path)
    g_vals, g_grad, g_len = [], [], []
    for f in range(3):
        for r in range(W):
            embs, lengths, vals, g, _ = ret[r]
            pos = np.concatenate([[0], np.cumsum([lengths[ff * B:(ff + 1) * B].sum() for ff in range(3)])])
            g_vals.append(vals[pos[f]:pos[f + 1]])
            g_grad.append(g[pos[f]:pos[f + 1]])  # no 1/W on the sequence path (comm_ops.py:718-749)
            g_len.append(lengths[f * B:(f + 1) * B])
    g_vals, g_grad, g_len = np.concatenate(g_vals), np.concatenate(g_grad), np.concatenate(g_len)
    offs = np.concatenate([[0], np.cumsum(g_len)]).astype(np.int64)
    oracle.tbe_backward(tabs, g_vals, offs, g_grad, oracle.OPT_EXACT_SGD, 0.5, None, oracle.POOL_NONE)
    seen = {"t0": 0, "t1": 0, "t2": 0}
    for r in range(W):
        for name, (w, row0) in ret[r][4].items():
            np.testing.assert_allclose(w, tabs.weights[int(name[1:])][row0:row0 + w.shape[0]], rtol=1e-5, atol=1e-5)
            seen[name] += w.shape[0]
    assert seen == {"t0": 30, "t1": 11, "t2": 19}


def _half_exchange_worker(rank, W, port, n_rw, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    try:
        import _cpu_ops
        _cpu_ops.register()
        from _oracle_tbe import oracle_dp_tbe_factory, oracle_tbe_factory
        from torchrec_amd.distributed._rehearsal import stage_all_to_all_through_host
        from torchrec_amd.distributed.embeddingbag import ShardedEmbeddingBagCollection, _ExchangeState
        from torchrec_amd.distributed.planner import EmbeddingShardingPlanner, Topology
        from torchrec_amd.distributed.types import ShardingEnv
        from torchrec_amd.modules.embedding_configs import EmbeddingBagConfig
        from torchrec_amd.modules.embedding_modules import EmbeddingBagCollection

        stage_all_to_all_through_host()  # gloo has no list-form all-to-all: one all_to_all_single underneath
        keys = [f"f{i}" for i in range(len(ROWS))]
        tables = [EmbeddingBagConfig(name=f"t{i}", embedding_dim=DIMS[i], num_embeddings=ROWS[i], feature_names=[keys[i]])
                  for i in range(len(ROWS))]
        ebc = EmbeddingBagCollection(tables, device=torch.device("meta"))
        plan = EmbeddingShardingPlanner(Topology(W, "cpu"), num_row_wise=n_rw, dp_max_rows=0).plan_tables(tables)
        sebc = ShardedEmbeddingBagCollection(ebc, plan, ShardingEnv.from_process_group(dist.group.WORLD), {"learning_rate": LR},
                                             torch.device("cpu"), tbe_factory=oracle_tbe_factory,
                                             dp_tbe_factory=oracle_dp_tbe_factory)
        B = 6
        g = torch.Generator().manual_seed(100 + rank)
        emb = torch.randn(W * B, sebc._D_local, generator=g)  # what this rank's lookup would hand to the exchange
        grad = torch.randn(B, sebc._D_total, generator=g)     # gradient of this rank's pooled output
        whole = _ExchangeState(sebc, B)
        whole.start_forward(emb)
        out_whole = whole.finish_forward().clone()
        whole.start_backward(grad)
        back_whole = whole.finish_backward().clone()
        halves = _ExchangeState(sebc, B)
        halves.start_forward_halves(emb)
        rows = [halves.finish_forward_half(h) for h in range(2)]
        out_halves = halves.output_destination().clone()
        assert rows[0].shape == (B // 2, sebc._D_total) and rows[1].data_ptr() == halves.output_destination()[B // 2:].data_ptr()
        for h in range(2):
            halves.start_backward_half(h, grad[h * (B // 2):(h + 1) * (B // 2)].contiguous())
        back_halves = halves.finish_backward().clone()
        ret[rank] = (out_whole.numpy(), out_halves.numpy(), back_whole.numpy(), back_halves.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_rw", [0, 2, 5])
def test_half_batch_exchange_equals_whole_batch_world2(n_rw):
    """The pooled exchange in two half-batches (_ExchangeState.start_forward_halves / finish_forward_half /
    start_backward_half: list-form all-to-all over row-range views, unpack / pack with the B/2 layout on the half's rows)
    moves exactly what the whole-batch exchange moves, forward and backward, table-wise and row-wise shards."""
    from _results import ResultStore

    W = 2
    ret = ResultStore()
    mp.spawn(_half_exchange_worker, args=(W, _free_port(), n_rw, ret), nprocs=W, join=True)
    for r in range(W):
        out_whole, out_halves, back_whole, back_halves = ret[r]
        np.testing.assert_array_equal(out_halves, out_whole)
        np.testing.assert_array_equal(back_halves, back_whole)
        assert np.abs(out_whole).sum() > 0 and np.abs(back_whole).sum() > 0
