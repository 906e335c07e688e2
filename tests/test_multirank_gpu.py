"""GPU rehearsal of the N > 1 path on a ONE-GPU box (the pool has no multi-GPU box for this author):

* world_size 2, both ranks on cuda:0, `gloo` process group: real HIP TBE (a2a-ready output layout,
  row-wise masking through the bounds check), real exchange unpack / pack kernels, real dense-gradient
  TBE for the replicated tables, DDP on device tensors.  gloo has no device all-to-all, so the test
  stages `all_to_all_single` through host memory — that is test plumbing around the product path.
* world_size 1 on RCCL (`nccl` backend) with the exchange forced on: the asynchronous id / pooled
  all-to-all, their stream hand-over and the exchange kernels run through the real RCCL
  ProcessGroup.

Checks follow the reference's sharded-vs-unsharded pattern
(torchrec/distributed/test_utils/test_model_parallel_base.py:148-294).
"""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _paths  # noqa: F401
from _results import ResultStore
from test_sharded_gloo import _free_port

pytestmark = pytest.mark.gpu

ROWS = [5000, 7, 230, 90000, 5, 1201]
D = 128
LR = 0.25
B_LOCAL = 48


class _Done:
    def wait(self):
        return True


def _stage_a2a_through_host():
    """gloo has no device all-to-all: the package's rehearsal plumbing stages both forms through the host."""
    from torchrec_amd.distributed._rehearsal import stage_all_to_all_through_host

    stage_all_to_all_through_host()


def _data(W, fixed_len, weighted, seed=11, max_len=3):
    rng = np.random.default_rng(seed)
    F = len(ROWS)
    per_rank = []
    for _ in range(W):
        lengths = (np.full(F * B_LOCAL, fixed_len) if fixed_len else rng.integers(0, max_len + 1, size=F * B_LOCAL)).astype(np.int32)
        vals = np.concatenate([rng.integers(0, ROWS[f], size=int(lengths[f * B_LOCAL:(f + 1) * B_LOCAL].sum()))
                               for f in range(F)]).astype(np.int64)
        wts = (rng.random(vals.size).astype(np.float32) + 0.5) if weighted else None
        grad = rng.standard_normal((B_LOCAL, F * D)).astype(np.float32)
        per_rank.append((lengths, vals, wts, grad))
    init = [rng.standard_normal((r, D)).astype(np.float32) for r in ROWS]
    return per_rank, init


def _build_sharded(W, backend_env, weighted, n_rw, dp_max_rows, offload=False, adagrad=False, mean=False, rw_mode=None):
    from torchrec_amd.distributed.embeddingbag import ShardedEmbeddingBagCollection
    from torchrec_amd.distributed.planner import EmbeddingShardingPlanner, ParameterConstraints, Topology
    from torchrec_amd.modules.embedding_configs import EmbeddingBagConfig
    from torchrec_amd.modules.embedding_modules import EmbeddingBagCollection

    keys = [f"f{i}" for i in range(len(ROWS))]
    from torchrec_amd.modules.embedding_configs import PoolingType
    # mean: False = SUM, True = MEAN, "mixed" = odd tables MEAN (both pooling types in ONE collection / ONE lookup)
    tables = [EmbeddingBagConfig(name=f"t{i}", embedding_dim=D, num_embeddings=ROWS[i], feature_names=[keys[i]],
                                 pooling=PoolingType.MEAN if ((i % 2 == 1) if mean == "mixed" else mean) else PoolingType.SUM)
              for i in range(len(ROWS))]
    ebc = EmbeddingBagCollection(tables, is_weighted=weighted, device=torch.device("meta"))
    # offload: the largest table row-wise in host memory behind the HBM row cache (tiny cache: evictions),
    # another one table-wise in plain host-mapped memory
    cons = ({"t3": ParameterConstraints(["row_wise"], ["batched_fused_uvm_caching"]),
             "t0": ParameterConstraints(["table_wise"], ["batched_fused_uvm"])} if offload else None)
    plan = EmbeddingShardingPlanner(Topology(W), constraints=cons, num_row_wise=n_rw,
                                    dp_max_rows=dp_max_rows).plan_tables(tables)
    fused = {"learning_rate": LR, "cache_sets": 2} if offload else {"learning_rate": LR}
    if adagrad:
        from fbgemm_gpu.split_embedding_configs import EmbOptimType
        fused.update({"optimizer": EmbOptimType.EXACT_ROWWISE_ADAGRAD, "eps": 1e-3})
    sebc = ShardedEmbeddingBagCollection(ebc, plan, backend_env, fused, torch.device("cuda", 0), rw_input_dist=rw_mode)
    return keys, plan, sebc


def _run_rank(sebc, keys, per_rank, init, rank, W, fixed_len, weighted):
    from torchrec_amd.sparse.jagged_tensor import KeyedJaggedTensor

    dev = torch.device("cuda", 0)
    for name, (w, row0) in sebc.local_shards().items():
        w.copy_(torch.from_numpy(init[int(name[1:])][row0:row0 + w.shape[0]]))
    with torch.no_grad():
        for name, w in sebc.dp_tables().items():
            w.copy_(torch.from_numpy(init[int(name[1:])]))
    lengths, vals, wts, grad = per_rank[rank]
    wt = torch.from_numpy(wts).to(dev) if weighted else None
    if fixed_len:
        kjt = KeyedJaggedTensor.from_fixed_lengths(keys, torch.from_numpy(vals).to(dev), [fixed_len] * len(keys), weights=wt)
    else:
        kjt = KeyedJaggedTensor.from_lengths_sync(keys, torch.from_numpy(vals).to(dev), torch.from_numpy(lengths).to(dev),
                                                  weights=wt)
    out = sebc(kjt).wait()
    vals_out = out.values()
    vals_out.backward(torch.from_numpy(grad).to(dev))
    torch.cuda.synchronize()
    shards = {n: (w.detach().cpu().numpy().copy(), r0) for n, (w, r0) in sebc.local_shards().items()}
    if sebc._dp_module is not None:
        g = sebc._dp_module.weights.grad.detach().clone()
        if W > 1:
            gc = g.cpu()
            dist.all_reduce(gc)
            g = gc.to(dev)
        g /= W
        with torch.no_grad():
            sebc._dp_module.weights -= LR * g
        for n, w in sebc.dp_tables().items():
            shards[n] = (w.detach().cpu().numpy().copy(), 0)
    return vals_out.detach().cpu().numpy().copy(), shards


def _worker(rank, W, port, fixed_len, weighted, n_rw, dp_max_rows, ret, offload=False, adagrad=False, mean=False,
            rw_mode=None, max_len=3):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    try:
        _stage_a2a_through_host()
        from torchrec_amd.distributed.types import ShardingEnv

        per_rank, init = _data(W, fixed_len, weighted, max_len=max_len)
        keys, plan, sebc = _build_sharded(W, ShardingEnv.from_process_group(dist.group.WORLD), weighted, n_rw, dp_max_rows,
                                          offload, adagrad, mean, rw_mode)
        out, shards = _run_rank(sebc, keys, per_rank, init, rank, W, fixed_len, weighted)
        if adagrad:  # per-table row-wise state of the local shards (batched_embedding_kernel.py:133-148)
            states = sebc._emb_module.split_optimizer_states()
            for lt, st in zip(sebc._local_tables, states):
                shards[lt.cfg.name] = shards[lt.cfg.name] + (st[0].detach().cpu().numpy().copy(),)
        if offload:
            assert plan["t3"].compute_kernel == "batched_fused_uvm_caching" and sebc._emb_module._cache is not None
        ret[rank] = (out, shards, {n: p.sharding_type for n, p in plan.items()})
        ret[f"errors{rank}"] = sebc._emb_module.bounds_check_errors() if sebc._emb_module is not None else 0
        ret[f"rw_mode{rank}"] = sebc._rw_mode_active
    finally:
        dist.destroy_process_group()


def _check_against_oracle(ret, W, fixed_len, weighted, n_rw, dp_max_rows, adagrad=False, mean=False, max_len=3):
    from _util import oracle_backward_mixed, oracle_forward_mixed
    from oracle import oracle

    feat_mean = [(i % 2 == 1) if mean == "mixed" else bool(mean) for i in range(len(ROWS))]

    per_rank, init = _data(W, fixed_len, weighted, max_len=max_len)
    F, B = len(ROWS), B_LOCAL
    tabs = oracle.Tables(ROWS, [D] * F)
    for t in range(F):
        tabs.weights[t][...] = init[t]
    for r in range(W):
        lengths, vals, wts, _ = per_rank[r]
        offs = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
        ref = oracle_forward_mixed(tabs, vals, offs, wts, feat_mean)
        if fixed_len == 1 and not weighted:
            np.testing.assert_array_equal(ret[r][0], ref)  # pure gather: bit-exact through the whole exchange
        else:
            np.testing.assert_allclose(ret[r][0], ref, rtol=1e-5, atol=1e-5)
    kinds = ret[0][2]
    for r in range(W):  # every id is valid: rows held by another rank's shard must NOT count as bounds errors
        if f"errors{r}" in ret:
            assert ret[f"errors{r}"] == 0
    assert sum(1 for k in kinds.values() if k == "row_wise") == n_rw
    assert sum(1 for k in kinds.values() if k == "data_parallel") == (sum(1 for r in ROWS if r <= dp_max_rows) if W > 1 else 0)
    g_len = np.concatenate([np.concatenate([per_rank[r][0][f * B:(f + 1) * B] for r in range(W)]) for f in range(F)])
    pos = [np.concatenate([[0], np.cumsum(per_rank[r][0])]) for r in range(W)]
    cat = lambda i: np.concatenate([np.concatenate([per_rank[r][i][pos[r][f * B]:pos[r][(f + 1) * B]]  # noqa: E731
                                                    for r in range(W)]) for f in range(F)])
    g_vals, g_w = cat(1), (cat(2) if weighted else None)
    g_grad = np.concatenate([per_rank[r][3] for r in range(W)], axis=0) / W
    g_offs = np.concatenate([[0], np.cumsum(g_len)]).astype(np.int64)
    s0 = [np.zeros(r, dtype=np.float32) for r in ROWS]
    if adagrad:
        # replicated tables are dense parameters stepped by plain SGD in this test; the fused optimizer owns the rest
        sgd_tabs = oracle.Tables(ROWS, [D] * F)
        for t in range(F):
            sgd_tabs.weights[t][...] = init[t]
        oracle.tbe_backward(sgd_tabs, g_vals, g_offs, g_grad, oracle.OPT_EXACT_SGD, LR, g_w)
        oracle.tbe_backward(tabs, g_vals, g_offs, g_grad, oracle.OPT_EXACT_ROWWISE_ADAGRAD, LR, g_w, eps=1e-3, state0=s0)
        for t in range(F):
            if kinds[f"t{t}"] == "data_parallel":
                tabs.weights[t][...] = sgd_tabs.weights[t]
    else:
        oracle_backward_mixed(tabs, g_vals, g_offs, g_grad, oracle.OPT_EXACT_SGD, LR, g_w, feat_mean)
    seen = {t: 0 for t in range(F)}
    for r in range(W):
        for name, shard in ret[r][1].items():
            w, row0 = shard[0], shard[1]
            t = int(name[1:])
            np.testing.assert_allclose(w, tabs.weights[t][row0:row0 + w.shape[0]], rtol=3e-5, atol=3e-5)
            if adagrad and len(shard) > 2:  # sharding invariance of the fused optimizer state (test_fused_optim.py:201-306)
                np.testing.assert_allclose(shard[2], s0[t][row0:row0 + w.shape[0]], rtol=3e-5, atol=3e-5)
            seen[t] += w.shape[0]
    for t in range(F):
        assert seen[t] == ROWS[t] * (W if kinds[f"t{t}"] == "data_parallel" else 1)


@pytest.mark.parametrize("fixed_len,weighted,n_rw,dp_max_rows", [
    (1, False, 0, 0), (1, False, 2, 10), (2, True, 1, 0), (0, False, 1, 10), (0, True, 6, 0)])
def test_sharded_world2_on_one_gpu(fixed_len, weighted, n_rw, dp_max_rows):
    W = 2
    ret = ResultStore()
    mp.spawn(_worker, args=(W, _free_port(), fixed_len, weighted, n_rw, dp_max_rows, ret), nprocs=W, join=True)
    _check_against_oracle(ret, W, fixed_len, weighted, n_rw, dp_max_rows)


@pytest.mark.parametrize("fixed_len,weighted,n_rw,dp_max_rows,offload,adagrad", [
    (1, False, 2, 10, False, False), (0, False, 3, 0, False, False), (0, True, 2, 10, False, False), (1, True, 6, 0, False, False),
    (5, False, 2, 0, False, False), (0, False, 1, 10, True, True)])
@pytest.mark.parametrize("rw_mode", ["bucketize", "windows"])
def test_row_wise_input_dist_modes_world2_on_one_gpu(fixed_len, weighted, n_rw, dp_max_rows, offload, adagrad, rw_mode):
    """the bucketized row-wise input dist of the POOLED path with the real kernels
    (block_bucketize_sparse_features -> lengths / ids exchange -> lookup on LOCAL rows without a row window) next to the
    row-window one: pooling factor 1, ragged bags of up to 6 ids, per-sample weights, all-row-wise plans, row-wise shards
    behind the HBM row cache with fused row-wise Adagrad — sharded == unsharded (oracle), zero bounds errors.
    Reference: embedding_sharding.py:121-184, sharding/rw_sharding.py:229-236, test_model_parallel_base.py:148-294."""
    W = 2
    ret = ResultStore()
    mp.spawn(_worker, args=(W, _free_port(), fixed_len, weighted, n_rw, dp_max_rows, ret, offload, adagrad, False, rw_mode, 6),
             nprocs=W, join=True)
    assert ret["rw_mode0"] == rw_mode and ret["rw_mode1"] == rw_mode
    _check_against_oracle(ret, W, fixed_len, weighted, n_rw, dp_max_rows, adagrad=adagrad, max_len=6)


def _rccl_worker(rank, port, fixed_len, weighted, dp_max_rows, ret, plain_group=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    from torchrec_amd.distributed.comm import init_rccl_process_group

    if plain_group:  # what a launcher written for CUDA does (examples/dlrm/dlrm_main.py:469-478)
        dist.init_process_group("nccl", rank=0, world_size=1)
    else:
        init_rccl_process_group(torch.device("cuda", 0), rank=0, world_size=1)
    try:
        import torchrec_amd.distributed.embeddingbag as eb
        from torchrec_amd.distributed.types import ShardingEnv

        eb.FORCE_EXCHANGE = True
        per_rank, init = _data(1, fixed_len, weighted)
        keys, plan, sebc = _build_sharded(1, ShardingEnv.from_process_group(dist.group.WORLD), weighted, 0, dp_max_rows)
        assert sebc._exchange
        # the exchanges run on a high-priority collective stream either way: the environment's group if it has one, else a
        # second communicator the sharded module made for itself (distributed/comm.py exchange_group)
        assert (sebc._pg is dist.group.WORLD) == (not plain_group)
        assert sebc._pg._get_backend(torch.device("cuda", 0)).options.is_high_priority_stream
        out, shards = _run_rank(sebc, keys, per_rank, init, 0, 1, fixed_len, weighted)
        ret[0] = (out, shards, {n: p.sharding_type for n, p in plan.items()})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_rw,offload", [(0, False), (2, False), (1, True)])
def test_sharded_world2_fused_rowwise_adagrad(n_rw, offload):
    """The reference's fused-optimizer test (torchrec/distributed/tests/test_fused_optim.py:60-140, 201-306:
    RW / TW x EXACT_ROWWISE_ADAGRAD): weights AND optimizer state of the shards equal the unsharded run —
    here with the real kernels, incl. row-wise shards behind the HBM row cache."""
    W = 2
    ret = ResultStore()
    mp.spawn(_worker, args=(W, _free_port(), 2, False, n_rw, 10, ret, offload, True), nprocs=W, join=True)
    _check_against_oracle(ret, W, 2, False, n_rw, 10, adagrad=True)


def test_sharded_world2_mean_pooling_over_row_wise_shards():
    """MEAN pooling with row-wise shards + replicated tables: every rank divides its partial sum by the FULL bag
    length (all ids reach every rank; rows outside its block are masked by the bounds check)."""
    W = 2
    ret = ResultStore()
    mp.spawn(_worker, args=(W, _free_port(), 0, False, 2, 10, ret, False, False, True), nprocs=W, join=True)
    _check_against_oracle(ret, W, 0, False, 2, 10, mean=True)


def test_sharded_world2_sum_and_mean_tables_in_one_collection():
    """Lookup groups without groups: SUM and MEAN tables, table-wise + row-wise + replicated, in ONE
    sharded collection with ONE fused lookup per rank (per-feature pooling in the kernels) against the oracle."""
    W = 2
    ret = ResultStore()
    mp.spawn(_worker, args=(W, _free_port(), 0, False, 2, 10, ret, False, False, "mixed"), nprocs=W, join=True)
    _check_against_oracle(ret, W, 0, False, 2, 10, mean="mixed")


def test_sharded_world2_with_host_offloaded_tables():
    """Row-wise shards of a table in host memory behind the HBM row cache + a table-wise table in plain
    host-mapped memory (BASELINE config 4's placement), two ranks."""
    W = 2
    ret = ResultStore()
    mp.spawn(_worker, args=(W, _free_port(), 2, False, 1, 10, ret, True), nprocs=W, join=True)
    _check_against_oracle(ret, W, 2, False, 1, 10)


def _bad_id_worker(rank, W, port, offload, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    try:
        _stage_a2a_through_host()
        from torchrec_amd.distributed.types import ShardingEnv
        from torchrec_amd.sparse.jagged_tensor import KeyedJaggedTensor

        per_rank, init = _data(W, 1, False)
        keys, plan, sebc = _build_sharded(W, ShardingEnv.from_process_group(dist.group.WORLD), False, 2, 0, offload)
        dev = torch.device("cuda", 0)
        for name, (w, row0) in sebc.local_shards().items():
            w.copy_(torch.from_numpy(init[int(name[1:])][row0:row0 + w.shape[0]]))
        vals = _inject_bad_ids(per_rank[rank][1].copy(), plan, rank)
        kjt = KeyedJaggedTensor.from_fixed_lengths(keys, torch.from_numpy(vals).to(dev), [1] * len(keys))
        with torch.no_grad():  # forward only: the backward's linearize would count the same ids a second time
            out = sebc(kjt).wait().values()
        torch.cuda.synchronize()
        ret[rank] = (out.detach().cpu().numpy().copy(), {n: (p.sharding_type, p.ranks) for n, p in plan.items()})
        ret[f"errors{rank}"] = sebc._emb_module.bounds_check_errors()
    finally:
        dist.destroy_process_group()


def _inject_bad_ids(vals, plan, rank):
    """Rank r's batch gets, in EVERY feature, one id just past the table (sample 3 + r) and one negative id
    (sample 7 + r)."""
    for f in range(len(ROWS)):
        vals[f * B_LOCAL + 3 + rank] = ROWS[f] + rank
        vals[f * B_LOCAL + 7 + rank] = -1 - rank
    return vals


@pytest.mark.parametrize("offload", [False, True])
def test_row_wise_shards_report_exactly_the_truly_out_of_range_ids(offload):
    """Two ranks, two row-wise tables (one behind the HBM row cache when offload) + table-wise ones.  Every rank
    sees every id of a row-wise feature: rows of the other rank's shard are skipped SILENTLY, ids outside
    [0, global rows) are counted by every rank that sees them; a table-wise feature's bad ids are counted by its
    owner only.  Outputs equal the oracle's (a bad id contributes a zero row).  Reference contract:
    embedding_sharding.py:121-184 + rw_sharding.py:229-236 (bucketize, so foreign rows never arrive)."""
    from oracle import oracle

    W = 2
    ret = ResultStore()
    mp.spawn(_bad_id_worker, args=(W, _free_port(), offload, ret), nprocs=W, join=True)
    per_rank, init = _data(W, 1, False)
    F = len(ROWS)
    tabs = oracle.Tables(ROWS, [D] * F)
    for t in range(F):
        tabs.weights[t][...] = init[t]
    plan = ret[0][1]
    for r in range(W):
        vals = _inject_bad_ids(per_rank[r][1].copy(), plan, r)
        ref, nbad = oracle.tbe_forward(tabs, vals, np.arange(F * B_LOCAL + 1, dtype=np.int64), None, oracle.POOL_SUM)
        assert nbad == 2 * F
        np.testing.assert_array_equal(ret[r][0], ref)
    for r in range(W):
        expect = 0
        for f in range(F):
            kind, ranks = plan[f"t{f}"]
            if kind == "row_wise":
                expect += 2 * W  # both ranks' two bad ids reach this rank
            elif kind == "table_wise" and ranks[0] == r:
                expect += 2 * W  # the owner sees both ranks' batches
        assert ret[f"errors{r}"] == expect, (r, ret[f"errors{r}"], expect)


@pytest.mark.parametrize("fixed_len,weighted,dp_max_rows,plain_group", [(1, False, 10, False), (0, True, 0, False),
                                                                        (1, False, 10, True)])
def test_exchange_through_rccl_world1(fixed_len, weighted, dp_max_rows, plain_group):
    """The asynchronous id + pooled all-to-all and the exchange kernels over a real RCCL group; plain_group: the process
    group as a CUDA launcher creates it — the sharded module then makes itself a communicator with a high-priority stream."""
    ret = ResultStore()
    mp.spawn(_rccl_worker, args=(_free_port(), fixed_len, weighted, dp_max_rows, ret, plain_group), nprocs=1, join=True)
    _check_against_oracle(ret, 1, fixed_len, weighted, 0, dp_max_rows)


# ---- full DLRM train loop: 2 ranks on one GPU == 1 rank on the global batch -------------------------

E_ROWS = [3000, 9, 31000, 170, 8, 5, 999]
E_B = 64  # per rank
E_STEPS = 4
E_LR = 0.05


def _e2e_batches(W):
    rng = np.random.default_rng(5)
    out = []
    for _ in range(E_STEPS + 2):
        dense = rng.standard_normal((W * E_B, 13)).astype(np.float32)
        ids = np.stack([rng.integers(0, r, size=W * E_B) for r in E_ROWS]).astype(np.int64)  # [F, W*B]
        labels = rng.integers(0, 2, size=W * E_B).astype(np.int64)
        out.append((dense, ids, labels))
    return out


def _e2e_model(env, dev, dp_max_rows, graph_batch=0, flat=False, seed=0, halves=False):
    from torchrec_amd.distributed.embeddingbag import EmbeddingBagCollectionSharder
    from torchrec_amd.distributed.model_parallel import DistributedModelParallel
    from torchrec_amd.distributed.planner import EmbeddingShardingPlanner, Topology
    from torchrec_amd.models.dlrm import DLRMTrain
    from torchrec_amd.modules.embedding_configs import EmbeddingBagConfig
    from torchrec_amd.modules.embedding_modules import EmbeddingBagCollection
    from torchrec_amd.optim.keyed import CombinedOptimizer, KeyedOptimizerWrapper

    torch.manual_seed(seed)
    keys = [f"c{i}" for i in range(len(E_ROWS))]
    tables = [EmbeddingBagConfig(name=f"t{i}", embedding_dim=D, num_embeddings=E_ROWS[i], feature_names=[keys[i]])
              for i in range(len(E_ROWS))]
    ebc = EmbeddingBagCollection(tables, device=torch.device("meta"))
    tm = DLRMTrain(ebc, 13, [64, D], [96, 32, 1], dense_device=dev)
    model = DistributedModelParallel(tm, env=env, device=dev, sharders=[EmbeddingBagCollectionSharder({"learning_rate": E_LR})],
                                     planner=EmbeddingShardingPlanner(Topology(env.world_size), num_row_wise=1,
                                                                      dp_max_rows=dp_max_rows),
                                     init_data_parallel=not graph_batch)
    if graph_batch:  # HIP-graph segments must be captured BEFORE DistributedDataParallel wraps the dense modules
        tm.capture_hip_graphs(graph_batch, flat_grads=flat, process_group=env.process_group, half_batches=halves)
        model.init_data_parallel()
    opt = CombinedOptimizer([model.fused_optimizer,
                             # flat mode: the one-kernel SGD over the flat buffers, as bench.py builds it (optim/flat.py)
                             KeyedOptimizerWrapper(dict(model.named_parameters()),
                                                   (lambda p: tm.dense_optimizer(p, lr=E_LR)) if (graph_batch and flat)
                                                   else (lambda p: torch.optim.SGD(p, lr=E_LR)))])
    return keys, model, opt


def _e2e_init_tables(model):
    s = model.sharded_modules()[0]
    for name, (w, row0) in s.local_shards().items():
        t = int(name[1:])
        full = np.random.default_rng(900 + t).standard_normal((E_ROWS[t], D)).astype(np.float32) * 0.1
        w.copy_(torch.from_numpy(full[row0:row0 + w.shape[0]]))
    with torch.no_grad():
        for name, w in s.dp_tables().items():
            t = int(name[1:])
            w.copy_(torch.from_numpy(np.random.default_rng(900 + t).standard_normal((E_ROWS[t], D)).astype(np.float32) * 0.1))


def _e2e_run(model, opt, keys, batches, rank, W, dev, hip_graphs=False):
    from torchrec_amd.datasets.random import Batch
    from torchrec_amd.distributed.train_pipeline import TrainPipelineSparseDist
    from torchrec_amd.sparse.jagged_tensor import KeyedJaggedTensor

    sl = slice(rank * E_B, (rank + 1) * E_B) if W > 1 else slice(None)
    bl = [Batch(torch.from_numpy(d[sl]).to(dev),
                KeyedJaggedTensor.from_fixed_lengths(keys, torch.from_numpy(np.ascontiguousarray(i[:, sl]).reshape(-1)).to(dev),
                                                     [1] * len(keys)),
                torch.from_numpy(lab[sl]).to(dev)) for d, i, lab in batches]
    # the flat-gradient graph mode also runs with the next step's lookup prefetched behind the embedding backward
    pipe = TrainPipelineSparseDist(model, opt, dev, hip_graphs=hip_graphs, prefetch_lookup=bool(hip_graphs))
    model.train()
    it = iter(bl)
    losses = []
    for _ in range(E_STEPS):
        loss = pipe.progress(it)[0]
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    s = model.sharded_modules()[0]
    tabs = {n: (w.detach().cpu().numpy().copy(), r0) for n, (w, r0) in s.local_shards().items()}
    for n, w in s.dp_tables().items():
        tabs[n] = (w.detach().cpu().numpy().copy(), 0)
    dense = {k: v.detach().cpu().numpy().copy() for k, v in model.named_parameters() if "_dp_module" not in k}
    return losses, tabs, dense


def _e2e_worker(rank, W, port, ret, hip_graphs=False):
    os.environ["TORCHREC_AMD_WGRAD_LATE_LAYERS"] = "2"  # flat modes: with the late weight-gradient graph and its all-reduce
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    try:
        _stage_a2a_through_host()
        from torchrec_amd.distributed.types import ShardingEnv

        halves = hip_graphs == "flat-halves"  # the exchange in two half-batches (capture_hip_graphs(half_batches=True))
        if hip_graphs == "flat-early":  # every piece of the flat gradient all-reduced as soon as it exists (split mode "early")
            os.environ["TORCHREC_AMD_WGRAD_SPLIT_MODE"] = "early"
            hip_graphs = "flat"
        if halves:
            hip_graphs = "flat"
        keys, model, opt = _e2e_model(ShardingEnv.from_process_group(dist.group.WORLD), dev, dp_max_rows=10,
                                      graph_batch=E_B if hip_graphs else 0, flat=(hip_graphs == "flat"),
                                      seed=0 if rank == 0 else 77,  # ranks > 0 must receive rank 0's dense weights
                                      halves=halves)
        if hip_graphs == "flat":
            assert len(model.module.flat_grad_parameters()) > 0
        _e2e_init_tables(model)
        # "flat": the pipeline is ALSO asked for HIP graphs: its lazy capture must recognise the owner's
        # capture and leave it alone — re-capturing would swap in a world-1 flat-gradient state (no dense all-reduce,
        # replicas diverge: the d0 == d1 check of the test would fail) and strand the optimizer's flat buffers
        flat_state = getattr(model.module, "_flat_dense", None)
        ret[rank] = _e2e_run(model, opt, keys, _e2e_batches(W), rank, W, dev, hip_graphs == "flat")
        assert getattr(model.module, "_flat_dense", None) is flat_state
        assert (model.module._graphs is not None) == bool(hip_graphs)
        # flat-gradient graph mode runs forward AND backward by hand (DLRMTrain._explicit_step), no autograd engine
        assert (getattr(model.module, "explicit_steps", 0) == E_STEPS) == (hip_graphs == "flat")
        assert getattr(model.module, "half_batch_steps", 0) == (E_STEPS if halves else 0)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("hip_graphs", [False, True, "flat", "flat-halves", "flat-early"])
def test_dlrm_train_world2_on_one_gpu_matches_world1(hip_graphs):
    """hip_graphs=True: the dense segments replay from HIP graphs under DistributedDataParallel — captured
    BEFORE the DDP wrap (capturing a backward graph over DDP-managed parameters crashes in
    hipStreamEndCapture on this stack; the pipeline's lazy capture is therefore declined under DDP).
    "flat": the graphed segments' gradients additionally travel through one flat all-reduced buffer instead
    of DDP (models/dlrm.py capture_hip_graphs(flat_grads=True)); ranks start from different dense weights
    and must end identical (the rank-0 broadcast DDP would have done).
    "flat-halves": the same with the pooled exchange and the head segment in two half-batches per step."""
    W = 2
    ret = ResultStore()
    mp.spawn(_e2e_worker, args=(W, _free_port(), ret, hip_graphs), nprocs=W, join=True)
    from torchrec_amd.distributed.types import ShardingEnv

    dev = torch.device("cuda", 0)
    keys, model, opt = _e2e_model(ShardingEnv.from_local(1, 0), dev, dp_max_rows=0)
    _e2e_init_tables(model)
    losses1, tabs1, dense1 = _e2e_run(model, opt, keys, _e2e_batches(W), 0, 1, dev)
    l0, t0, d0 = ret[0]
    l1, t1, d1 = ret[1]
    # mean loss over the global batch = mean of the two ranks' local means
    np.testing.assert_allclose((np.array(l0) + np.array(l1)) / 2, np.array(losses1), rtol=2e-4, atol=2e-5)
    for k in d0:
        np.testing.assert_array_equal(d0[k], d1[k])  # DDP keeps the replicas identical
        np.testing.assert_allclose(d0[k], dense1[k], rtol=2e-3, atol=2e-5)
    seen = {n: 0 for n in tabs1}
    for tabs in (t0, t1):
        for n, (w, r0) in tabs.items():
            np.testing.assert_allclose(w, tabs1[n][0][r0:r0 + w.shape[0]], rtol=2e-3, atol=2e-5)
            seen[n] += w.shape[0]
    for n in tabs1:
        t = int(n[1:])
        assert seen[n] in (E_ROWS[t], 2 * E_ROWS[t])


# ---- sequence (unpooled) embeddings, table-wise + row-wise, two ranks on one GPU -------------------------

def _seq_gpu_worker(rank, W, port, ret, row_wise):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    try:
        _stage_a2a_through_host()
        from torchrec_amd.distributed.embedding import ShardedEmbeddingCollection
        from torchrec_amd.distributed.types import ParameterSharding, ShardingEnv
        from torchrec_amd.modules.embedding_configs import EmbeddingConfig
        from torchrec_amd.sparse.jagged_tensor import KeyedJaggedTensor

        rows, Dd, B = [30, 11, 19], 4, 5  # the shapes tests/test_sharded_gloo.py::_check_seq expects
        keys = ["a", "b", "c"]
        cfgs = [EmbeddingConfig(name=f"t{i}", embedding_dim=Dd, num_embeddings=rows[i], feature_names=[keys[i]]) for i in range(3)]
        plan = {"t0": ParameterSharding("table_wise", "batched_fused", [1]),
                "t1": ParameterSharding("table_wise", "batched_fused", [0]),
                "t2": ParameterSharding("table_wise", "batched_fused", [1])}
        for n in row_wise:
            plan[n] = ParameterSharding("row_wise", "batched_fused", list(range(W)))
        sec = ShardedEmbeddingCollection(cfgs, plan, ShardingEnv.from_process_group(dist.group.WORLD), {"learning_rate": 0.5}, dev)
        init = [np.random.default_rng(100 + t).standard_normal((rows[t], Dd)).astype(np.float32) for t in range(3)]
        r0 = sec.local_shard_row_offsets()
        for name, w in sec.local_shards().items():
            w.copy_(torch.from_numpy(init[int(name[1:])][r0[name]:r0[name] + w.shape[0]]))
        rng = np.random.default_rng(7 + rank)
        lengths = rng.integers(0, 4, size=3 * B).astype(np.int32)
        vals = np.concatenate([rng.integers(0, rows[f], size=int(lengths[f * B:(f + 1) * B].sum())) for f in range(3)]).astype(np.int64)
        kjt = KeyedJaggedTensor.from_lengths_sync(keys, torch.from_numpy(vals).to(dev), torch.from_numpy(lengths).to(dev))
        out = sec(kjt).wait()
        embs = {k: out[k].values() for k in keys}
        cat = torch.cat([embs[k] for k in keys])
        g = np.random.default_rng(70 + rank).standard_normal(tuple(cat.shape)).astype(np.float32)
        cat.backward(torch.from_numpy(g).to(dev))
        torch.cuda.synchronize()
        ret[rank] = ({k: embs[k].detach().cpu().numpy().copy() for k in keys}, lengths, vals, g,
                     {n: (w.detach().cpu().numpy().copy(), r0[n]) for n, w in sec.local_shards().items()})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("row_wise", [(), ("t0", "t2")])
def test_sharded_sequence_embedding_world2_on_one_gpu(row_wise):
    from test_sharded_gloo import _check_seq

    W = 2
    ret = ResultStore()
    mp.spawn(_seq_gpu_worker, args=(W, _free_port(), ret, row_wise), nprocs=W, join=True)
    _check_seq(ret, W)


# ---- a prefetched lookup must not survive a load_state_dict between two steps ---------------------------------------

def _reload_worker(rank, port, prefetch, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from torchrec_amd.distributed.comm import init_rccl_process_group

    init_rccl_process_group(dev, rank=0, world_size=1)
    try:
        import torchrec_amd.distributed.embeddingbag as eb
        from torchrec_amd.datasets.random import Batch
        from torchrec_amd.distributed.train_pipeline import TrainPipelineSparseDist
        from torchrec_amd.distributed.types import ShardingEnv
        from torchrec_amd.sparse.jagged_tensor import KeyedJaggedTensor

        eb.FORCE_EXCHANGE = True
        keys, model, opt = _e2e_model(ShardingEnv.from_process_group(dist.group.WORLD), dev, dp_max_rows=10,
                                      graph_batch=E_B, flat=True)
        _e2e_init_tables(model)
        def plain(sd):  # sharded tables arrive as ShardedTensors over the local shard: take the shard
            return {k: (v.local_shards()[0].tensor if hasattr(v, "local_shards") else v).detach().clone()
                    for k, v in sd.items() if torch.is_tensor(v) or hasattr(v, "local_shards")}

        start = plain(model.state_dict())
        bl = [Batch(torch.from_numpy(d[:E_B]).to(dev),
                    KeyedJaggedTensor.from_fixed_lengths(keys, torch.from_numpy(np.ascontiguousarray(i[:, :E_B]).reshape(-1)).to(dev),
                                                         [1] * len(keys)),
                    torch.from_numpy(lab[:E_B]).to(dev)) for d, i, lab in _e2e_batches(1)]
        pipe = TrainPipelineSparseDist(model, opt, dev, hip_graphs=True, prefetch_lookup=prefetch)
        model.train()
        it = iter(bl)
        losses = [float(pipe.progress(it)[0].detach()) for _ in range(2)]
        model.load_state_dict(start)  # back to the initial model: the next step must see THESE tables
        losses += [float(pipe.progress(it)[0].detach()) for _ in range(2)]
        torch.cuda.synchronize()
        ret[0] = (losses, int(getattr(model.module, "prefetched_lookups", 0)),
                  {k: v.cpu().numpy() for k, v in plain(model.state_dict()).items()})
    finally:
        dist.destroy_process_group()


def test_load_state_dict_between_steps_discards_the_prefetched_lookup():
    """The default explicit step prefetches the next batch's lookup at the end of a step.  A load_state_dict before the
    next step rewrites the tables that lookup read: it must be redone (ExplicitLookupStep.epoch), so that the run equals
    one without prefetch bit for bit."""
    out = []
    for prefetch in (False, True):
        ret = ResultStore()
        mp.spawn(_reload_worker, args=(_free_port(), prefetch, ret), nprocs=1, join=True)
        out.append(ret[0])
    (l0, n0, s0), (l1, n1, s1) = out
    assert n0 == 0 and n1 >= 2  # steps 2 and 4 used their prefetched lookups, step 3's was discarded
    assert l0 == l1
    for k in s0:
        np.testing.assert_array_equal(s0[k], s1[k], err_msg=k)
