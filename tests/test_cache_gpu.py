"""EmbeddingLocation.MANAGED_CACHING: host-resident tables behind the 64-way HBM row cache
(csrc/tbe_cache.hip; reference surface: torchrec/distributed/embedding_types.py:57-76
`batched_fused_uvm_caching`, batched_embedding_kernel.py:563 `flush()`).

The cache is transparent: forward outputs, updated rows and optimizer state must equal those of
the oracle and those of the same module with DEVICE tables (to rounding: rows that receive several
contributions are summed in chunks whose boundaries depend on the sort-key layout), whatever the cache
size — including caches far smaller than a batch, where rows go through eviction and the
per-batch staging area.  fbgemm's own cache implementation is absent from the reference tree
(parity unpinned beyond this transparency contract)."""
import copy

import numpy as np
import pytest
import torch

import _paths  # noqa: F401
from _util import make_inputs, to_dev
from oracle import oracle

pytestmark = pytest.mark.gpu


def _modules(rows, dims, locs, opt, ftm=None, pooling=None, **kw):
    from fbgemm_gpu.split_table_batched_embeddings_ops import (
        ComputeDevice, EmbeddingLocation, PoolingMode, SplitTableBatchedEmbeddingBagsCodegen)

    dev = torch.device("cuda", 0)
    pooling = PoolingMode.SUM if pooling is None else pooling

    def build(locations):
        return SplitTableBatchedEmbeddingBagsCodegen(
            [(r, d, loc, ComputeDevice.CUDA) for r, d, loc in zip(rows, dims, locations)], feature_table_map=ftm,
            device=dev, optimizer=opt, pooling_mode=pooling, learning_rate=0.1, eps=1e-3, **kw)

    return build(locs), build([EmbeddingLocation.DEVICE] * len(rows))


@pytest.mark.parametrize("cache_sets", [1, 3, 0])  # 64 slots, 192 slots, default load factor 0.2
@pytest.mark.parametrize("optname", ["sgd", "rowwise_adagrad"])
def test_cached_tables_match_oracle_and_device_tables(cache_sets, optname):
    from fbgemm_gpu.split_embedding_configs import EmbOptimType
    from fbgemm_gpu.split_table_batched_embeddings_ops import EmbeddingLocation as L

    rng = np.random.default_rng(4 + cache_sets)
    rows, dims = [5000, 300, 70000, 11], [128, 128, 128, 128]
    ftm = [0, 1, 2, 3, 0]  # two features share the first cached table
    locs = [L.MANAGED_CACHING, L.DEVICE, L.MANAGED_CACHING, L.MANAGED]
    opt = EmbOptimType.EXACT_SGD if optname == "sgd" else EmbOptimType.EXACT_ROWWISE_ADAGRAD
    ocode = oracle.OPT_EXACT_SGD if optname == "sgd" else oracle.OPT_EXACT_ROWWISE_ADAGRAD
    cached, plain = _modules(rows, dims, locs, opt, ftm=ftm, cache_sets=cache_sets)
    assert cached._cache is not None and plain._cache is None
    tabs = oracle.Tables(rows, dims, ftm)
    wc, wp = cached.split_embedding_weights(), plain.split_embedding_weights()
    for t in range(len(rows)):
        init = rng.standard_normal((rows[t], dims[t])).astype(np.float32)
        tabs.weights[t][...] = init
        wc[t].copy_(torch.from_numpy(init))
        wp[t].copy_(torch.from_numpy(init))
    assert not wc[0].is_cuda and wc[1].is_cuda
    s0 = [np.zeros(r, dtype=np.float32) for r in rows]
    B = 192
    for step in range(7):
        indices, offsets, _ = make_inputs(rng, rows, B, 3, ftm=ftm, zipf=(step % 2 == 0))
        if step == 3:
            indices[::17] = 10 ** 9  # out-of-range ids contribute zero rows and are counted
        out_c = cached(to_dev(indices), to_dev(offsets))
        out_p = plain(to_dev(indices), to_dev(offsets))
        ref, _ = oracle.tbe_forward(tabs, indices, offsets)
        torch.testing.assert_close(out_c, out_p, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(out_c.detach().cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
        grad = rng.standard_normal(tuple(out_c.shape)).astype(np.float32)
        out_c.backward(to_dev(grad))
        out_p.backward(to_dev(grad))
        oracle.tbe_backward(tabs, indices, offsets, grad, ocode, 0.1, eps=1e-3, state0=s0)
    st = cached.cache_stats()
    assert st["hits"] + st["misses"] > 0 and st["misses"] > 0
    if cache_sets == 1:
        assert st["evictions"] > 0, st
    cached.flush()  # write-back, cache stays warm
    torch.cuda.synchronize()
    host = [cached._table_view(cached._flat_weights, t).cpu().numpy().copy() for t in range(len(rows))]
    for t in range(len(rows)):
        np.testing.assert_allclose(host[t], plain.split_embedding_weights()[t].cpu().numpy(), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(host[t], tabs.weights[t], rtol=3e-5, atol=3e-5)
    if optname == "rowwise_adagrad":
        for t, (sc, sp) in enumerate(zip(cached.split_optimizer_states(), plain.split_optimizer_states())):
            np.testing.assert_allclose(sc[0].cpu().numpy(), sp[0].cpu().numpy(), rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(sc[0].cpu().numpy(), s0[t], rtol=3e-5, atol=3e-5)
    assert cached.bounds_check_errors() == plain.bounds_check_errors() > 0


def test_cache_survives_host_writes_staging_growth_and_deepcopy():
    """split_embedding_weights() empties the cache (the caller may write the views, as
    batched_embedding_kernel.py:541-544 / embedding_lookup.py:70 do); a larger batch grows the staging
    area without losing cached rows; a deep copy of the module (model_parallel.py:294-298) keeps working."""
    from fbgemm_gpu.split_embedding_configs import EmbOptimType
    from fbgemm_gpu.split_table_batched_embeddings_ops import EmbeddingLocation as L

    rng = np.random.default_rng(9)
    rows, dims = [4000, 900], [64, 64]
    cached, plain = _modules(rows, dims, [L.MANAGED_CACHING, L.MANAGED_CACHING], EmbOptimType.EXACT_SGD, cache_sets=4)

    def load(mods, seed):
        r = np.random.default_rng(seed)
        for t in range(len(rows)):
            init = r.standard_normal((rows[t], dims[t])).astype(np.float32)
            for m in mods:
                m.split_embedding_weights()[t].copy_(torch.from_numpy(init))

    def step(mods, B):
        indices, offsets, _ = make_inputs(rng, rows, B, 2)
        outs = [m(to_dev(indices), to_dev(offsets)) for m in mods]
        grad = to_dev(rng.standard_normal(tuple(outs[0].shape)).astype(np.float32))
        for o in outs:
            o.backward(grad)
        for o in outs[1:]:
            torch.testing.assert_close(outs[0], o, rtol=1e-5, atol=1e-5)

    load([cached, plain], 1)
    step([cached, plain], 100)
    cap0 = cached._cache.staging_cap
    step([cached, plain], 3000)  # ids per batch > staging capacity -> grows, cached rows kept
    assert cached._cache.staging_cap > cap0
    load([cached, plain], 2)     # host rewrite: stale cached rows must not come back
    step([cached, plain], 100)
    twin = copy.deepcopy(cached)
    step([cached, plain, twin], 200)
    for a, b, c in zip(cached.split_embedding_weights(), plain.split_embedding_weights(), twin.split_embedding_weights()):
        torch.testing.assert_close(a.cpu(), b.cpu(), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(a.cpu(), c.cpu(), rtol=1e-5, atol=1e-5)


def test_one_outstanding_training_forward():
    from fbgemm_gpu.split_embedding_configs import EmbOptimType
    from fbgemm_gpu.split_table_batched_embeddings_ops import EmbeddingLocation as L

    rng = np.random.default_rng(2)
    cached, _ = _modules([500], [32], [L.MANAGED_CACHING], EmbOptimType.EXACT_SGD)
    indices, offsets, _ = make_inputs(rng, [500], 16, 2)
    out = cached(to_dev(indices), to_dev(offsets))
    with pytest.raises(RuntimeError, match="one outstanding training forward"):
        cached(to_dev(indices), to_dev(offsets))
    out.sum().backward()
    with torch.no_grad():  # forward-only calls never block
        cached(to_dev(indices), to_dev(offsets))
        cached(to_dev(indices), to_dev(offsets))


def test_unpooled_lookup_through_the_cache():
    """PoolingMode.NONE (sequence embeddings, batched_embedding_kernel.py:385-456) with cached tables."""
    from fbgemm_gpu.split_embedding_configs import EmbOptimType
    from fbgemm_gpu.split_table_batched_embeddings_ops import EmbeddingLocation as L
    from fbgemm_gpu.split_table_batched_embeddings_ops import PoolingMode

    rng = np.random.default_rng(12)
    rows, dims = [3000, 800], [64, 64]
    cached, plain = _modules(rows, dims, [L.MANAGED_CACHING, L.DEVICE], EmbOptimType.EXACT_SGD, pooling=PoolingMode.NONE,
                             cache_sets=2)
    for t in range(2):
        init = rng.standard_normal((rows[t], dims[t])).astype(np.float32)
        cached.split_embedding_weights()[t].copy_(torch.from_numpy(init))
        plain.split_embedding_weights()[t].copy_(torch.from_numpy(init))
    for _ in range(3):
        indices, offsets, _ = make_inputs(rng, rows, 150, 4)
        a, b = cached(to_dev(indices), to_dev(offsets)), plain(to_dev(indices), to_dev(offsets))
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5)
        g = to_dev(rng.standard_normal(tuple(a.shape)).astype(np.float32))
        a.backward(g)
        b.backward(g)
    for x, y in zip(cached.split_embedding_weights(), plain.split_embedding_weights()):
        torch.testing.assert_close(x.cpu(), y.cpu(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_cache_stress_random_shapes(seed):
    """Many steps with random batch sizes / pooling factors / id skew over tiny caches: every step's output and
    the final tables must track the uncached module (exercises the lock-free way claiming, eviction write-back,
    staging growth and re-insertion of evicted rows under heavy set conflicts)."""
    from fbgemm_gpu.split_embedding_configs import EmbOptimType
    from fbgemm_gpu.split_table_batched_embeddings_ops import EmbeddingLocation as L

    rng = np.random.default_rng(100 + seed)
    rows, dims = [int(rng.integers(200, 20000)), int(rng.integers(50, 5000))], [64, 64]
    sets = int(rng.choice([1, 2, 5]))
    opt = [EmbOptimType.EXACT_SGD, EmbOptimType.EXACT_ROWWISE_ADAGRAD][seed % 2]
    cached, plain = _modules(rows, dims, [L.MANAGED_CACHING, L.MANAGED_CACHING], opt, cache_sets=sets)
    for t in range(2):
        init = rng.standard_normal((rows[t], dims[t])).astype(np.float32)
        cached.split_embedding_weights()[t].copy_(torch.from_numpy(init))
        plain.split_embedding_weights()[t].copy_(torch.from_numpy(init))
    for step in range(25):
        B = int(rng.integers(1, 400))
        indices, offsets, _ = make_inputs(rng, rows, B, int(rng.integers(1, 5)), zipf=bool(step % 3))
        a, b = cached(to_dev(indices), to_dev(offsets)), plain(to_dev(indices), to_dev(offsets))
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5, msg=lambda m: f"step {step}: {m}")
        g = to_dev(rng.standard_normal(tuple(a.shape)).astype(np.float32))
        a.backward(g)
        b.backward(g)
    st = cached.cache_stats()
    assert st["evictions"] > 0 and st["hits"] > 0
    for x, y in zip(cached.split_embedding_weights(), plain.split_embedding_weights()):
        torch.testing.assert_close(x.cpu(), y.cpu(), rtol=1e-4, atol=1e-5)
