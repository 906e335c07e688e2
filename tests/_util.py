"""Helpers shared by the GPU parity tests."""
import numpy as np
import torch

import _paths  # noqa: F401
from oracle import oracle


def make_inputs(rng, rows, B, max_len, ftm=None, fixed_len=None, weighted=False, zipf=False):
    ftm = ftm if ftm is not None else list(range(len(rows)))
    F = len(ftm)
    lengths = (np.full(F * B, fixed_len, dtype=np.int64) if fixed_len is not None
               else rng.integers(0, max_len + 1, size=F * B).astype(np.int64))
    vals = []
    for f in range(F):
        n = int(lengths[f * B:(f + 1) * B].sum())
        r = rows[ftm[f]]
        if zipf:
            v = np.minimum(rng.zipf(1.2, size=n) - 1, r - 1)
        else:
            v = rng.integers(0, r, size=n)
        vals.append(v)
    indices = np.concatenate(vals).astype(np.int64) if vals else np.zeros(0, np.int64)
    offsets = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    psw = rng.random(indices.size).astype(np.float32) + 0.5 if weighted else None
    return indices, offsets, psw


def build_pair(rows, dims, ftm, pooling, optimizer=None, rng=None, dense=False, **opt_kwargs):
    """Returns (hip module on cuda:0, oracle Tables) holding identical random weights."""
    from fbgemm_gpu.split_embedding_configs import EmbOptimType
    from fbgemm_gpu.split_table_batched_embeddings_ops import (
        ComputeDevice, DenseTableBatchedEmbeddingBagsCodegen, EmbeddingLocation, PoolingMode,
        SplitTableBatchedEmbeddingBagsCodegen)

    pm = {0: PoolingMode.SUM, 1: PoolingMode.MEAN, 2: PoolingMode.NONE}[pooling]
    dev = torch.device("cuda", 0)
    if dense:
        mod = DenseTableBatchedEmbeddingBagsCodegen(list(zip(rows, dims)), feature_table_map=ftm, pooling_mode=pm)
    else:
        mod = SplitTableBatchedEmbeddingBagsCodegen(
            [(r, d, EmbeddingLocation.DEVICE, ComputeDevice.CUDA) for r, d in zip(rows, dims)],
            feature_table_map=ftm, pooling_mode=pm, device=dev,
            optimizer=optimizer if optimizer is not None else EmbOptimType.EXACT_SGD, **opt_kwargs)
    tabs = oracle.Tables(rows, dims, ftm)
    rng = rng if rng is not None else np.random.default_rng(0)
    for t, w in enumerate(mod.split_embedding_weights()):
        init = rng.standard_normal((rows[t], dims[t])).astype(np.float32)
        tabs.weights[t][...] = init
        w.copy_(torch.from_numpy(init))
    return mod, tabs


def to_dev(a, dtype=None):
    if a is None:
        return None
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def oracle_forward_mixed(tabs, indices, offsets, psw, feat_mean):
    """Oracle forward with per-FEATURE pooling: SUM and MEAN runs spliced by column block."""
    out_s, _ = oracle.tbe_forward(tabs, indices, offsets, psw, oracle.POOL_SUM)
    out_m, _ = oracle.tbe_forward(tabs, indices, offsets, psw, oracle.POOL_MEAN)
    cols = np.repeat(np.asarray(feat_mean, dtype=bool), np.asarray(tabs.feat_D))
    return np.where(cols[None, :], out_m, out_s)


def oracle_backward_mixed(tabs, indices, offsets, grad, optimizer, lr, psw, feat_mean, **kw):
    """Oracle backward with per-FEATURE pooling: one SUM pass on the gradient with the MEAN features' blocks zeroed,
    one MEAN pass on the rest (exact: a table belongs to one pooling type, a zero gradient leaves a row unchanged)."""
    cols = np.repeat(np.asarray(feat_mean, dtype=bool), np.asarray(tabs.feat_D))
    g = np.ascontiguousarray(grad, dtype=np.float32)
    oracle.tbe_backward(tabs, indices, offsets, np.ascontiguousarray(np.where(cols[None, :], 0.0, g), dtype=np.float32), optimizer, lr,
                        psw, oracle.POOL_SUM, **kw)
    oracle.tbe_backward(tabs, indices, offsets, np.ascontiguousarray(np.where(cols[None, :], g, 0.0), dtype=np.float32), optimizer, lr,
                        psw, oracle.POOL_MEAN, **kw)
