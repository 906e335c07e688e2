"""numpy front-end of the CPU oracle (``oracle/tbe_oracle.c``).

TEST INFRASTRUCTURE ONLY — imported by tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``; never by the product package.  See the header of
``tbe_oracle.c`` for which reference file:line each function follows and which results are
"parity unpinned".
"""
import ctypes
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

POOL_SUM, POOL_MEAN, POOL_NONE = 0, 1, 2
OPT_EXACT_SGD, OPT_EXACT_ROWWISE_ADAGRAD, OPT_ADAM, OPT_EXACT_ADAGRAD, OPT_DENSE_GRAD = 0, 1, 2, 3, 100


def build() -> str:
    subprocess.run(["make", "-s", "-C", _HERE], check=True)
    return _SO


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        srcs = [os.path.join(_HERE, f) for f in ("tbe_oracle.c", "dlrm_oracle.c")]
        if not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_tbe_forward_pooled.restype = ctypes.c_int64
        _lib.oracle_tbe_forward_nobag.restype = ctypes.c_int64
        _lib.oracle_tbe_backward.restype = ctypes.c_int64
        _lib.oracle_permute_2d.restype = ctypes.c_int64
    return _lib


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=dtype)


class Tables:
    """Feature metadata + storage for the oracle (same meaning as the feat_* arrays of
    include/tbe_hip.h)."""

    def __init__(self, rows: Sequence[int], dims: Sequence[int],
                 feature_table_map: Optional[Sequence[int]] = None):
        self.rows = [int(r) for r in rows]
        self.dims = [int(d) for d in dims]
        self.ftm = list(feature_table_map) if feature_table_map is not None else list(range(len(rows)))
        self.F = len(self.ftm)
        self.weights: List[np.ndarray] = [np.zeros((r, d), dtype=np.float32) for r, d in zip(self.rows, self.dims)]
        self.feat_D = np.array([self.dims[t] for t in self.ftm], dtype=np.int32)
        self.feat_D_offset = np.concatenate([[0], np.cumsum(self.feat_D)]).astype(np.int32)
        self.feat_rows = np.array([self.rows[t] for t in self.ftm], dtype=np.int64)
        base = np.concatenate([[0], np.cumsum(self.rows)]).astype(np.int64)
        self.feat_row_base = np.array([base[t] for t in self.ftm], dtype=np.int64)
        self.total_D = int(self.feat_D_offset[-1])

    def ptr_array(self, arrays: Sequence[np.ndarray]):
        arr = (ctypes.c_void_p * self.F)()
        for f, t in enumerate(self.ftm):
            arr[f] = arrays[t].ctypes.data
        return arr


def tbe_forward(tables: Tables, indices, offsets, per_sample_weights=None,
                pooling_mode: int = POOL_SUM) -> Tuple[np.ndarray, int]:
    indices = _c(indices, np.int64)
    offsets = _c(offsets, np.int64)
    psw = _c(per_sample_weights, np.float32) if per_sample_weights is not None else None
    B = (offsets.size - 1) // tables.F
    wp = tables.ptr_array(tables.weights)
    if pooling_mode == POOL_NONE:
        D = tables.dims[0]
        out = np.zeros((indices.size, D), dtype=np.float32)
        bad = lib().oracle_tbe_forward_nobag(wp, _p(tables.feat_rows), tables.F, B, D, _p(indices),
                                             _p(offsets), _p(out))
        return out, int(bad)
    out = np.zeros((B, tables.total_D), dtype=np.float32)
    bad = lib().oracle_tbe_forward_pooled(wp, _p(tables.feat_D), _p(tables.feat_D_offset),
                                          _p(tables.feat_rows), tables.F, B, _p(indices), _p(offsets),
                                          _p(psw), pooling_mode, _p(out), ctypes.c_int64(tables.total_D))
    return out, int(bad)


def tbe_backward(tables: Tables, indices, offsets, grad_out, optimizer: int, lr: float,
                 per_sample_weights=None, pooling_mode: int = POOL_SUM, eps: float = 1e-8,
                 weight_decay: float = 0.0, beta1: float = 0.9, beta2: float = 0.999,
                 iteration: int = 1, state0: Optional[List[np.ndarray]] = None,
                 state1: Optional[List[np.ndarray]] = None) -> int:
    """Updates tables.weights (and the state arrays) in place; returns the bad-index count."""
    indices = _c(indices, np.int64)
    offsets = _c(offsets, np.int64)
    grad_out = _c(grad_out, np.float32)
    psw = _c(per_sample_weights, np.float32) if per_sample_weights is not None else None
    B = (offsets.size - 1) // tables.F
    hyper = np.array([lr, eps, weight_decay, beta1, beta2], dtype=np.float32)
    wp = tables.ptr_array(tables.weights)
    s0 = tables.ptr_array(state0) if state0 is not None else None
    s1 = tables.ptr_array(state1) if state1 is not None else None
    bad = lib().oracle_tbe_backward(wp, _p(tables.feat_D), _p(tables.feat_D_offset),
                                    _p(tables.feat_rows), _p(tables.feat_row_base), s0, s1,
                                    tables.F, B, _p(indices), ctypes.c_int64(indices.size),
                                    _p(offsets), _p(psw), pooling_mode, _p(grad_out),
                                    ctypes.c_int64(grad_out.shape[1]), optimizer, _p(hyper),
                                    ctypes.c_int64(iteration))
    return int(bad)


def cumsum(x, mode: int = 0) -> np.ndarray:
    x = np.ascontiguousarray(x)
    assert x.dtype in (np.int32, np.int64)
    out = np.zeros(x.size + 1 if mode == 0 else x.size, dtype=x.dtype)
    fn = lib().oracle_cumsum_i32 if x.dtype == np.int32 else lib().oracle_cumsum_i64
    fn(_p(x), _p(out), ctypes.c_int64(x.size), mode)
    return out


def permute_2d(permute, lengths, values, weights=None):
    """lengths [T, B]; returns (permuted_lengths [T', B], values', weights')."""
    lengths = np.asarray(lengths)
    T_in, B = lengths.shape
    perm = _c(permute, np.int32)
    T_out = perm.size
    l64 = _c(lengths, np.int64)
    values = np.ascontiguousarray(values)
    weights = np.ascontiguousarray(weights) if weights is not None else None
    out_l = np.zeros((T_out, B), dtype=np.int64)
    total = lib().oracle_permute_2d(_p(perm), T_in, T_out, B, _p(l64), _p(out_l), None, None, 0, None, None, 0)
    out_v = np.zeros(total, dtype=values.dtype)
    out_w = np.zeros(total, dtype=weights.dtype) if weights is not None else None
    lib().oracle_permute_2d(_p(perm), T_in, T_out, B, _p(l64), _p(out_l), _p(values), _p(out_v),
                            values.itemsize, _p(weights), _p(out_w),
                            weights.itemsize if weights is not None else 0)
    return out_l.astype(lengths.dtype), out_v, out_w


def block_bucketize(lengths, indices, block_sizes, my_size: int, weights=None,
                    bucketize_pos: bool = False, sequence: bool = False):
    lengths_a = np.asarray(lengths)
    indices_a = np.asarray(indices)
    l64 = _c(lengths_a.reshape(-1), np.int64)
    i64 = _c(indices_a.reshape(-1), np.int64)
    b64 = _c(np.asarray(block_sizes).reshape(-1), np.int64)
    w = _c(weights, np.float32) if weights is not None else None
    F = b64.size
    nl = np.zeros(l64.size * my_size, dtype=np.int64)
    ni = np.zeros(i64.size, dtype=np.int64)
    nw = np.zeros(i64.size, dtype=np.float32) if w is not None else None
    npos = np.zeros(i64.size, dtype=np.int64) if bucketize_pos else None
    unb = np.zeros(i64.size, dtype=np.int64) if sequence else None
    lib().oracle_block_bucketize(_p(l64), ctypes.c_int64(l64.size), _p(i64), _p(b64), F, my_size, _p(w),
                                 _p(nl), _p(ni), _p(nw), _p(npos), _p(unb))
    return (nl.astype(lengths_a.dtype), ni.astype(indices_a.dtype), nw,
            npos.astype(indices_a.dtype) if npos is not None else None,
            unb.astype(indices_a.dtype) if unb is not None else None)


def a2a_pooled_unpack(recv, dims, B_local: int, scale: float = 1.0) -> np.ndarray:
    dims = _c(dims, np.int32)
    recv = _c(recv, np.float32)
    D_total = int(dims.sum())
    out = np.zeros((B_local, D_total), dtype=np.float32)
    lib().oracle_a2a_pooled_unpack(_p(recv), _p(out), _p(dims), dims.size, B_local, D_total, ctypes.c_float(scale))
    return out


def a2a_pooled_pack(grad, dims, scale: float = 1.0) -> np.ndarray:
    dims = _c(dims, np.int32)
    grad = _c(grad, np.float32)
    B_local, D_total = grad.shape
    send = np.zeros(B_local * D_total, dtype=np.float32)
    lib().oracle_a2a_pooled_pack(_p(grad), _p(send), _p(dims), dims.size, B_local, D_total, ctypes.c_float(scale))
    return send


def jagged_2d_to_dense(values, offsets, max_L: int) -> np.ndarray:
    values = _c(values, np.float32)
    offsets = _c(offsets, np.int64)
    B = offsets.size - 1
    D = values.shape[1]
    dense = np.zeros((B, max_L, D), dtype=np.float32)
    lib().oracle_jagged_2d_to_dense(_p(values), _p(offsets), B, D, max_L, _p(dense))
    return dense


def interaction_forward(dense, sparse) -> np.ndarray:
    """dense [B, D], sparse [B, F, D] -> [B, D + (F+1)F/2] (oracle/dlrm_oracle.c)."""
    dense = _c(dense, np.float32)
    sparse = _c(sparse, np.float32)
    B, F, D = sparse.shape
    out = np.zeros((B, D + (F + 1) * F // 2), dtype=np.float32)
    lib().oracle_interaction_forward(_p(dense), _p(sparse), B, F, D, _p(out))
    return out


def interaction_backward(dense, sparse, grad_out) -> Tuple[np.ndarray, np.ndarray]:
    dense = _c(dense, np.float32)
    sparse = _c(sparse, np.float32)
    grad_out = _c(grad_out, np.float32)
    B, F, D = sparse.shape
    gd = np.zeros((B, D), dtype=np.float32)
    gs = np.zeros((B, F, D), dtype=np.float32)
    lib().oracle_interaction_backward(_p(dense), _p(sparse), _p(grad_out), B, F, D, _p(gd), _p(gs))
    return gd, gs


def offsets_range(offsets, range_size: int) -> np.ndarray:
    offsets = _c(offsets, np.int64)
    out = np.zeros(range_size, dtype=np.int64)
    lib().oracle_offsets_range(_p(offsets), ctypes.c_int64(offsets.size), ctypes.c_int64(range_size), _p(out))
    return out


def pooled_exchange(buf_or_mat, feat_out_col, feat_src, feat_slab_col, slab_offset, slab_stride, B_local: int,
                    pack: bool, scale: float = 1.0, buf_numel: int = 0) -> np.ndarray:
    """unpack: buf [exchange buffer] -> [B_local, D_total]; pack: matrix -> exchange buffer."""
    col = _c(feat_out_col, np.int32)
    src = _c(feat_src, np.int32)
    scol = _c(feat_slab_col, np.int32)
    soff = _c(slab_offset, np.int64)
    sstr = _c(slab_stride, np.int32)
    Fg, W, D_total = src.size, soff.size, int(col[-1])
    x = _c(buf_or_mat, np.float32)
    if pack:
        out = np.zeros(buf_numel, dtype=np.float32)
        lib().oracle_pooled_exchange_pack(_p(x), _p(out), _p(col), _p(src), _p(scol), _p(soff), _p(sstr), Fg, W,
                                          B_local, D_total, ctypes.c_float(scale))
    else:
        out = np.zeros((B_local, D_total), dtype=np.float32)
        lib().oracle_pooled_exchange_unpack(_p(x), _p(out), _p(col), _p(src), _p(scol), _p(soff), _p(sstr), Fg, W,
                                            B_local, D_total, ctypes.c_float(scale))
    return out
