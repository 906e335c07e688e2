"""TEST-ONLY: registers oracle-backed CPU implementations of ``torch.ops.fbgemm.*``.

The product registers the HIP dispatch key only (no CPU fallback).  Host-logic tests that run
on CPU tensors (gloo world_size-2 tests, golden-vector generation against the reference's
Python harness) need *some* CPU implementation of the index ops; this module supplies the
oracle's, from the tests' side, and is never imported by the product.
"""
import numpy as np
import torch

import _paths  # noqa: F401
import fbgemm_gpu  # noqa: F401  (defines the op schemas)
from oracle import oracle

import torchrec_amd.distributed._device_ops  # noqa: F401,E402  (defines torch.ops.tbe_hip schemas)

_lib = torch.library.Library("fbgemm", "IMPL", "CPU")
_lib2 = torch.library.Library("tbe_hip", "IMPL", "CPU")
_registered = False


def _cumsum(mode):
    def fn(t_in):
        return torch.from_numpy(oracle.cumsum(t_in.contiguous().view(-1).numpy(), mode))
    return fn


def _permute_2d(permute, lengths, values, weights=None, permuted_lengths_sum=None):
    l, v, w = oracle.permute_2d(permute.numpy(), lengths.numpy(), values.contiguous().numpy(),
                                weights.contiguous().numpy() if weights is not None else None)
    return (torch.from_numpy(l), torch.from_numpy(v), torch.from_numpy(w) if w is not None else None)


def _bucketize(lengths, indices, bucketize_pos, sequence, block_sizes, my_size, weights=None):
    nl, ni, nw, npos, unb = oracle.block_bucketize(
        lengths.numpy(), indices.numpy(), block_sizes.numpy(), my_size,
        weights.numpy() if weights is not None else None, bucketize_pos, sequence)
    t = lambda a: torch.from_numpy(a) if a is not None else None  # noqa: E731
    return t(nl), t(ni), t(nw), t(npos), t(unb)


def _offsets_range(offsets, range_size):
    return torch.from_numpy(oracle.offsets_range(offsets.numpy().astype(np.int64), range_size)).to(offsets.dtype)


def _jagged_2d_to_dense(values, offsets, max_sequence_length):
    return torch.from_numpy(oracle.jagged_2d_to_dense(values.detach().numpy(), offsets.numpy().astype(np.int64),
                                                      max_sequence_length))


def _x_unpack(recv, feat_out_col, feat_src, feat_slab_col, slab_offset, slab_stride, B_local, D_total, vec, scale):
    return torch.from_numpy(oracle.pooled_exchange(recv.numpy(), feat_out_col.numpy(), feat_src.numpy(),
                                                   feat_slab_col.numpy(), slab_offset.numpy(), slab_stride.numpy(),
                                                   B_local, False, scale))


def _x_unpack_into(recv, feat_out_col, feat_src, feat_slab_col, slab_offset, slab_stride, B_local, D_total, vec, scale, out):
    out.copy_(_x_unpack(recv, feat_out_col, feat_src, feat_slab_col, slab_offset, slab_stride, B_local, D_total, vec, scale))
    return out


def _x_pack(grad, feat_out_col, feat_src, feat_slab_col, slab_offset, slab_stride, numel, vec, scale):
    return torch.from_numpy(oracle.pooled_exchange(grad.contiguous().numpy(), feat_out_col.numpy(), feat_src.numpy(),
                                                   feat_slab_col.numpy(), slab_offset.numpy(), slab_stride.numpy(),
                                                   grad.shape[0], True, scale, numel))


def _s_unpack(recv, dims, B_local, D_total, vec, scale):
    return torch.from_numpy(oracle.a2a_pooled_unpack(recv.numpy(), dims.numpy(), B_local, scale))


def _s_pack(grad, dims, vec, scale):
    return torch.from_numpy(oracle.a2a_pooled_pack(grad.contiguous().numpy(), dims.numpy(), scale))


def register() -> None:
    global _registered
    if _registered:
        return
    _lib.impl("asynchronous_complete_cumsum", _cumsum(0))
    _lib.impl("asynchronous_inclusive_cumsum", _cumsum(1))
    _lib.impl("asynchronous_exclusive_cumsum", _cumsum(2))
    _lib.impl("permute_2D_sparse_data", _permute_2d)
    _lib.impl("block_bucketize_sparse_features", _bucketize)
    _lib.impl("offsets_range", _offsets_range)
    _lib.impl("jagged_2d_to_dense", _jagged_2d_to_dense)
    _lib2.impl("pooled_exchange_unpack", _x_unpack)
    _lib2.impl("pooled_exchange_unpack_into", _x_unpack_into)
    _lib2.impl("pooled_exchange_pack", _x_pack)
    _lib2.impl("a2a_pooled_unpack", _s_unpack)
    _lib2.impl("a2a_pooled_pack", _s_pack)
    _registered = True
