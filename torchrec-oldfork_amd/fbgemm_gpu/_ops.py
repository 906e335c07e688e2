"""``torch.ops.fbgemm.*`` — the dispatcher-op half of the drop-in boundary.

Schemas are the ones the reference calls:
  * asynchronous_complete_cumsum        torchrec/sparse/jagged_tensor.py:35-36
  * permute_2D_sparse_data              torchrec/sparse/jagged_tensor.py:946-952,
                                        torchrec/distributed/dist_data.py:257-263,
                                        torchrec/distributed/comm_ops.py:633-639, 691-697
  * block_bucketize_sparse_features     torchrec/distributed/embedding_sharding.py:160-168
  * offsets_range                       torchrec/modules/feature_processor.py:65
  * jagged_2d_to_dense                  examples/bert4rec/models/bert4rec.py:394-400
Only the CUDA (= HIP on ROCm) dispatch key is registered: a CPU tensor raises
``NotImplementedError`` from the dispatcher — there is no CPU fallback in the product.
"""
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import check, ptr, require_gpu, stream_ptr, workspace

_DEFS = [
    "asynchronous_complete_cumsum(Tensor t_in) -> Tensor",
    "asynchronous_inclusive_cumsum(Tensor t_in) -> Tensor",
    "asynchronous_exclusive_cumsum(Tensor t_in) -> Tensor",
    "permute_2D_sparse_data(Tensor permute, Tensor lengths, Tensor values, Tensor? weights=None, "
    "int? permuted_lengths_sum=None) -> (Tensor, Tensor, Tensor?)",
    "permute_1D_sparse_data(Tensor permute, Tensor lengths, Tensor values, Tensor? weights=None, "
    "int? permuted_lengths_sum=None) -> (Tensor, Tensor, Tensor?)",
    "expand_into_jagged_permute(Tensor permute, Tensor input_offset, Tensor output_offset, "
    "int output_size) -> Tensor",
    "block_bucketize_sparse_features(Tensor lengths, Tensor indices, bool bucketize_pos, "
    "bool sequence, Tensor block_sizes, int my_size, Tensor? weights=None) -> "
    "(Tensor, Tensor, Tensor?, Tensor?, Tensor?)",
    "offsets_range(Tensor offsets, int range_size) -> Tensor",
    "jagged_2d_to_dense(Tensor values, Tensor offsets, int max_sequence_length) -> Tensor",
]

_def_lib = torch.library.Library("fbgemm", "DEF")
for _d in _DEFS:
    _def_lib.define(_d)
_impl_lib = torch.library.Library("fbgemm", "IMPL", "CUDA")


def _cumsum(t_in: torch.Tensor, mode: int) -> torch.Tensor:
    dev = require_gpu(t_in)
    if t_in.dtype not in (torch.int32, torch.int64):
        raise RuntimeError(f"cumsum: dtype {t_in.dtype} not supported (int32/int64)")
    x = t_in.contiguous().view(-1)
    n = x.numel()
    out = torch.empty(n + 1 if mode == 0 else n, dtype=x.dtype, device=dev)
    lib = _lib.load()
    ws_bytes = lib.tbe_cumsum_workspace_bytes(n)
    ws = workspace(ws_bytes, dev)
    with torch.cuda.device(dev):
        check(
            lib.tbe_cumsum(ptr(x), ptr(out), n, x.element_size(), mode, ptr(ws), ws.numel(),
                           stream_ptr(dev)),
            "tbe_cumsum",
        )
    return out


def asynchronous_complete_cumsum(t_in: torch.Tensor) -> torch.Tensor:
    return _cumsum(t_in, 0)


def asynchronous_inclusive_cumsum(t_in: torch.Tensor) -> torch.Tensor:
    return _cumsum(t_in, 1)


def asynchronous_exclusive_cumsum(t_in: torch.Tensor) -> torch.Tensor:
    return _cumsum(t_in, 2)


def permute_2D_sparse_data(
    permute: torch.Tensor,
    lengths: torch.Tensor,
    values: torch.Tensor,
    weights: Optional[torch.Tensor] = None,
    permuted_lengths_sum: Optional[int] = None,
) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
    dev = require_gpu(permute, lengths, values, weights)
    if lengths.dim() != 2:
        raise RuntimeError("permute_2D_sparse_data: lengths must be 2-D [T, B]")
    if lengths.dtype not in (torch.int32, torch.int64):
        raise RuntimeError(f"permute_2D_sparse_data: lengths dtype {lengths.dtype}")
    T_in, B = lengths.shape
    T_out = permute.numel()
    perm = permute.to(torch.int32).contiguous()
    lengths_c = lengths.contiguous()
    values_c = values.contiguous().view(-1)
    weights_c = weights.contiguous().view(-1) if weights is not None else None
    lib = _lib.load()
    out_lengths = torch.empty((T_out, B), dtype=lengths.dtype, device=dev)
    in_offsets = torch.empty(T_in * B + 1, dtype=torch.int64, device=dev)
    out_offsets = torch.empty(T_out * B + 1, dtype=torch.int64, device=dev)
    ws = workspace(lib.tbe_permute_2d_workspace_bytes(T_in, T_out, B), dev)
    st = stream_ptr(dev)
    with torch.cuda.device(dev):
        check(
            lib.tbe_permute_2d_lengths(ptr(perm), T_in, T_out, B, ptr(lengths_c),
                                       lengths_c.element_size(), ptr(out_lengths), ptr(in_offsets),
                                       ptr(out_offsets), ptr(ws), ws.numel(), st),
            "tbe_permute_2d_lengths",
        )
        if permuted_lengths_sum is None:
            # data-dependent output size: one D2H read, as the reference op does
            permuted_lengths_sum = int(out_offsets[-1].item())
        out_values = torch.empty(permuted_lengths_sum, dtype=values.dtype, device=dev)
        out_weights = (
            torch.empty(permuted_lengths_sum, dtype=weights_c.dtype, device=dev)
            if weights_c is not None else None
        )
        check(
            lib.tbe_permute_2d_data(ptr(perm), T_out, B, ptr(in_offsets), ptr(out_offsets),
                                    ptr(values_c), ptr(out_values), values_c.element_size(),
                                    ptr(weights_c), ptr(out_weights),
                                    weights_c.element_size() if weights_c is not None else 4, st),
            "tbe_permute_2d_data",
        )
    return out_lengths, out_values, out_weights


def block_bucketize_sparse_features(
    lengths: torch.Tensor,
    indices: torch.Tensor,
    bucketize_pos: bool,
    sequence: bool,
    block_sizes: torch.Tensor,
    my_size: int,
    weights: Optional[torch.Tensor] = None,
) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor], Optional[torch.Tensor],
           Optional[torch.Tensor]]:
    dev = require_gpu(lengths, indices, block_sizes, weights)
    if lengths.dtype not in (torch.int32, torch.int64) or indices.dtype not in (torch.int32, torch.int64):
        raise RuntimeError("block_bucketize_sparse_features: lengths/indices must be int32 or int64")
    if block_sizes.dtype != indices.dtype:
        raise RuntimeError("block_bucketize_sparse_features: block_sizes dtype must match indices")
    if weights is not None and weights.dtype != torch.float32:
        raise RuntimeError("block_bucketize_sparse_features: weights must be float32")
    lengths_c = lengths.contiguous().view(-1)
    indices_c = indices.contiguous().view(-1)
    blocks_c = block_sizes.contiguous().view(-1)
    weights_c = weights.contiguous().view(-1) if weights is not None else None
    F = blocks_c.numel()
    L = lengths_c.numel()
    N = indices_c.numel()
    new_lengths = torch.empty(L * my_size, dtype=lengths.dtype, device=dev)
    new_indices = torch.empty(N, dtype=indices.dtype, device=dev)
    new_weights = torch.empty(N, dtype=torch.float32, device=dev) if weights_c is not None else None
    new_pos = torch.empty(N, dtype=indices.dtype, device=dev) if bucketize_pos else None
    unbucketize = torch.empty(N, dtype=indices.dtype, device=dev) if sequence else None
    lib = _lib.load()
    ws = workspace(lib.tbe_bucketize_workspace_bytes(L, my_size), dev)
    with torch.cuda.device(dev):
        check(
            lib.tbe_block_bucketize(ptr(lengths_c), lengths_c.element_size(), L, ptr(indices_c),
                                    indices_c.element_size(), N, ptr(blocks_c), F, my_size,
                                    ptr(weights_c), int(bucketize_pos), int(sequence),
                                    ptr(new_lengths), ptr(new_indices), ptr(new_weights),
                                    ptr(new_pos), ptr(unbucketize), ptr(ws), ws.numel(),
                                    stream_ptr(dev)),
            "tbe_block_bucketize",
        )
    return new_lengths, new_indices, new_weights, new_pos, unbucketize


def offsets_range(offsets: torch.Tensor, range_size: int) -> torch.Tensor:
    dev = require_gpu(offsets)
    offs = offsets.to(torch.int64).contiguous().view(-1)
    out = torch.empty(range_size, dtype=torch.int64, device=dev)
    lib = _lib.load()
    with torch.cuda.device(dev):
        check(lib.tbe_offsets_range(ptr(offs), offs.numel(), range_size, ptr(out), stream_ptr(dev)),
              "tbe_offsets_range")
    return out.to(offsets.dtype)


def jagged_2d_to_dense_forward(values: torch.Tensor, offsets: torch.Tensor,
                               max_sequence_length: int) -> torch.Tensor:
    dev = require_gpu(values, offsets)
    if values.dim() != 2 or values.dtype != torch.float32:
        raise RuntimeError("jagged_2d_to_dense: values must be float32 [N, D]")
    vals = values.contiguous()
    offs = offsets.to(torch.int64).contiguous().view(-1)
    B = offs.numel() - 1
    D = vals.shape[1]
    dense = torch.empty((B, max_sequence_length, D), dtype=torch.float32, device=dev)
    lib = _lib.load()
    with torch.cuda.device(dev):
        check(lib.tbe_jagged_2d_to_dense_f32(ptr(vals), ptr(offs), B, D, max_sequence_length,
                                             ptr(dense), stream_ptr(dev)),
              "tbe_jagged_2d_to_dense_f32")
    return dense


def _jagged_2d_to_dense_setup(ctx, inputs, output):
    values, offsets, max_sequence_length = inputs
    ctx.save_for_backward(offsets)
    ctx.max_L = max_sequence_length
    ctx.N, ctx.D = values.shape


def _jagged_2d_to_dense_backward(ctx, grad_dense):
    (offsets,) = ctx.saved_tensors
    if not grad_dense.is_cuda:
        raise RuntimeError("jagged_2d_to_dense backward: no CPU fallback in the MI355X build")
    dev = grad_dense.device
    g = grad_dense.contiguous().float()
    offs = offsets.to(torch.int64).contiguous().view(-1)
    B = offs.numel() - 1
    grad_values = torch.empty((ctx.N, ctx.D), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(_lib.load().tbe_dense_to_jagged_2d_f32(ptr(g), ptr(offs), B, ctx.D, ctx.max_L, ctx.N,
                                                     ptr(grad_values), stream_ptr(dev)),
              "tbe_dense_to_jagged_2d_f32")
    return grad_values, None, None


torch.library.register_autograd("fbgemm::jagged_2d_to_dense", _jagged_2d_to_dense_backward,
                                setup_context=_jagged_2d_to_dense_setup)

_impl_lib.impl("asynchronous_complete_cumsum", asynchronous_complete_cumsum)
_impl_lib.impl("asynchronous_inclusive_cumsum", asynchronous_inclusive_cumsum)
_impl_lib.impl("asynchronous_exclusive_cumsum", asynchronous_exclusive_cumsum)
_impl_lib.impl("permute_2D_sparse_data", permute_2D_sparse_data)
_impl_lib.impl("block_bucketize_sparse_features", block_bucketize_sparse_features)
_impl_lib.impl("offsets_range", offsets_range)
_impl_lib.impl("jagged_2d_to_dense", jagged_2d_to_dense_forward)
# permute_1D_sparse_data / expand_into_jagged_permute: schema only (variable-batch path,
# SURVEY.md §2a "OUT OF SCOPE"); calling them raises from the dispatcher.
