"""Device ops of the distributed layer that are not part of the fbgemm surface, exposed as
dispatcher ops (`torch.ops.tbe_hip.*`) over the C ABI (include/tbe_hip.h
tbe_pooled_exchange_*).  Only the HIP key is registered here; CPU tensors raise."""
import torch

from fbgemm_gpu import _lib
from fbgemm_gpu._lib import check, ptr, require_gpu, stream_ptr

_def = torch.library.Library("tbe_hip", "DEF")
_def.define("pooled_exchange_unpack(Tensor recv, Tensor feat_out_col, Tensor feat_src, Tensor feat_slab_col, "
            "Tensor slab_offset, Tensor slab_stride, int B_local, int D_total, bool vec, float scale) -> Tensor")
_def.define("pooled_exchange_unpack_into(Tensor recv, Tensor feat_out_col, Tensor feat_src, Tensor feat_slab_col, "
            "Tensor slab_offset, Tensor slab_stride, int B_local, int D_total, bool vec, float scale, Tensor(a!) out) -> Tensor(a!)")
_def.define("pooled_exchange_pack(Tensor grad, Tensor feat_out_col, Tensor feat_src, Tensor feat_slab_col, "
            "Tensor slab_offset, Tensor slab_stride, int numel, bool vec, float scale) -> Tensor")
_def.define("pooled_exchange_pack_into(Tensor grad, Tensor feat_out_col, Tensor feat_src, Tensor feat_slab_col, "
            "Tensor slab_offset, Tensor slab_stride, bool vec, float scale, Tensor(a!) send) -> Tensor(a!)")
_def.define("a2a_pooled_unpack(Tensor recv, Tensor dim_sum_per_rank, int B_local, int D_total, bool vec, float scale) -> Tensor")
_def.define("a2a_pooled_pack(Tensor grad, Tensor dim_sum_per_rank, bool vec, float scale) -> Tensor")
_def.define("relu_backward_bias_grad(Tensor grad_out, Tensor act) -> (Tensor, Tensor)")
_def.define("weighted_colsum(Tensor x, Tensor w) -> Tensor")
_def.define("copy_rows(Tensor src, Tensor rows) -> Tensor")
_def.define("relu_backward_bias_partials(Tensor grad_out, Tensor act) -> (Tensor, Tensor)")
_def.define("weighted_colsum_partials(Tensor x, Tensor w) -> Tensor")
_def.define("multi_chunk_sum(Tensor seg_table, int nseg, int max_numel, Tensor(a!) dst, float scale) -> Tensor(a!)")
_def.define("bce_with_logits(Tensor logits, Tensor labels) -> (Tensor, Tensor)")
_impl = torch.library.Library("tbe_hip", "IMPL", "CUDA")


def _copy_rows(src, rows):
    """src[rows] for a contiguous 2-D `src` whose rows are multiples of 16 bytes and a device int32 `rows`
    (csrc/sparse_ops.hip copy_rows_kernel)."""
    dev = require_gpu(src, rows)
    if src.dim() != 2 or not src.is_contiguous() or rows.dtype != torch.int32 or rows.dim() != 1:
        raise RuntimeError(f"copy_rows: need a contiguous 2-D src and int32 rows, got {tuple(src.shape)} / {rows.dtype}")
    out = src.new_empty((rows.numel(), src.shape[1]))
    lib = _lib.load()
    with torch.cuda.device(dev):
        check(lib.tbe_copy_rows(ptr(src), src.shape[0], ptr(rows), rows.numel(), src.shape[1] * src.element_size(), ptr(out),
                                stream_ptr(dev)), "tbe_copy_rows")
    return out


def _weighted_colsum(x, w):
    """out[c] = sum_b w[b] * x[b, c] (csrc/mlp_epilogue.hip)."""
    from fbgemm_gpu._lib import workspace

    dev = require_gpu(x, w)
    if x.dim() != 2 or w.numel() != x.shape[0] or x.dtype != torch.float32 or w.dtype != torch.float32:
        raise RuntimeError(f"weighted_colsum: need float32 x [B, N] and w [B], got {tuple(x.shape)} and {tuple(w.shape)}")
    x, w = x.contiguous(), w.contiguous().view(-1)
    B, N = x.shape
    out = torch.empty(N, dtype=torch.float32, device=dev)
    lib = _lib.load()
    with torch.cuda.device(dev):
        ws = workspace(lib.tbe_weighted_colsum_workspace_bytes(B, N), dev)
        check(lib.tbe_weighted_colsum_f32(ptr(x), ptr(w), B, N, ptr(out), ptr(ws), ws.numel(), stream_ptr(dev)),
              "tbe_weighted_colsum_f32")
    return out


def _relu_backward_bias_grad(grad_out, act):
    """(grad_out * (act > 0), its column sums) in one pass (csrc/mlp_epilogue.hip)."""
    from fbgemm_gpu._lib import workspace

    dev = require_gpu(grad_out, act)
    if grad_out.dim() != 2 or grad_out.shape != act.shape or grad_out.dtype != torch.float32 or act.dtype != torch.float32:
        raise RuntimeError(f"relu_backward_bias_grad: need two float32 [B, N] tensors of one shape, got "
                           f"{tuple(grad_out.shape)} and {tuple(act.shape)}")
    grad_out, act = grad_out.contiguous(), act.contiguous()
    B, N = grad_out.shape
    gx = torch.empty_like(grad_out)
    gb = torch.empty(N, dtype=torch.float32, device=dev)
    lib = _lib.load()
    with torch.cuda.device(dev):
        ws = workspace(lib.tbe_relu_backward_bias_grad_workspace_bytes(B, N), dev)
        check(lib.tbe_relu_backward_bias_grad_f32(ptr(grad_out), ptr(act), B, N, ptr(gx), ptr(gb), ptr(ws), ws.numel(),
                                                  stream_ptr(dev)), "tbe_relu_backward_bias_grad_f32")
    return gx, gb


def _relu_backward_bias_partials(grad_out, act):
    """(grad_out * (act > 0), column sums of its row blocks [row blocks, N]) — the first stage of relu_backward_bias_grad
    alone; `multi_chunk_sum` finishes every layer's bias gradient in one launch (csrc/mlp_epilogue.hip)."""
    dev = require_gpu(grad_out, act)
    if grad_out.dim() != 2 or grad_out.shape != act.shape or grad_out.dtype != torch.float32 or act.dtype != torch.float32:
        raise RuntimeError(f"relu_backward_bias_partials: need two float32 [B, N] tensors of one shape, got "
                           f"{tuple(grad_out.shape)} and {tuple(act.shape)}")
    grad_out, act = grad_out.contiguous(), act.contiguous()
    B, N = grad_out.shape
    lib = _lib.load()
    gx = torch.empty_like(grad_out)
    partial = torch.empty((lib.tbe_colsum_row_blocks(B, N), N), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(lib.tbe_relu_backward_bias_partials_f32(ptr(grad_out), ptr(act), B, N, ptr(gx), ptr(partial),
                                                      partial.numel() * 4, stream_ptr(dev)), "tbe_relu_backward_bias_partials_f32")
    return gx, partial


def _weighted_colsum_partials(x, w):
    """Row-block partials [row blocks, N] of out[c] = sum_b w[b] * x[b, c] (first stage of weighted_colsum)."""
    dev = require_gpu(x, w)
    if x.dim() != 2 or w.numel() != x.shape[0] or x.dtype != torch.float32 or w.dtype != torch.float32:
        raise RuntimeError(f"weighted_colsum_partials: need float32 x [B, N] and w [B], got {tuple(x.shape)} and {tuple(w.shape)}")
    x, w = x.contiguous(), w.contiguous().view(-1)
    B, N = x.shape
    lib = _lib.load()
    partial = torch.empty((lib.tbe_colsum_row_blocks(B, N), N), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(lib.tbe_weighted_colsum_partials_f32(ptr(x), ptr(w), B, N, ptr(partial), partial.numel() * 4, stream_ptr(dev)),
              "tbe_weighted_colsum_partials_f32")
    return partial


def _multi_chunk_sum(seg_table, nseg, max_numel, dst, scale):
    """dst[off_s + i] = scale * sum_c src_s[c, i] for every segment of `seg_table` (device int64 [nseg, 4]: src address,
    chunks, numel, dst element offset) in one launch; the caller keeps the sources alive."""
    dev = require_gpu(seg_table, dst)
    if seg_table.dtype != torch.int64 or not seg_table.is_contiguous() or seg_table.numel() != 4 * nseg:
        raise RuntimeError("multi_chunk_sum: seg_table must be a contiguous int64 [nseg, 4] device tensor")
    if dst.dtype != torch.float32 or not dst.is_contiguous():
        raise RuntimeError("multi_chunk_sum: dst must be a contiguous float32 tensor")
    with torch.cuda.device(dev):
        check(_lib.load().tbe_multi_chunk_sum_f32(ptr(seg_table), nseg, max_numel, ptr(dst), scale, stream_ptr(dev)),
              "tbe_multi_chunk_sum_f32")
    return dst


_bce_ws = {}


def _bce_with_logits(logits, labels):
    """(mean BCE-with-logits loss [scalar], d loss / d logits [B]) in one launch; labels float32 or int64."""
    dev = require_gpu(logits, labels)
    if logits.dim() != 1 or labels.shape != logits.shape or logits.dtype != torch.float32:
        raise RuntimeError(f"bce_with_logits: need float32 logits [B] and labels [B], got {tuple(logits.shape)} / {tuple(labels.shape)}")
    if labels.dtype not in (torch.float32, torch.int64):
        labels = labels.float()
    logits, labels = logits.contiguous(), labels.contiguous()
    lib = _lib.load()
    key = (dev, stream_ptr(dev))  # one workspace per stream: two streams must not share block partials / the ticket
    ws = _bce_ws.get(key)
    if ws is None:  # [block partials | ticket]: zeroed once, the kernel resets its ticket
        if torch.cuda.is_current_stream_capturing():
            # a persistent buffer must not come from a graph's memory pool (an earlier graph of the pool may use its
            # address for transient tensors and clobber it at every replay): run the op once on this stream before capturing
            raise RuntimeError("bce_with_logits: first use on this stream happens inside a graph capture; warm the op up "
                               "on the capture stream first (GraphedSegment does)")
        ws = _bce_ws[key] = torch.zeros(lib.tbe_bce_with_logits_workspace_bytes(), dtype=torch.uint8, device=dev)
    loss = torch.empty((), dtype=torch.float32, device=dev)
    dlogits = torch.empty_like(logits)
    with torch.cuda.device(dev):
        check(lib.tbe_bce_with_logits_f32(ptr(logits), ptr(labels), labels.element_size(), logits.numel(), ptr(loss),
                                          ptr(dlogits), ptr(ws), ws.numel(), stream_ptr(dev)), "tbe_bce_with_logits_f32")
    return loss, dlogits


def _simple_unpack(recv, dims, B_local, D_total, vec, scale):
    dev = require_gpu(recv, dims)
    recv = recv.contiguous()
    out = torch.empty((B_local, D_total), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(_lib.load().tbe_a2a_pooled_unpack(ptr(recv), ptr(out), ptr(dims), dims.numel(), B_local, D_total,
                                                int(vec), scale, stream_ptr(dev)), "tbe_a2a_pooled_unpack")
    return out


def _simple_pack(grad, dims, vec, scale):
    dev = require_gpu(grad, dims)
    grad = grad.contiguous()
    B_local, D_total = grad.shape
    send = torch.empty(B_local * D_total, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(_lib.load().tbe_a2a_pooled_pack(ptr(grad), ptr(send), ptr(dims), dims.numel(), B_local, D_total,
                                              int(vec), scale, stream_ptr(dev)), "tbe_a2a_pooled_pack")
    return send



def _unpack(recv, feat_out_col, feat_src, feat_slab_col, slab_offset, slab_stride, B_local, D_total, vec, scale):
    out = torch.empty((B_local, D_total), dtype=torch.float32, device=recv.device)
    return _unpack_into(recv, feat_out_col, feat_src, feat_slab_col, slab_offset, slab_stride, B_local, D_total, vec,
                        scale, out)


def _unpack_into(recv, feat_out_col, feat_src, feat_slab_col, slab_offset, slab_stride, B_local, D_total, vec, scale, out):
    """Same, into a caller-provided [B_local, D_total] buffer (e.g. the static input of a HIP-graph segment)."""
    dev = require_gpu(recv, feat_out_col, feat_src, feat_slab_col, slab_offset, slab_stride, out)
    recv = recv.contiguous()
    if out.dtype != torch.float32 or not out.is_contiguous() or out.numel() != B_local * D_total:
        raise RuntimeError("pooled_exchange_unpack_into: out must be a contiguous float32 [B_local, D_total] buffer")
    with torch.cuda.device(dev):
        check(_lib.load().tbe_pooled_exchange_unpack(
            ptr(recv), ptr(out), ptr(feat_out_col), ptr(feat_src), ptr(feat_slab_col), ptr(slab_offset),
            ptr(slab_stride), feat_src.numel(), slab_offset.numel(), B_local, D_total, int(vec), scale,
            stream_ptr(dev)), "tbe_pooled_exchange_unpack")
    return out


def _pack(grad, feat_out_col, feat_src, feat_slab_col, slab_offset, slab_stride, numel, vec, scale):
    dev = require_gpu(grad, feat_out_col, feat_src, feat_slab_col, slab_offset, slab_stride)
    grad = grad.contiguous()
    B_local, D_total = grad.shape
    send = torch.empty(numel, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(_lib.load().tbe_pooled_exchange_pack(
            ptr(grad), ptr(send), ptr(feat_out_col), ptr(feat_src), ptr(feat_slab_col), ptr(slab_offset),
            ptr(slab_stride), feat_src.numel(), slab_offset.numel(), B_local, D_total, int(vec), scale,
            stream_ptr(dev)), "tbe_pooled_exchange_pack")
    return send


def _pack_into(grad, feat_out_col, feat_src, feat_slab_col, slab_offset, slab_stride, vec, scale, send):
    """pooled_exchange_pack into the caller's (persistent) send buffer — what a captured backward graph needs."""
    dev = require_gpu(grad, feat_out_col, feat_src, feat_slab_col, slab_offset, slab_stride, send)
    if grad.dim() != 2 or not grad.is_contiguous() or grad.dtype != torch.float32:
        raise RuntimeError("pooled_exchange_pack_into: grad must be a contiguous float32 [B_local, D_total] matrix")
    if send.dtype != torch.float32 or not send.is_contiguous():
        raise RuntimeError("pooled_exchange_pack_into: send must be a contiguous float32 buffer")
    B_local, D_total = grad.shape
    with torch.cuda.device(dev):
        check(_lib.load().tbe_pooled_exchange_pack(
            ptr(grad), ptr(send), ptr(feat_out_col), ptr(feat_src), ptr(feat_slab_col), ptr(slab_offset),
            ptr(slab_stride), feat_src.numel(), slab_offset.numel(), B_local, D_total, int(vec), scale,
            stream_ptr(dev)), "tbe_pooled_exchange_pack")
    return send


_impl.impl("relu_backward_bias_grad", _relu_backward_bias_grad)
_impl.impl("weighted_colsum", _weighted_colsum)
_impl.impl("copy_rows", _copy_rows)
_impl.impl("relu_backward_bias_partials", _relu_backward_bias_partials)
_impl.impl("weighted_colsum_partials", _weighted_colsum_partials)
_impl.impl("multi_chunk_sum", _multi_chunk_sum)
_impl.impl("bce_with_logits", _bce_with_logits)
_impl.impl("pooled_exchange_unpack", _unpack)
_impl.impl("pooled_exchange_unpack_into", _unpack_into)
_impl.impl("pooled_exchange_pack", _pack)
_impl.impl("pooled_exchange_pack_into", _pack_into)
_impl.impl("a2a_pooled_unpack", _simple_unpack)
_impl.impl("a2a_pooled_pack", _simple_pack)
