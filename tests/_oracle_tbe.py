"""TEST-ONLY stand-in for SplitTableBatchedEmbeddingBagsCodegen that computes with the CPU oracle.
Lets the world_size-2 gloo tests exercise the distributed host logic (input dist, pooled exchange,
sharding bookkeeping) without a GPU.  Never imported by the product."""
from types import SimpleNamespace

import numpy as np
import torch
from torch import nn

import _paths  # noqa: F401
from oracle import oracle


class _Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, placeholder, mod, indices, offsets, psw):
        ctx.mod = mod
        ctx.save_for_backward(indices, offsets, psw)
        out, _ = oracle.tbe_forward(mod.tables, indices.numpy(), offsets.numpy(),
                                    psw.numpy() if psw is not None else None, mod.pooling)
        if mod.pooling == oracle.POOL_NONE:
            return torch.from_numpy(out)
        return mod._to_layout(mod._scale_blocks(torch.from_numpy(out), offsets))

    @staticmethod
    def backward(ctx, grad):
        indices, offsets, psw = ctx.saved_tensors
        mod = ctx.mod
        g = grad.contiguous() if mod.pooling == oracle.POOL_NONE else mod._scale_blocks(mod._from_layout(grad.contiguous()), offsets)
        g = g.contiguous()
        oracle.tbe_backward(mod.tables, indices.numpy(), offsets.numpy(), g.numpy(), oracle.OPT_EXACT_SGD,
                            mod.optimizer_args.learning_rate, psw.numpy() if psw is not None else None, mod.pooling)
        return None, None, None, None, None


class _MixedPooling:
    def set_feature_pooling(self, modes):
        """Mixed SUM / MEAN features: computed as SUM, MEAN features' blocks scaled by 1 / bag length (forward) and
        their gradient blocks scaled the same way (backward) — the same arithmetic the kernels do per feature."""
        self._feat_mean = None if modes is None else [int(m) == oracle.POOL_MEAN for m in modes]
        if modes is not None:
            self.pooling = oracle.POOL_SUM

    def _mean_scale(self, offsets):
        """[B, F] factors (1 for SUM features, 1 / len for MEAN ones; 0 for empty bags), or None."""
        fm = getattr(self, "_feat_mean", None)
        if not fm:
            return None
        B = (offsets.numel() - 1) // self.F
        lens = (offsets[1:] - offsets[:-1]).view(self.F, B).t().float()
        inv = torch.where(lens > 0, 1.0 / lens.clamp(min=1), torch.zeros_like(lens))
        return torch.where(torch.tensor(fm).view(1, -1), inv, torch.ones_like(inv))

    def _scale_blocks(self, x, offsets):
        sc = self._mean_scale(offsets)
        if sc is None:
            return x
        cols = torch.repeat_interleave(sc, torch.tensor(self.tables.feat_D.tolist()), dim=1)
        return x * cols


class OracleTBE(_MixedPooling, nn.Module):
    def __init__(self, specs, ftm, pooling_mode, device, fused_params):
        super().__init__()
        rows, dims = [s[0] for s in specs], [s[1] for s in specs]
        self.tables = oracle.Tables(rows, dims, ftm)
        self.pooling = int(pooling_mode)
        self.F = len(ftm)
        self.optimizer_args = SimpleNamespace(learning_rate=fused_params.get("learning_rate", 0.01))
        self._optimizer = fused_params.get("optimizer")
        self._W = 0
        # not a registered parameter (as in the product): fused tables expose no parameters
        object.__setattr__(self, "placeholder", torch.zeros(0, requires_grad=True))

    def set_a2a_output_layout(self, W):
        self._W = W

    def set_row_windows(self, first_rows, global_rows=None):
        """Stand-in for the product's row windows: global ids -> shard-local ids (rows of other shards become -1,
        which the oracle treats as a zero row)."""
        self._win_first = None if first_rows is None else np.asarray(first_rows, dtype=np.int64)

    def _localize(self, indices, offsets):
        first = getattr(self, "_win_first", None)
        if first is None:
            return indices
        B = (offsets.numel() - 1) // self.F
        per_feat = (offsets[B::B] - offsets[:-1:B]).numpy()
        shift = np.repeat(first, per_feat)
        rows = np.repeat(np.asarray([self.tables.rows[t] for t in self.tables.ftm], dtype=np.int64), per_feat)
        loc = indices.numpy() - shift
        return torch.from_numpy(np.where((loc >= 0) & (loc < rows), loc, -1))

    def _to_layout(self, out):  # [B, W*Dl] -> [W*B, Dl]
        if not self._W:
            return out
        B = out.shape[0]
        return out.view(B, self._W, -1).permute(1, 0, 2).reshape(self._W * B, -1).contiguous()

    def _from_layout(self, g):
        if not self._W:
            return g
        B = g.shape[0] // self._W
        return g.view(self._W, B, -1).permute(1, 0, 2).reshape(B, -1).contiguous()

    def split_embedding_weights(self):
        return [torch.from_numpy(w) for w in self.tables.weights]

    def split_optimizer_states(self):
        if "ROWWISE_ADAGRAD" in repr(getattr(self, "_optimizer", "")):  # key / shape surface only (tests of names)
            if not hasattr(self, "_m1"):
                self._m1 = [torch.zeros(w.shape[0]) for w in self.tables.weights]
            return [(m,) for m in self._m1]
        return [() for _ in self.tables.weights]

    def set_learning_rate(self, lr):
        self.optimizer_args.learning_rate = lr

    def forward(self, indices, offsets, psw=None):
        return _Fn.apply(self.placeholder, self, self._localize(indices.long(), offsets.long()), offsets.long(), psw)


def oracle_tbe_factory(specs, ftm, pooling_mode, device, fused_params):
    return OracleTBE(specs, ftm, pooling_mode, device, fused_params)


def oracle_seq_tbe_factory(specs, ftm, device, fused_params):
    return OracleTBE(specs, ftm, oracle.POOL_NONE, device, fused_params)


def _scatter_cols(out, block, offs, dims, stride):
    """out[b*stride + offs[f] + d] = block[b, Doff_f + d] on a [B, stride] view."""
    o = out.view(-1, stride)
    c = 0
    for off, d in zip(offs, dims):
        o[:, off:off + d] = block[:, c:c + d]
        c += d


def _gather_cols(grad, offs, dims, stride):
    g = grad.contiguous().view(-1, stride)
    return torch.cat([g[:, off:off + d] for off, d in zip(offs, dims)], dim=1).contiguous()


class _FnInto(torch.autograd.Function):
    """fused stand-in writing into a caller buffer (SplitTable...forward_into)."""

    @staticmethod
    def forward(ctx, out, placeholder, mod, indices, offsets, psw, offs, stride):
        ctx.mod, ctx.offs, ctx.stride = mod, offs, stride
        ctx.save_for_backward(indices, offsets, psw)
        block, _ = oracle.tbe_forward(mod.tables, indices.numpy(), offsets.numpy(),
                                      psw.numpy() if psw is not None else None, mod.pooling)
        _scatter_cols(out, mod._scale_blocks(torch.from_numpy(block), offsets), offs, mod.tables.feat_D.tolist(), stride)
        ctx.mark_dirty(out)
        return out

    @staticmethod
    def backward(ctx, grad):
        indices, offsets, psw = ctx.saved_tensors
        mod = ctx.mod
        g = mod._scale_blocks(_gather_cols(grad, ctx.offs, mod.tables.feat_D.tolist(), ctx.stride), offsets).contiguous()
        oracle.tbe_backward(mod.tables, indices.numpy(), offsets.numpy(), g.numpy(), oracle.OPT_EXACT_SGD,
                            mod.optimizer_args.learning_rate, psw.numpy() if psw is not None else None, mod.pooling)
        return (grad,) + (None,) * 7


def _fused_forward_into(self, out, out_offsets, row_stride, indices, offsets, psw=None):
    return _FnInto.apply(out, self.placeholder, self, self._localize(indices.long(), offsets.long()), offsets.long(), psw,
                         out_offsets.tolist(),
                         int(row_stride))


OracleTBE.forward_into = _fused_forward_into


class _DenseFnInto(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out, weights, mod, indices, offsets, psw, offs, stride):
        ctx.mod, ctx.offs, ctx.stride = mod, offs, stride
        ctx.save_for_backward(indices, offsets, psw)
        mod._sync_tables()
        block, _ = oracle.tbe_forward(mod.tables, indices.numpy(), offsets.numpy(),
                                      psw.numpy() if psw is not None else None, mod.pooling)
        _scatter_cols(out, mod._scale_blocks(torch.from_numpy(block), offsets), offs, mod.tables.feat_D.tolist(), stride)
        ctx.mark_dirty(out)
        return out

    @staticmethod
    def backward(ctx, grad):
        indices, offsets, psw = ctx.saved_tensors
        mod = ctx.mod
        g = mod._scale_blocks(_gather_cols(grad, ctx.offs, mod.tables.feat_D.tolist(), ctx.stride), offsets).contiguous()
        gw = [np.zeros((r, d), dtype=np.float32) for r, d in zip(mod.tables.rows, mod.tables.dims)]
        oracle.tbe_backward(mod.tables, indices.numpy(), offsets.numpy(), g.numpy(), oracle.OPT_DENSE_GRAD, 0.0,
                            psw.numpy() if psw is not None else None, mod.pooling, state0=gw)
        flat = torch.from_numpy(np.concatenate([x.reshape(-1) for x in gw])) if gw else torch.zeros(0)
        return (grad, flat) + (None,) * 6


class OracleDenseTBE(_MixedPooling, nn.Module):
    """TEST-ONLY stand-in for DenseTableBatchedEmbeddingBagsCodegen (replicated / data-parallel tables):
    `.weights` is a real nn.Parameter so DDP all-reduces its gradient and a dense optimizer steps it."""

    def __init__(self, specs, ftm, pooling_mode, device):
        super().__init__()
        rows, dims = [s[0] for s in specs], [s[1] for s in specs]
        self.tables = oracle.Tables(rows, dims, ftm)
        self.pooling = int(pooling_mode)
        self.F = len(self.tables.ftm)
        self.weights = nn.Parameter(torch.zeros(sum(r * d for r, d in zip(rows, dims))))

    def split_embedding_weights(self):
        out, o = [], 0
        for r, d in zip(self.tables.rows, self.tables.dims):
            out.append(self.weights.detach()[o:o + r * d].view(r, d))
            o += r * d
        return out

    def _sync_tables(self):
        for t, w in enumerate(self.split_embedding_weights()):
            self.tables.weights[t][...] = w.numpy()

    def forward_into(self, out, out_offsets, row_stride, indices, offsets, psw=None):
        return _DenseFnInto.apply(out, self.weights, self, indices.long(), offsets.long(), psw, out_offsets.tolist(),
                                  int(row_stride))


def oracle_dp_tbe_factory(specs, ftm, pooling_mode, device):
    return OracleDenseTBE(specs, ftm, pooling_mode, device)
