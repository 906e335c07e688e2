"""``fbgemm_gpu.split_table_batched_embeddings_ops`` for MI355X (gfx950).

Mirrors the module surface the reference constructs and calls
(torchrec/distributed/batched_embedding_kernel.py:629-640 ctor, :546-554 forward,
:124-148 / :250-257 optimizer surface, :468-477 / :677-704 dense variant) on top of the
C ABI in ``include/tbe_hip.h``.  Names, argument meaning and error behaviour follow the
public fbgemm_gpu API of the reference's era; what the reference's own tests do not pin
(row-wise Adagrad / Adam arithmetic, defaults) is marked "parity unpinned" in DESIGN.md.

There is no CPU implementation here: ``ComputeDevice.CPU`` / ``use_cpu=True`` raise.
"""
import enum
import os
import types
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
from torch import nn

from . import _lib
from ._lib import OptimizerArgs, check, ptr, require_gpu, stream_ptr, workspace
from .split_embedding_configs import EmbOptimType as OptimType
from .split_embedding_configs import SparseType


class EmbeddingLocation(enum.IntEnum):
    DEVICE = 0
    MANAGED = 1
    MANAGED_CACHING = 2
    HOST = 3


class ComputeDevice(enum.IntEnum):
    CPU = 0
    CUDA = 1


class PoolingMode(enum.IntEnum):
    SUM = 0
    MEAN = 1
    NONE = 2


class BoundsCheckMode(enum.IntEnum):
    FATAL = 0
    WARNING = 1
    IGNORE = 2
    NONE = 3


class CacheAlgorithm(enum.Enum):
    LRU = 0
    LFU = 1


class WeightDecayMode(enum.IntEnum):
    NONE = 0
    L2 = 1
    DECOUPLE = 2


_OPT_CODE = {
    OptimType.EXACT_SGD: 0,
    OptimType.SGD: 0,  # the exact (coalescing) form is a valid SGD
    OptimType.EXACT_ROWWISE_ADAGRAD: 1,
    OptimType.ROWWISE_ADAGRAD: 1,
    OptimType.ADAM: 2,
    OptimType.EXACT_ADAGRAD: 3,
}
_OPT_DENSE_GRAD = 100


@dataclass
class OptimizerArgsView:
    """``emb_module.optimizer_args`` (read at batched_embedding_kernel.py:252)."""

    stochastic_rounding: bool
    gradient_clipping: bool
    max_gradient: float
    learning_rate: float
    eps: float
    beta1: float
    beta2: float
    weight_decay: float
    weight_decay_mode: int
    eta: float
    momentum: float


def rounded_row_size_in_bytes(dim: int, weight_ty: SparseType) -> int:
    raise NotImplementedError("quantized inference TBE is out of scope of the MI355X hot path")


class IntNBitTableBatchedEmbeddingBagsCodegen(nn.Module):
    def __init__(self, *args, **kwargs) -> None:
        super().__init__()
        raise NotImplementedError("quantized inference TBE is out of scope of the MI355X hot path")


def _bit_length(n: int) -> int:
    return max(1, int(n).bit_length())


class _Layout:
    """Device-side feature metadata for the kernels (rebuilt when storage moves)."""

    def __init__(self) -> None:
        self.key = None
        self.feat_weights = None
        self.feat_D = None
        self.feat_out_offset = None
        self.feat_rows = None
        self.feat_row_base = None
        self.feat_state0 = None
        self.feat_state1 = None


class _TBEBase(nn.Module):
    """Storage + launch logic shared by the fused (split) and dense variants."""

    def _init_tables(
        self,
        rows: List[int],
        dims: List[int],
        locations: List[EmbeddingLocation],
        feature_table_map: Optional[List[int]],
        pooling_mode: PoolingMode,
        device: Optional[torch.device],
    ) -> None:
        T = len(rows)
        if T == 0:
            raise ValueError("embedding_specs is empty")
        self.pooling_mode = PoolingMode(pooling_mode)
        self.feature_table_map: List[int] = (
            list(feature_table_map) if feature_table_map is not None else list(range(T))
        )
        if any(t < 0 or t >= T for t in self.feature_table_map):
            raise ValueError("feature_table_map entry out of range")
        self.rows_per_table = [int(r) for r in rows]
        self.dims_per_table = [int(d) for d in dims]
        if any(d <= 0 or d > 2048 for d in self.dims_per_table):
            raise ValueError("embedding dim must be in (0, 2048]")
        if any(r < 0 for r in self.rows_per_table):
            raise ValueError("negative row count")
        self.locations = [EmbeddingLocation(loc) for loc in locations]
        self.T = T
        self.F = len(self.feature_table_map)
        self.feat_D = [self.dims_per_table[t] for t in self.feature_table_map]
        self.D_offsets = [0]
        for d in self.feat_D:
            self.D_offsets.append(self.D_offsets[-1] + d)
        self.total_D = self.D_offsets[-1]
        self.max_D = max(self.dims_per_table)
        if self.pooling_mode == PoolingMode.NONE and len(set(self.dims_per_table)) != 1:
            raise ValueError("PoolingMode.NONE requires every table to have the same dim")
        self.row_base = [0]
        for r in self.rows_per_table:
            self.row_base.append(self.row_base[-1] + r)
        self.total_rows = self.row_base[-1]
        self.key_bits = _bit_length(self.total_rows)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.current_device = torch.device(device)
        # element offsets of each table inside its placement's flat buffer
        self.weights_offsets: List[int] = []
        self.state_row_offsets: List[int] = []
        sizes = {"dev": 0, "uvm": 0}
        row_sizes = {"dev": 0, "uvm": 0}
        self.placement: List[str] = []
        for r, d, loc in zip(self.rows_per_table, self.dims_per_table, self.locations):
            if loc == EmbeddingLocation.HOST:
                raise RuntimeError(
                    "EmbeddingLocation.HOST (CPU compute) is not provided by the MI355X build: "
                    "there is no CPU fallback"
                )
            p = "dev" if loc == EmbeddingLocation.DEVICE else "uvm"
            self.placement.append(p)
            # keep every table 16-B aligned inside the flat buffer
            sizes[p] = (sizes[p] + 3) // 4 * 4
            self.weights_offsets.append(sizes[p])
            self.state_row_offsets.append(row_sizes[p])
            sizes[p] += r * d
            row_sizes[p] += r
        self._flat_sizes = sizes
        self._row_sizes = row_sizes
        self._layout = _Layout()
        self._bounds_errors: Optional[torch.Tensor] = None
        self._side_stream = None
        # sort the batch's row keys on a side stream during forward (see _prepare_backward)
        # "auto": only for lookups small enough to leave CUs idle (measured on MI355X: at 1.7 M ids the
        # side-stream sort steals bandwidth from the GEMMs, -2 %; at 213 K ids it hides, +4 %)
        self.overlap_backward_sort = os.environ.get("TBE_OVERLAP_SORT", "auto")
        self.overlap_backward_sort_max_ids = 1 << 20

    # -- storage helpers ------------------------------------------------------------------
    def _alloc(self, placement: str, numel: int) -> torch.Tensor:
        if self.current_device.type == "meta":
            return torch.empty(numel, dtype=torch.float32, device="meta")
        if placement == "dev":
            return torch.zeros(numel, dtype=torch.float32, device=self.current_device)
        # MANAGED / MANAGED_CACHING: pinned host memory, GPU-mapped at the same address
        # (HIP unified addressing); the kernels read it over the host link.
        return torch.zeros(numel, dtype=torch.float32).pin_memory()

    def _flat_weights(self, placement: str) -> torch.Tensor:
        raise NotImplementedError

    def _table_view(self, flat_of, t: int) -> torch.Tensor:
        off = self.weights_offsets[t]
        r, d = self.rows_per_table[t], self.dims_per_table[t]
        return flat_of(self.placement[t]).detach()[off:off + r * d].view(r, d)

    def split_embedding_weights(self) -> List[torch.Tensor]:
        """Per-table ``[rows, dim]`` views aliasing the module's storage
        (written in place by batched_embedding_kernel.py:541-544 and embedding_lookup.py:70)."""
        return [self._table_view(self._flat_weights, t) for t in range(self.T)]

    # -- device metadata --------------------------------------------------------------------
    def _state_ptrs(self) -> Tuple[Optional[List[int]], Optional[List[int]]]:
        return None, None

    def _storage_key(self) -> Tuple:
        return (self._flat_weights("dev").data_ptr(), self._flat_weights("uvm").data_ptr())

    def _get_layout(self) -> _Layout:
        s0, s1 = self._state_ptrs()
        key = (self._storage_key(), tuple(s0 or ()), tuple(s1 or ()))
        lay = self._layout
        if lay.key == key:
            return lay
        dev = self.current_device
        base = {p: self._flat_weights(p).data_ptr() for p in ("dev", "uvm")}
        wptr = [base[self.placement[t]] + 4 * self.weights_offsets[t] for t in range(self.T)]
        ftm = self.feature_table_map

        def i64(vals):
            return torch.tensor(vals, dtype=torch.int64).to(dev)

        def i32(vals):
            return torch.tensor(vals, dtype=torch.int32).to(dev)

        lay.feat_weights = i64([wptr[t] for t in ftm])
        lay.feat_D = i32(self.feat_D)
        lay.feat_out_offset = i64(self.D_offsets[:-1])
        lay.feat_rows = i64([self.rows_per_table[t] for t in ftm])
        lay.feat_row_base = i64([self.row_base[t] for t in ftm])
        lay.feat_state0 = i64([s0[t] for t in ftm]) if s0 is not None else None
        lay.feat_state1 = i64([s1[t] for t in ftm]) if s1 is not None else None
        lay.key = key
        return lay

    def _errors(self) -> torch.Tensor:
        if self._bounds_errors is None or self._bounds_errors.device != self.current_device:
            self._bounds_errors = torch.zeros(1, dtype=torch.int32, device=self.current_device)
        return self._bounds_errors

    def bounds_check_errors(self) -> int:
        """Number of out-of-range indices seen so far (they contribute zero rows). Syncs."""
        return int(self._errors().item())

    # -- launches ---------------------------------------------------------------------------
    def _check_inputs(self, indices, offsets, per_sample_weights) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor], int]:
        require_gpu(indices, offsets, per_sample_weights)
        if indices.device != self.current_device or offsets.device != self.current_device:
            raise RuntimeError(f"indices/offsets must be on {self.current_device}")
        if indices.dtype != torch.int64 or offsets.dtype != torch.int64:
            # the reference always passes .long() (batched_embedding_kernel.py:551-552)
            indices = indices.long()
            offsets = offsets.long()
        indices = indices.contiguous().view(-1)
        offsets = offsets.contiguous().view(-1)
        if (offsets.numel() - 1) % self.F != 0:
            raise RuntimeError(
                f"offsets has {offsets.numel()} entries; expected F*B+1 with F={self.F}")
        B = (offsets.numel() - 1) // self.F
        if per_sample_weights is not None:
            if per_sample_weights.dtype != torch.float32:
                per_sample_weights = per_sample_weights.float()
            per_sample_weights = per_sample_weights.contiguous().view(-1)
            if per_sample_weights.numel() != indices.numel():
                raise RuntimeError("per_sample_weights must have one entry per index")
        return indices, offsets, per_sample_weights, B

    # -- output layout --------------------------------------------------------------------
    def set_a2a_output_layout(self, world_size: int) -> None:
        """Pooled output (and the gradient read by backward) laid out all-to-all-ready:
        the module's features are (src rank w, local feature f) pairs, w-major, each rank
        contributing ``B`` samples; output is ``[world_size, B, D_local]`` flattened, so slab w
        is exactly what rank w receives.  Replaces the reference's recat permute +
        split/cat copies (dist_data.py:257-263, comm_ops.py:555-561, :418-428)."""
        if world_size < 1 or self.F % world_size != 0:
            raise ValueError("feature count must be a multiple of world_size")
        self._a2a_world = world_size
        self._a2a_cache = {}

    def _pooled_layout(self, B: int):
        """(feat_out_offset tensor, row_stride, out_shape) for batch size B."""
        lay = self._get_layout()
        W = getattr(self, "_a2a_world", 0)
        if not W:
            return lay.feat_out_offset, self.total_D, (B, self.total_D)
        hit = self._a2a_cache.get(B)
        if hit is None:
            Fl = self.F // W
            Dl = self.D_offsets[Fl]
            offs = [w * B * Dl + self.D_offsets[f] for w in range(W) for f in range(Fl)]
            hit = (torch.tensor(offs, dtype=torch.int64).to(self.current_device), Dl, (W * B, Dl))
            self._a2a_cache[B] = hit
        return hit

    def _forward_impl(self, indices, offsets, per_sample_weights, B: int, into=None) -> torch.Tensor:
        lay = self._get_layout()
        dev = self.current_device
        lib = _lib.load()
        N = indices.numel()
        with torch.cuda.device(dev):
            if self.pooling_mode == PoolingMode.NONE:
                D = self.dims_per_table[0]
                out = torch.empty((N, D), dtype=torch.float32, device=dev)
                check(
                    lib.tbe_forward_nobag_f32(ptr(lay.feat_weights), ptr(lay.feat_rows), self.F, B,
                                              D, ptr(indices), N, ptr(offsets), ptr(out),
                                              ptr(self._errors()), stream_ptr(dev)),
                    "tbe_forward_nobag_f32",
                )
                return out
            if into is not None:
                # caller-provided buffer + addressing (feat_out_offset, row stride): several lookups
                # can fill disjoint column blocks of one [B, sum D] matrix without a cat
                out, out_off, stride = into
            else:
                out_off, stride, shape = self._pooled_layout(B)
                out = torch.empty(shape, dtype=torch.float32, device=dev)
            check(
                lib.tbe_forward_pooled_f32(ptr(lay.feat_weights), ptr(lay.feat_D),
                                           ptr(out_off), ptr(lay.feat_rows), self.F, B,
                                           self.max_D, ptr(indices), N, ptr(offsets),
                                           ptr(per_sample_weights), int(self.pooling_mode), ptr(out),
                                           stride, ptr(self._errors()), stream_ptr(dev)),
                "tbe_forward_pooled_f32",
            )
        return out

    def _prepare_backward(self, indices, offsets, B: int):
        """Enqueues the gradient-independent half of backward (linearize + stable sort of the row
        keys) on a side stream, right after the forward kernel, so that it overlaps whatever the
        caller does between forward and backward (the dense MLPs).  Returns (workspace, event)."""
        N = indices.numel()
        if N == 0 or B == 0:
            return None
        lay = self._get_layout()
        dev = self.current_device
        lib = _lib.load()
        with torch.cuda.device(dev):
            nbytes = lib.tbe_backward_workspace_bytes(N, self.F, B, self.max_D, self.key_bits)
            if nbytes == 0:
                check(-2, "tbe_backward_workspace_bytes")
            ws = workspace(nbytes, dev)
            side = self._side_stream
            if side is None or side.device != dev:
                side = self._side_stream = torch.cuda.Stream(dev)
            cur = torch.cuda.current_stream(dev)
            side.wait_stream(cur)
            for t in (ws, indices, offsets):
                t.record_stream(side)
            check(
                lib.tbe_backward_prepare(ptr(lay.feat_rows), ptr(lay.feat_row_base), self.F, B, self.max_D,
                                         self.key_bits, ptr(indices), N, ptr(offsets), int(self.pooling_mode),
                                         ptr(ws), ws.numel(), ptr(self._errors()), side.cuda_stream),
                "tbe_backward_prepare",
            )
            ev = torch.cuda.Event()
            ev.record(side)
        return ws, ev

    def _backward_impl(self, grad_out, indices, offsets, per_sample_weights, B: int,
                       opt: OptimizerArgs, state0_override: Optional[torch.Tensor] = None,
                       prepared=None, layout=None) -> None:
        lay = self._get_layout()
        dev = self.current_device
        lib = _lib.load()
        N = indices.numel()
        if N == 0 or B == 0:
            return
        grad_out = grad_out.contiguous()
        if grad_out.dtype != torch.float32:
            grad_out = grad_out.float()
        if layout is not None:
            out_off, stride = layout
        elif self.pooling_mode == PoolingMode.NONE:
            out_off, stride = lay.feat_out_offset, grad_out.shape[1]
        else:
            out_off, stride, _ = self._pooled_layout(B)
        feat_state0 = state0_override if state0_override is not None else lay.feat_state0
        # TBE_FLAG_UNIFORM_ALIGNED: one dim for every feature, multiple of 4 (table bases are laid
        # out 16-B aligned by _init_tables; out offsets are then multiples of 4 as well)
        flags = 1 if (len(set(self.dims_per_table)) == 1 and self.max_D % 4 == 0 and stride % 4 == 0
                      and state0_override is None) else 0
        with torch.cuda.device(dev):
            if prepared is not None:
                ws, ev = prepared
                torch.cuda.current_stream(dev).wait_event(ev)
                check(
                    lib.tbe_backward_apply_f32(ptr(lay.feat_weights), ptr(lay.feat_D), ptr(out_off),
                                               ptr(lay.feat_rows), ptr(lay.feat_row_base), ptr(feat_state0),
                                               ptr(lay.feat_state1), self.F, B, self.max_D, self.key_bits,
                                               ptr(indices), N, ptr(offsets), ptr(per_sample_weights),
                                               int(self.pooling_mode), ptr(grad_out), stride, opt, flags,
                                               ptr(ws), ws.numel(), stream_ptr(dev)),
                    "tbe_backward_apply_f32",
                )
                return
            nbytes = lib.tbe_backward_workspace_bytes(N, self.F, B, self.max_D, self.key_bits)
            if nbytes == 0:
                check(-2, "tbe_backward_workspace_bytes")
            ws = workspace(nbytes, dev)
            check(
                lib.tbe_backward_fused_f32(ptr(lay.feat_weights), ptr(lay.feat_D),
                                           ptr(out_off), ptr(lay.feat_rows),
                                           ptr(lay.feat_row_base), ptr(feat_state0),
                                           ptr(lay.feat_state1), self.F, B,
                                           self.max_D, self.key_bits, ptr(indices), N, ptr(offsets),
                                           ptr(per_sample_weights), int(self.pooling_mode),
                                           ptr(grad_out), stride, opt, flags, ptr(ws), ws.numel(),
                                           ptr(self._errors()), stream_ptr(dev)),
                "tbe_backward_fused_f32",
            )


class _FusedLookupInto(torch.autograd.Function):
    """Like _FusedLookup, but writes its column blocks into a caller-provided [B, stride] buffer."""

    @staticmethod
    def forward(ctx, out, placeholder, module, indices, offsets, per_sample_weights, B, out_off, stride, prepare):
        ctx.module, ctx.B, ctx.layout = module, B, (out_off, stride)
        ctx.save_for_backward(indices, offsets, per_sample_weights)
        module._forward_impl(indices, offsets, per_sample_weights, B, into=(out, out_off, stride))
        ctx.prepared = module._prepare_backward(indices, offsets, B) if prepare else None
        ctx.mark_dirty(out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        indices, offsets, psw = ctx.saved_tensors
        module = ctx.module
        module.iter += 1
        module._backward_impl(grad_out, indices, offsets, psw, ctx.B, module._optimizer_struct(),
                              prepared=ctx.prepared, layout=ctx.layout)
        ctx.prepared = None
        return (grad_out,) + (None,) * 9


class _DenseLookupInto(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out, weights, module, indices, offsets, per_sample_weights, B, out_off, stride):
        ctx.module, ctx.B, ctx.layout = module, B, (out_off, stride)
        ctx.save_for_backward(indices, offsets, per_sample_weights)
        module._forward_impl(indices, offsets, per_sample_weights, B, into=(out, out_off, stride))
        ctx.mark_dirty(out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        indices, offsets, psw = ctx.saved_tensors
        module = ctx.module
        grad_w = torch.zeros_like(module.weights)
        base = grad_w.data_ptr()
        ptrs = [base + 4 * module.weights_offsets[t] for t in module.feature_table_map]
        state0 = torch.tensor(ptrs, dtype=torch.int64).to(grad_w.device)
        opt = OptimizerArgs(_OPT_DENSE_GRAD, 0.0, 0.0, 0.0, 0.0, 0.0, 1)
        module._backward_impl(grad_out, indices, offsets, psw, ctx.B, opt, state0_override=state0, layout=ctx.layout)
        return (grad_out, grad_w) + (None,) * 7


class _FusedLookup(torch.autograd.Function):
    """forward = TBE gather/pool; backward = coalesce + fused optimizer (no weight grad)."""

    @staticmethod
    def forward(ctx, placeholder, module, indices, offsets, per_sample_weights, B, prepare):
        ctx.module = module
        ctx.B = B
        ctx.save_for_backward(indices, offsets, per_sample_weights)
        out = module._forward_impl(indices, offsets, per_sample_weights, B)
        ctx.prepared = module._prepare_backward(indices, offsets, B) if prepare else None
        return out

    @staticmethod
    def backward(ctx, grad_out):
        indices, offsets, psw = ctx.saved_tensors
        module = ctx.module
        module.iter += 1
        module._backward_impl(grad_out, indices, offsets, psw, ctx.B, module._optimizer_struct(),
                              prepared=ctx.prepared)
        ctx.prepared = None
        return None, None, None, None, None, None, None


class SplitTableBatchedEmbeddingBagsCodegen(_TBEBase):
    """Table-batched embedding bags with the optimizer fused into backward.

    ``embedding_specs``: ``[(rows, dim, EmbeddingLocation, ComputeDevice)]`` as built at
    torchrec/distributed/batched_embedding_kernel.py:631-633.
    """

    def __init__(
        self,
        embedding_specs: List[Tuple[int, int, EmbeddingLocation, ComputeDevice]],
        feature_table_map: Optional[List[int]] = None,
        cache_algorithm: CacheAlgorithm = CacheAlgorithm.LRU,
        cache_load_factor: float = 0.2,
        cache_sets: int = 0,
        cache_reserved_memory: float = 0.0,
        cache_precision: SparseType = SparseType.FP32,
        weights_precision: SparseType = SparseType.FP32,
        output_dtype: SparseType = SparseType.FP32,
        enforce_hbm: bool = False,
        optimizer: OptimType = OptimType.EXACT_SGD,
        record_cache_metrics=None,
        stochastic_rounding: bool = True,
        gradient_clipping: bool = False,
        max_gradient: float = 1.0,
        learning_rate: float = 0.01,
        eps: float = 1.0e-8,
        momentum: float = 0.9,
        weight_decay: float = 0.0,
        weight_decay_mode: WeightDecayMode = WeightDecayMode.NONE,
        eta: float = 0.001,
        beta1: float = 0.9,
        beta2: float = 0.999,
        pooling_mode: PoolingMode = PoolingMode.SUM,
        device: Optional[torch.device] = None,
        bounds_check_mode: BoundsCheckMode = BoundsCheckMode.WARNING,
    ) -> None:
        super().__init__()
        rows, dims, locations, compute_devices = zip(*embedding_specs)
        if any(ComputeDevice(c) == ComputeDevice.CPU for c in compute_devices):
            raise RuntimeError(
                "ComputeDevice.CPU is not provided by the MI355X build (no CPU fallback); "
                "use the reference's `dense`/`sparse` compute kernels on CPU"
            )
        if weights_precision != SparseType.FP32 or output_dtype != SparseType.FP32:
            raise NotImplementedError("only FP32 tables / outputs are implemented")
        if optimizer not in _OPT_CODE:
            raise NotImplementedError(f"optimizer {optimizer} is not implemented")
        if gradient_clipping:
            raise NotImplementedError("gradient_clipping is not implemented")
        self._init_tables(list(rows), list(dims), list(locations), feature_table_map,
                          pooling_mode, device)
        self.optimizer = optimizer
        self.bounds_check_mode = bounds_check_mode
        self.cache_load_factor = cache_load_factor
        self.optimizer_args = OptimizerArgsView(
            stochastic_rounding=stochastic_rounding,
            gradient_clipping=gradient_clipping,
            max_gradient=max_gradient,
            learning_rate=learning_rate,
            eps=eps,
            beta1=beta1,
            beta2=beta2,
            weight_decay=weight_decay,
            weight_decay_mode=int(weight_decay_mode),
            eta=eta,
            momentum=momentum,
        )
        self.iter = 0
        self.register_buffer("weights_dev", self._alloc("dev", self._flat_sizes["dev"]), persistent=False)
        self.register_buffer("weights_uvm", self._alloc("uvm", self._flat_sizes["uvm"]), persistent=False)
        code = _OPT_CODE[optimizer]
        rowwise = code == 1
        elementwise = code in (2, 3)
        for name, needed in (("momentum1", rowwise or elementwise), ("momentum2", code == 2)):
            for p in ("dev", "uvm"):
                n = 0
                if needed:
                    n = self._row_sizes[p] if rowwise else self._flat_sizes[p]
                self.register_buffer(f"{name}_{p}", self._alloc(p, n), persistent=False)
        # A zero-size leaf that requires grad so autograd reaches backward.  Deliberately NOT an
        # nn.Parameter: it must stay invisible to DistributedDataParallel and to dense optimizers
        # (the reference hides its counterpart with `named_parameters -> ()`,
        # batched_embedding_kernel.py:655-658).
        object.__setattr__(self, "placeholder_autograd_tensor", torch.zeros(
            0, dtype=torch.float32, requires_grad=True,
            device=self.current_device if self.current_device.type != "meta" else "meta"))

    # storage ------------------------------------------------------------------------------
    def _flat_weights(self, placement: str) -> torch.Tensor:
        return self.weights_dev if placement == "dev" else self.weights_uvm

    def _state_flat(self, name: str, placement: str) -> torch.Tensor:
        return getattr(self, f"{name}_{placement}")

    def _state_ptrs(self):
        code = _OPT_CODE[self.optimizer]
        if code == 0:
            return None, None
        out = []
        for name in ("momentum1", "momentum2"):
            if name == "momentum2" and code != 2:
                out.append(None)
                continue
            ptrs = []
            for t in range(self.T):
                flat = self._state_flat(name, self.placement[t])
                off = self.state_row_offsets[t] if code == 1 else self.weights_offsets[t]
                ptrs.append(flat.data_ptr() + 4 * off)
            out.append(ptrs)
        return out[0], out[1]

    def split_optimizer_states(self) -> List[Tuple[torch.Tensor, ...]]:
        """Per-table optimizer state views (batched_embedding_kernel.py:133-148):
        ``()`` for SGD, ``(momentum1[rows],)`` for row-wise Adagrad,
        ``(m[rows, D], v[rows, D])`` for Adam, ``(momentum1[rows, D],)`` for Adagrad."""
        code = _OPT_CODE[self.optimizer]
        states: List[Tuple[torch.Tensor, ...]] = []
        for t in range(self.T):
            r, d, p = self.rows_per_table[t], self.dims_per_table[t], self.placement[t]
            if code == 0:
                states.append(())
            elif code == 1:
                o = self.state_row_offsets[t]
                states.append((self.momentum1_dev[o:o + r] if p == "dev" else self.momentum1_uvm[o:o + r],))
            else:
                o = self.weights_offsets[t]
                m1 = self._state_flat("momentum1", p)[o:o + r * d].view(r, d)
                if code == 2:
                    m2 = self._state_flat("momentum2", p)[o:o + r * d].view(r, d)
                    states.append((m1, m2))
                else:
                    states.append((m1,))
        return states

    # optimizer surface ---------------------------------------------------------------------
    def set_learning_rate(self, lr: float) -> None:
        self.optimizer_args.learning_rate = float(lr)

    def flush(self) -> None:
        """Write-back hook for MANAGED_CACHING (batched_embedding_kernel.py:563, 664).
        Tables are read and updated in place (no HBM row cache yet), so nothing is pending."""
        return None

    def _optimizer_struct(self) -> OptimizerArgs:
        a = self.optimizer_args
        return OptimizerArgs(_OPT_CODE[self.optimizer], a.learning_rate, a.eps, a.weight_decay,
                             a.beta1, a.beta2, max(self.iter, 1))

    def forward(self, indices: torch.Tensor, offsets: torch.Tensor,
                per_sample_weights: Optional[torch.Tensor] = None,
                feature_requires_grad: Optional[torch.Tensor] = None) -> torch.Tensor:
        indices, offsets, per_sample_weights, B = self._check_inputs(indices, offsets, per_sample_weights)
        mode = self.overlap_backward_sort
        prepare = torch.is_grad_enabled() and (
            mode in (True, "1") or (mode == "auto" and indices.numel() <= self.overlap_backward_sort_max_ids))
        return _FusedLookup.apply(self.placeholder_autograd_tensor, self, indices, offsets,
                                  per_sample_weights, B, prepare)

    def forward_into(self, out: torch.Tensor, out_offsets: torch.Tensor, row_stride: int, indices: torch.Tensor,
                     offsets: torch.Tensor, per_sample_weights: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Pooled lookup whose feature blocks land at `out[b * row_stride + out_offsets[f] + d]` of the
        given buffer (returned, marked dirty for autograd)."""
        indices, offsets, per_sample_weights, B = self._check_inputs(indices, offsets, per_sample_weights)
        mode = self.overlap_backward_sort
        prepare = torch.is_grad_enabled() and (
            mode in (True, "1") or (mode == "auto" and indices.numel() <= self.overlap_backward_sort_max_ids))
        return _FusedLookupInto.apply(out, self.placeholder_autograd_tensor, self, indices, offsets,
                                      per_sample_weights, B, out_offsets, int(row_stride), prepare)


class _DenseLookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weights, module, indices, offsets, per_sample_weights, B):
        ctx.module = module
        ctx.B = B
        ctx.save_for_backward(indices, offsets, per_sample_weights)
        return module._forward_impl(indices, offsets, per_sample_weights, B)

    @staticmethod
    def backward(ctx, grad_out):
        indices, offsets, psw = ctx.saved_tensors
        module = ctx.module
        grad_w = torch.zeros_like(module.weights)
        base = grad_w.data_ptr()
        ptrs = [base + 4 * module.weights_offsets[t] for t in module.feature_table_map]
        state0 = torch.tensor(ptrs, dtype=torch.int64).to(grad_w.device)
        opt = OptimizerArgs(_OPT_DENSE_GRAD, 0.0, 0.0, 0.0, 0.0, 0.0, 1)
        module._backward_impl(grad_out, indices, offsets, psw, ctx.B, opt, state0_override=state0)
        return grad_w, None, None, None, None, None


class DenseTableBatchedEmbeddingBagsCodegen(_TBEBase):
    """Table-batched embedding bags with a dense ``.weights`` parameter (optimizer external);
    built at torchrec/distributed/batched_embedding_kernel.py:677-686."""

    def __init__(
        self,
        embedding_specs: List[Tuple[int, int]],
        feature_table_map: Optional[List[int]] = None,
        pooling_mode: PoolingMode = PoolingMode.SUM,
        use_cpu: bool = False,
    ) -> None:
        super().__init__()
        if use_cpu:
            raise RuntimeError("use_cpu=True is not provided by the MI355X build (no CPU fallback)")
        rows, dims = zip(*embedding_specs)
        self._init_tables(list(rows), list(dims), [EmbeddingLocation.DEVICE] * len(rows),
                          feature_table_map, pooling_mode, None)
        self.weights = nn.Parameter(torch.zeros(self._flat_sizes["dev"], dtype=torch.float32,
                                                device=self.current_device))
        self._empty = torch.zeros(0, dtype=torch.float32)

    def _flat_weights(self, placement: str) -> torch.Tensor:
        return self.weights if placement == "dev" else self._empty

    def _storage_key(self):
        return (self.weights.data_ptr(), 0)

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self.current_device = self.weights.device
        return r

    def forward(self, indices: torch.Tensor, offsets: torch.Tensor,
                per_sample_weights: Optional[torch.Tensor] = None,
                feature_requires_grad: Optional[torch.Tensor] = None) -> torch.Tensor:
        indices, offsets, per_sample_weights, B = self._check_inputs(indices, offsets, per_sample_weights)
        return _DenseLookup.apply(self.weights, self, indices, offsets, per_sample_weights, B)

    def forward_into(self, out: torch.Tensor, out_offsets: torch.Tensor, row_stride: int, indices: torch.Tensor,
                     offsets: torch.Tensor, per_sample_weights: Optional[torch.Tensor] = None) -> torch.Tensor:
        indices, offsets, per_sample_weights, B = self._check_inputs(indices, offsets, per_sample_weights)
        return _DenseLookupInto.apply(out, self.weights, self, indices, offsets, per_sample_weights, B, out_offsets,
                                      int(row_stride))
