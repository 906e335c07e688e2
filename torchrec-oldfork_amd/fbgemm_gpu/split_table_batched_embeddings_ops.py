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
from ctypes import byref as ctypes_byref
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
from torch import nn

from . import _lib, _streams
from ._lib import CacheDesc, OptimizerArgs, check, ptr, raise_on_faults, require_gpu, stream_ptr, workspace
from .split_embedding_configs import EmbOptimType as OptimType
from .split_embedding_configs import SparseType


_FLAG_WEIGHTED = 2  # TBE_FLAG_WEIGHTED (include/tbe_hip.h)


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
        self.feat_window = None  # [2F] (first global row, global rows) of row-wise shards, or None
        self.feat_pooling = None  # [F] per-feature SUM / MEAN, or None = uniform
        self.feat_row_base = None
        self.feat_state0 = None
        self.feat_state1 = None


class _RowCache:
    """HBM row cache of a module's EmbeddingLocation.MANAGED_CACHING tables (csrc/tbe_cache.hip).

    The tables stay in pinned host memory; `rows` = [num_sets * 64 cache slots | staging slots] in HBM.
    `prefetch` rewrites a batch's ids of cached features into slot numbers, and the module's kernels
    see those features as lookups into the pseudo-table `rows` (see `redirect`)."""

    WAYS = 64

    def __init__(self, module: "_TBEBase", tables: List[int], load_factor: float, cache_sets: int,
                 rowwise_state: bool) -> None:
        self.m = module
        self.tables = tables
        dims = {module.dims_per_table[t] for t in tables}
        if len(dims) != 1:
            raise NotImplementedError("MANAGED_CACHING tables of one module must share one embedding dim")
        self.D = dims.pop()
        rows = [module.rows_per_table[t] for t in tables]
        self.key_base = [0]
        for r in rows:
            self.key_base.append(self.key_base[-1] + r)
        total = self.key_base[-1]
        self.key_bits = max(1, int(total).bit_length())
        if not 0.0 < load_factor <= 1.0:
            raise ValueError("cache_load_factor must be in (0, 1]")
        self.num_sets = int(cache_sets) if cache_sets else max(1, -(-int(total * load_factor) // self.WAYS))
        self.slots = self.num_sets * self.WAYS
        self.rowwise_state = rowwise_state
        self.staging_cap = 0
        self.iteration = 0
        self.pending = False   # a training forward whose backward has not run yet
        self.dirty = False     # cached rows may differ from the host tables
        self.rows: Optional[torch.Tensor] = None
        self.state: Optional[torch.Tensor] = None
        dev = module.current_device
        self.tags = torch.full((self.slots,), -1, dtype=torch.int64, device=dev)
        self.lru = torch.full((self.slots,), -1, dtype=torch.int32, device=dev)
        self.counters = torch.zeros(8, dtype=torch.int32, device=dev)
        self.staging_keys: Optional[torch.Tensor] = None
        self.tab_key_base = torch.tensor(self.key_base, dtype=torch.int64).to(dev)
        self.tab_D = torch.tensor([self.D] * len(tables), dtype=torch.int32).to(dev)
        ctab = {t: i for i, t in enumerate(tables)}
        self.feat_ctab_host = [ctab.get(t, -1) for t in module.feature_table_map]
        self.feat_ctab = torch.tensor(self.feat_ctab_host, dtype=torch.int32).to(dev)
        self.cached_feats = torch.tensor([f for f, c in enumerate(self.feat_ctab_host) if c >= 0], dtype=torch.int64).to(dev)
        self._desc_key = None
        self._desc = None
        self._tab_ptrs = None
        self._redirect_key = None
        self._redirected = None

    def __getstate__(self):
        d = dict(self.__dict__)
        for k in ("_desc", "_desc_key", "_tab_ptrs", "_redirect_key", "_redirected"):
            d[k] = None  # ctypes descriptors / device pointer tables are rebuilt on demand
        return d

    # -- storage ----------------------------------------------------------------------------
    def ensure(self, N: int) -> None:
        """(Re)allocates rows / state so that the staging area holds a batch of N ids."""
        if self.rows is not None and N <= self.staging_cap:
            return
        if self.pending:
            raise RuntimeError("MANAGED_CACHING: the staging area cannot grow between a training forward and its backward")
        dev = self.m.current_device
        cap = max(1024, -(-max(N, 2 * self.staging_cap) // 1024) * 1024)
        rows = torch.zeros((self.slots + cap) * self.D, dtype=torch.float32, device=dev)
        state = torch.zeros(self.slots + cap, dtype=torch.float32, device=dev) if self.rowwise_state else None
        if self.rows is not None:  # keep the cached rows (tags / lru stay valid)
            rows[:self.slots * self.D].copy_(self.rows[:self.slots * self.D])
            if state is not None:
                state[:self.slots].copy_(self.state[:self.slots])
        self.rows, self.state, self.staging_cap = rows, state, cap
        self.staging_keys = torch.zeros(cap, dtype=torch.int64, device=dev)
        self.m.key_bits = _bit_length(self.m.total_rows + self.slots + cap)

    def desc(self) -> CacheDesc:
        m = self.m
        m._ensure_pinned()
        wbase = m._flat_weights("uvm").data_ptr()
        sbase = m._state_flat("momentum1", "uvm").data_ptr() if self.rowwise_state else 0
        key = (wbase, sbase, self.rows.data_ptr(), self.staging_cap)
        if self._desc_key != key:
            dev = m.current_device
            w = torch.tensor([wbase + 4 * m.weights_offsets[t] for t in self.tables], dtype=torch.int64).to(dev)
            st = (torch.tensor([sbase + 4 * m.state_row_offsets[t] for t in self.tables], dtype=torch.int64).to(dev)
                  if self.rowwise_state else None)
            self._tab_ptrs = (w, st)
            self._desc = CacheDesc(ptr(self.tags), ptr(self.lru), ptr(self.rows), ptr(self.state), ptr(self.staging_keys),
                                   ptr(self.counters), ptr(self.tab_key_base), ptr(w), ptr(st), ptr(self.tab_D),
                                   self.num_sets, self.D, self.staging_cap, len(self.tables))
            self._desc_key = key
        return self._desc

    def redirect(self, lay: "_Layout") -> "_Layout":
        """The module's feature metadata with the cached features pointing at the pseudo-table."""
        key = (lay.key, self.rows.data_ptr(), self.staging_cap)
        if self._redirect_key != key:
            r = _Layout()
            r.key = key
            cf = self.cached_feats
            r.feat_weights = lay.feat_weights.clone()
            r.feat_weights[cf] = self.rows.data_ptr()
            r.feat_rows = lay.feat_rows.clone()
            r.feat_rows[cf] = self.slots + self.staging_cap
            r.feat_window = None
            if lay.feat_window is not None:
                # cached features arrive as slot numbers (or TBE_ID_SKIP / -1): their window is the pseudo-table's
                r.feat_window = lay.feat_window.clone().view(-1, 2)
                r.feat_window[cf, 0] = 0
                r.feat_window[cf, 1] = self.slots + self.staging_cap
                r.feat_window = r.feat_window.view(-1)
            r.feat_row_base = lay.feat_row_base.clone()
            r.feat_row_base[cf] = self.m.total_rows
            r.feat_D, r.feat_out_offset, r.feat_pooling = lay.feat_D, lay.feat_out_offset, lay.feat_pooling
            r.feat_state0, r.feat_state1 = lay.feat_state0, lay.feat_state1
            if self.rowwise_state:
                r.feat_state0 = lay.feat_state0.clone()
                r.feat_state0[cf] = self.state.data_ptr()
            self._redirected, self._redirect_key = r, key
        return self._redirected

    # -- per-step ---------------------------------------------------------------------------
    def prefetch(self, real: "_Layout", indices: torch.Tensor, offsets: torch.Tensor, B: int, training: bool) -> torch.Tensor:
        if self.pending:
            raise RuntimeError("MANAGED_CACHING supports one outstanding training forward per module: run the "
                               "backward of the previous forward first")
        N = indices.numel()
        self.ensure(N)
        dev = self.m.current_device
        lib = _lib.load()
        out = torch.empty_like(indices)
        self.iteration += 1
        if self.iteration >= (1 << 25):
            raise RuntimeError("MANAGED_CACHING: iteration counter exhausted; flush() and rebuild the module's cache")
        with torch.cuda.device(dev):
            ws = workspace(lib.tbe_cache_prefetch_workspace_bytes(N, self.key_bits), dev)
            check(lib.tbe_cache_prefetch(ctypes_byref(self.desc()), ptr(self.feat_ctab), ptr(real.feat_rows), self.m.F, B,
                                         ptr(indices), N, ptr(offsets), self.key_bits, self.iteration, ptr(out), ptr(ws),
                                         ws.numel(), ptr(real.feat_window), stream_ptr(dev)),
                  "tbe_cache_prefetch")
        self.pending = training
        self.dirty = self.dirty or training
        return out

    def after_backward(self) -> None:
        dev = self.m.current_device
        with torch.cuda.device(dev):
            check(_lib.load().tbe_cache_writeback_staging(ctypes_byref(self.desc()), stream_ptr(dev)),
                  "tbe_cache_writeback_staging")
        self.pending = False

    def flush(self, invalidate: bool) -> None:
        if self.rows is None or (not self.dirty and not invalidate):
            return
        if self.pending:
            raise RuntimeError("MANAGED_CACHING: flush() between a training forward and its backward")
        dev = self.m.current_device
        with torch.cuda.device(dev):
            check(_lib.load().tbe_cache_flush(ctypes_byref(self.desc()), 1 if invalidate else 0, stream_ptr(dev)),
                  "tbe_cache_flush")
        torch.cuda.current_stream(dev).synchronize()  # host views are read / written by the caller next
        self.dirty = False

    def stats(self) -> dict:
        c = self.counters.cpu().tolist()
        return {"hits": c[1], "misses": c[2], "evictions": c[3], "unique_last_batch": c[4], "misses_last_batch": c[5],
                "staged_last_batch": c[0], "num_sets": self.num_sets, "slots": self.slots}


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
        self._cache: Optional[_RowCache] = None
        self._bounds_errors: Optional[torch.Tensor] = None
        self._row_windows = None  # see set_row_windows
        self._feature_pooling = None  # see set_feature_pooling
        self._side_stream = None
        # sort the batch's row keys on a side stream during forward (see _prepare_backward)
        # "auto": only for lookups small enough to leave CUs idle (measured on MI355X: at 1.7 M ids the
        # side-stream sort steals bandwidth from the GEMMs, -2 %; at 213 K ids it hides, +4 %)
        self.overlap_backward_sort = os.environ.get("TBE_OVERLAP_SORT", "auto")
        self.overlap_backward_sort_max_ids = 1 << 22  # the one-launch-per-pass sort no longer steals GEMM bandwidth
        # ... in EAGER steps.  In the explicit (HIP-graph) step the dense segments replay back to back and a concurrent sort
        # costs more than it hides once it is big: one-rank rehearsal, overlap on / off, ms per step at per-rank batch
        # 4096: 0.936 / 0.954, 8192: 1.736 / 1.740, 16384: 2.943 / 2.880, 32768: 5.72 / 5.26 (ids per lookup 53 K ... 491 K;
        # eager at 65 536: 8.455 / 8.544).  lookup_no_autograd() therefore overlaps only below this many ids.
        self.overlap_backward_sort_max_ids_explicit = int(os.environ.get("TBE_OVERLAP_SORT_EXPLICIT_MAX_IDS", "150000"))

    def __getstate__(self):
        # copy.deepcopy / pickling (model_parallel.py:294-298 deep-copies sharded modules): HIP streams
        # cannot be copied and device-pointer tables must be rebuilt for the copy's own storage
        d = self.__dict__.copy()
        d["_side_stream"] = None
        d["_layout"] = _Layout()
        d["_pinned_key"] = None
        return d

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
        (written in place by batched_embedding_kernel.py:541-544 and embedding_lookup.py:70).
        With MANAGED_CACHING tables the HBM row cache is written back and emptied first, so the views
        are current and the caller may write them."""
        raise_on_faults("split_embedding_weights")
        if self._cache is not None:
            self._cache.flush(invalidate=True)
        return [self._table_view(self._flat_weights, t) for t in range(self.T)]

    # -- device metadata --------------------------------------------------------------------
    def _state_ptrs(self) -> Tuple[Optional[List[int]], Optional[List[int]]]:
        return None, None

    def _storage_key(self) -> Tuple:
        return (self._flat_weights("dev").data_ptr(), self._flat_weights("uvm").data_ptr())

    def _get_layout(self) -> _Layout:
        lay = self._real_layout()
        if self._cache is not None and self._cache.rows is not None:
            return self._cache.redirect(lay)
        return lay

    def _ensure_pinned(self) -> None:
        """Host-resident (MANAGED*) buffers must be pinned to be addressable by the kernels; a
        copy.deepcopy of the module (model_parallel.py:294-298) yields pageable copies — re-pin them."""
        names = [n for n, b in self._buffers.items() if n.endswith("_uvm") and b is not None and b.numel() > 0
                 and b.device.type == "cpu"]
        key = tuple(self._buffers[n].data_ptr() for n in names)
        if key == getattr(self, "_pinned_key", None):
            return
        for n in names:
            if not self._buffers[n].is_pinned():
                self._buffers[n] = self._buffers[n].pin_memory()
        self._pinned_key = tuple(self._buffers[n].data_ptr() for n in names)

    def _state_key(self) -> Tuple:
        """Addresses of the flat optimizer-state buffers (what _state_ptrs() is derived from)."""
        return ()

    def _real_layout(self) -> _Layout:
        self._ensure_pinned()
        # per step: a handful of data_ptr() calls; the per-table address lists are rebuilt only when a buffer moved
        key = (self._storage_key(), self._state_key(), self._row_windows, self._feature_pooling)
        lay = self._layout
        if lay.key == key:
            return lay
        s0, s1 = self._state_ptrs()
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
        lay.feat_pooling = i32(self._feature_pooling) if self._feature_pooling is not None else None
        lay.feat_window = None
        if self._row_windows is not None:
            lay.feat_window = i64([x for pair in zip(*self._row_windows) for x in pair])
        lay.feat_row_base = i64([self.row_base[t] for t in ftm])
        lay.feat_state0 = i64([s0[t] for t in ftm]) if s0 is not None else None
        lay.feat_state1 = i64([s1[t] for t in ftm]) if s1 is not None else None
        lay.key = key
        return lay

    def set_feature_pooling(self, modes: Optional[List["PoolingMode"]]) -> None:
        """Per-FEATURE pooling (SUM / MEAN) inside one module: tables of both pooling types share one lookup and one
        backward (the reference builds one TBE per pooling type and concatenates the results:
        embedding_sharding.py:393-490, embedding_lookup.py:219-253).  The module's pooling_mode becomes MEAN; features
        listed as SUM are not divided.  None restores the uniform mode given at construction."""
        if modes is None:
            self._feature_pooling = None
            self.pooling_mode = getattr(self, "_ctor_pooling_mode", self.pooling_mode)
            return
        modes = [PoolingMode(int(m)) for m in modes]
        if len(modes) != self.F or any(m == PoolingMode.NONE for m in modes) or self.pooling_mode == PoolingMode.NONE:
            raise ValueError("set_feature_pooling: one of SUM / MEAN per feature, on a pooled module")
        if not hasattr(self, "_ctor_pooling_mode"):
            self._ctor_pooling_mode = self.pooling_mode
        if all(m == modes[0] for m in modes):
            self._feature_pooling, self.pooling_mode = None, modes[0]
            return
        self._feature_pooling = tuple(int(m) for m in modes)
        self.pooling_mode = PoolingMode.MEAN

    def set_row_windows(self, first_rows: Optional[List[int]], global_rows: Optional[List[int]] = None) -> None:
        """Row-wise shards fed with un-bucketized GLOBAL ids: feature f's table holds global rows
        [first_rows[f], first_rows[f] + rows); ids of other shards (inside [0, global_rows[f])) are skipped
        silently, ids outside [0, global_rows[f]) count as bounds errors (include/tbe_hip.h `feat_window`).
        The reference bucketizes instead (embedding_sharding.py:121-184).  None switches windows off."""
        if first_rows is None:
            self._row_windows = None
            return
        if len(first_rows) != self.F or len(global_rows) != self.F:
            raise ValueError("set_row_windows: one (first row, global rows) pair per feature")
        self._row_windows = (tuple(int(x) for x in first_rows), tuple(int(x) for x in global_rows))

    def _errors(self) -> torch.Tensor:
        if self._bounds_errors is None or self._bounds_errors.device != self.current_device:
            self._bounds_errors = torch.zeros(1, dtype=torch.int32, device=self.current_device)
        return self._bounds_errors

    def _errors_ptr(self) -> Optional[int]:
        """The counter the kernels increment, or NULL for BoundsCheckMode.IGNORE / NONE (an out-of-range id still
        contributes a zero row and is never dereferenced: this build has no unchecked mode)."""
        if getattr(self, "bounds_check_mode", BoundsCheckMode.WARNING) in (BoundsCheckMode.IGNORE, BoundsCheckMode.NONE):
            return None
        return ptr(self._errors())

    def bounds_check_errors(self) -> int:
        """Number of out-of-range indices (and malformed bags) seen so far; they contribute zero rows.  Rows
        that belong to another rank's shard (`set_row_windows`) are NOT errors.  Syncs — and, with everything enqueued so
        far complete, raises KernelFaultError if a sort inside a backward / prefetch gave up (_lib.raise_on_faults)."""
        n = int(self._errors().item())
        raise_on_faults("bounds_check_errors")
        return n

    def _enforce_bounds_check_mode(self) -> None:
        """BoundsCheckMode.FATAL: raise as soon as a lookup has seen a bad id (costs a host sync per call);
        WARNING: warn once per new batch of errors is left to the caller reading bounds_check_errors()."""
        if getattr(self, "bounds_check_mode", BoundsCheckMode.WARNING) != BoundsCheckMode.FATAL:
            return
        n = self.bounds_check_errors()
        seen = getattr(self, "_fatal_seen", 0)
        if n > seen:
            self._fatal_seen = n
            raise RuntimeError(f"BoundsCheckMode.FATAL: {n - seen} out-of-range indices or malformed bags in this lookup")

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
                                              self._errors_ptr(), stream_ptr(dev)),
                    "tbe_forward_nobag_f32",
                )
                return out
            if into is not None:
                # caller-provided buffer + addressing (feat_out_offset, row stride): several lookups
                # can fill disjoint column blocks of one [B, sum D] matrix without a cat
                out, out_off, stride = into
                if (out.dtype != torch.float32 or not out.is_contiguous() or out.device != dev
                        or out.numel() < B * stride or out_off.numel() != self.F):
                    raise RuntimeError("forward_into: `out` must be a contiguous float32 buffer of at least "
                                       f"B * row_stride = {B * stride} elements on {dev}, with one offset per feature")
            else:
                out_off, stride, shape = self._pooled_layout(B)
                out = torch.empty(shape, dtype=torch.float32, device=dev)
            check(
                lib.tbe_forward_pooled_f32(ptr(lay.feat_weights), ptr(lay.feat_D),
                                           ptr(out_off), ptr(lay.feat_rows), self.F, B,
                                           self.max_D, ptr(indices), N, ptr(offsets),
                                           ptr(per_sample_weights), int(self.pooling_mode), ptr(lay.feat_pooling), ptr(out),
                                           stride, self._errors_ptr(), ptr(lay.feat_window), stream_ptr(dev)),
                "tbe_forward_pooled_f32",
            )
        return out

    def _prepare_or_defer(self, ctx, indices, offsets, B: int, weighted: bool):
        """The side-stream sort of this lookup's backward: started now, or — with `defer_backward_sort` — when the
        owner calls launch_deferred_backward_sort() (a model that knows its schedule puts the sort next to its
        MFMA-bound GEMMs instead of next to HBM-bound kernels: DESIGN.md §3)."""
        if not getattr(self, "defer_backward_sort", False):
            return self._prepare_backward(indices, offsets, B, weighted)
        self._deferred_sort = (ctx, indices, offsets, B, weighted)
        return None

    def launch_deferred_backward_sort(self) -> bool:
        """Starts the deferred sort of the latest forward (if any) on the side stream, ordered after the work
        enqueued so far on the current stream.  Without this call the backward sorts inline (fused call)."""
        d = getattr(self, "_deferred_sort", None)
        if d is None:
            return False
        self._deferred_sort = None
        ctx, indices, offsets, B, weighted = d
        ctx.prepared = self._prepare_backward(indices, offsets, B, weighted)
        return True

    def _prepare_backward(self, indices, offsets, B: int, weighted: bool = False):
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
                raise RuntimeError(f"TBE backward: {N} ids in one call is beyond the limit of 2^29 - 1 (include/tbe_hip.h): "
                                   "split the batch")
            ws = workspace(nbytes, dev)
            side = self._side_stream
            if side is None or side.device != dev:
                side = self._side_stream = _streams.side_stream(dev)
            cur = torch.cuda.current_stream(dev)
            side.wait_stream(cur)
            for t in (ws, indices, offsets):
                t.record_stream(side)
            check(
                lib.tbe_backward_prepare(ptr(lay.feat_rows), ptr(lay.feat_row_base), self.F, B, self.max_D,
                                         self.key_bits, ptr(indices), N, ptr(offsets), int(self.pooling_mode),
                                         _FLAG_WEIGHTED if weighted else 0, ptr(ws), ws.numel(),
                                         self._errors_ptr(), ptr(lay.feat_window), side.cuda_stream),
                "tbe_backward_prepare",
            )
            ev = torch.cuda.Event()
            ev.record(side)
        return ws, ev

    def _backward_impl(self, grad_out, indices, offsets, per_sample_weights, B: int,
                       opt: OptimizerArgs, state0_override: Optional[torch.Tensor] = None,
                       prepared=None, layout=None, state0_aligned: bool = False) -> None:
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
        # (with state0_override — the dense-gradient backward — the state bases are rows of a gradient buffer laid out like
        # the weights: aligned iff that buffer is, which the caller asserts through `state0_aligned`)
        flags = 1 if (len(set(self.dims_per_table)) == 1 and self.max_D % 4 == 0 and stride % 4 == 0
                      and (state0_override is None or state0_aligned)) else 0
        if per_sample_weights is not None:
            flags |= _FLAG_WEIGHTED  # the sort payload then carries positions too (set in prepare as well)
        with torch.cuda.device(dev):
            if prepared is not None:
                ws, ev = prepared
                torch.cuda.current_stream(dev).wait_event(ev)
                check(
                    lib.tbe_backward_apply_f32(ptr(lay.feat_weights), ptr(lay.feat_D), ptr(out_off),
                                               ptr(lay.feat_rows), ptr(lay.feat_row_base), ptr(feat_state0),
                                               ptr(lay.feat_state1), self.F, B, self.max_D, self.key_bits,
                                               ptr(indices), N, ptr(offsets), ptr(per_sample_weights),
                                               int(self.pooling_mode), ptr(lay.feat_pooling), ptr(grad_out), stride, opt, flags,
                                               ptr(ws), ws.numel(), stream_ptr(dev)),
                    "tbe_backward_apply_f32",
                )
                return
            nbytes = lib.tbe_backward_workspace_bytes(N, self.F, B, self.max_D, self.key_bits)
            if nbytes == 0:
                raise RuntimeError(f"TBE backward: {N} ids in one call is beyond the limit of 2^29 - 1 (include/tbe_hip.h): "
                                   "split the batch")
            ws = workspace(nbytes, dev)
            check(
                lib.tbe_backward_fused_f32(ptr(lay.feat_weights), ptr(lay.feat_D),
                                           ptr(out_off), ptr(lay.feat_rows),
                                           ptr(lay.feat_row_base), ptr(feat_state0),
                                           ptr(lay.feat_state1), self.F, B,
                                           self.max_D, self.key_bits, ptr(indices), N, ptr(offsets),
                                           ptr(per_sample_weights), int(self.pooling_mode), ptr(lay.feat_pooling),
                                           ptr(grad_out), stride, opt, flags, ptr(ws), ws.numel(),
                                           self._errors_ptr(), ptr(lay.feat_window), stream_ptr(dev)),
                "tbe_backward_fused_f32",
            )


class LookupRecord:
    """What a lookup made through `lookup_no_autograd` leaves for `backward_no_autograd` (the explicit counterpart of an
    autograd context: a train step that knows its own schedule — models/dlrm.py, HIP-graph mode — drives forward and
    backward itself and skips the autograd engine, its per-node Python hand-offs and its worker thread)."""

    __slots__ = ("indices", "offsets", "per_sample_weights", "B", "layout", "prepared")

    def __init__(self, indices, offsets, per_sample_weights, B, layout) -> None:
        self.indices, self.offsets, self.per_sample_weights, self.B, self.layout = indices, offsets, per_sample_weights, B, layout
        self.prepared = None


class _FusedLookupInto(torch.autograd.Function):
    """Like _FusedLookup, but writes its column blocks into a caller-provided [B, stride] buffer."""

    @staticmethod
    def forward(ctx, out, placeholder, module, indices, offsets, per_sample_weights, B, out_off, stride, prepare):
        ctx.module, ctx.B, ctx.layout = module, B, (out_off, stride)
        ctx.save_for_backward(indices, offsets, per_sample_weights)
        module._forward_impl(indices, offsets, per_sample_weights, B, into=(out, out_off, stride))
        ctx.prepared = module._prepare_or_defer(ctx, indices, offsets, B, per_sample_weights is not None) if prepare else None
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
        if module._cache is not None:
            module._cache.after_backward()
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
        state0 = module._dense_grad_ptrs(grad_w)
        opt = OptimizerArgs(_OPT_DENSE_GRAD, 0.0, 0.0, 0.0, 0.0, 0.0, 1)
        module._backward_impl(grad_out, indices, offsets, psw, ctx.B, opt, state0_override=state0, layout=ctx.layout,
                              state0_aligned=grad_w.data_ptr() % 16 == 0)
        return (grad_out, grad_w) + (None,) * 7


class _FusedLookup(torch.autograd.Function):
    """forward = TBE gather/pool; backward = coalesce + fused optimizer (no weight grad)."""

    @staticmethod
    def forward(ctx, placeholder, module, indices, offsets, per_sample_weights, B, prepare):
        ctx.module = module
        ctx.B = B
        ctx.save_for_backward(indices, offsets, per_sample_weights)
        out = module._forward_impl(indices, offsets, per_sample_weights, B)
        ctx.prepared = module._prepare_or_defer(ctx, indices, offsets, B, per_sample_weights is not None) if prepare else None
        return out

    @staticmethod
    def backward(ctx, grad_out):
        indices, offsets, psw = ctx.saved_tensors
        module = ctx.module
        module.iter += 1
        module._backward_impl(grad_out, indices, offsets, psw, ctx.B, module._optimizer_struct(),
                              prepared=ctx.prepared)
        ctx.prepared = None
        if module._cache is not None:
            module._cache.after_backward()
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
        # weight decay: one form per optimizer is implemented (parity unpinned: fbgemm's is absent);
        # any other request raises instead of silently computing something else
        wdm = WeightDecayMode(int(weight_decay_mode))
        if weight_decay != 0.0:
            if optimizer == OptimType.EXACT_ROWWISE_ADAGRAD and wdm != WeightDecayMode.L2:
                raise NotImplementedError("EXACT_ROWWISE_ADAGRAD implements weight_decay as L2 (g += weight_decay * w before "
                                          f"the row-wise moment): pass weight_decay_mode=WeightDecayMode.L2, not {wdm.name}")
            if optimizer == OptimType.ADAM and wdm == WeightDecayMode.L2:
                raise NotImplementedError("ADAM implements weight_decay in the decoupled form (w -= lr * weight_decay * w, as "
                                          "examples/bert4rec/bert4rec_main.py:488-491 uses it with the default mode); "
                                          "WeightDecayMode.L2 is not implemented")
            if optimizer in (OptimType.EXACT_SGD, OptimType.EXACT_ADAGRAD):
                raise NotImplementedError(f"weight_decay is not implemented for {optimizer}")
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
        # HBM row cache for the MANAGED_CACHING tables (SGD / row-wise Adagrad; tables whose optimizer
        # keeps per-element state are served straight from host memory, like MANAGED)
        cached = [t for t, loc in enumerate(self.locations) if loc == EmbeddingLocation.MANAGED_CACHING]
        if cached and code in (0, 1) and self.current_device.type == "cuda":
            self._cache = _RowCache(self, cached, cache_load_factor, cache_sets, rowwise_state=rowwise)
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

    def _state_key(self):
        code = _OPT_CODE[self.optimizer]
        if code == 0:
            return ()
        names = ("momentum1", "momentum2") if code == 2 else ("momentum1",)
        return tuple(self._state_flat(n, p).data_ptr() for n in names for p in ("dev", "uvm"))

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
        if self._cache is not None:
            self._cache.flush(invalidate=True)
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
        """Called by the fused optimizer's step() / zero_grad() every train step (batched_embedding_kernel.py:250-257):
        the one per-step host call of the module outside forward / backward, hence also where a fault reported by an
        earlier step's kernels surfaces (no sync; _lib.raise_on_faults)."""
        self.optimizer_args.learning_rate = float(lr)
        raise_on_faults("set_learning_rate")

    def flush(self) -> None:
        """Write-back of the HBM row cache of MANAGED_CACHING tables to their host tables
        (batched_embedding_kernel.py:563, 664); the cache stays warm.  Raises KernelFaultError if a kernel reported a
        fault since the last check (what is about to be saved would be wrong)."""
        raise_on_faults("flush")
        if self._cache is not None:
            self._cache.flush(invalidate=False)

    def cache_stats(self) -> Optional[dict]:
        """Hit / miss / eviction counters of the HBM row cache (None without MANAGED_CACHING tables). Syncs."""
        return self._cache.stats() if self._cache is not None else None

    def _optimizer_struct(self) -> OptimizerArgs:
        a = self.optimizer_args
        return OptimizerArgs(_OPT_CODE[self.optimizer], a.learning_rate, a.eps, a.weight_decay,
                             a.beta1, a.beta2, max(self.iter, 1))

    def forward(self, indices: torch.Tensor, offsets: torch.Tensor,
                per_sample_weights: Optional[torch.Tensor] = None,
                feature_requires_grad: Optional[torch.Tensor] = None) -> torch.Tensor:
        indices, offsets, per_sample_weights, B = self._check_inputs(indices, offsets, per_sample_weights)
        if self._cache is not None:
            indices = self._cache.prefetch(self._real_layout(), indices, offsets, B, torch.is_grad_enabled())
        mode = self.overlap_backward_sort
        prepare = torch.is_grad_enabled() and (
            mode in (True, "1") or (mode == "auto" and indices.numel() <= self.overlap_backward_sort_max_ids))
        out = _FusedLookup.apply(self.placeholder_autograd_tensor, self, indices, offsets,
                                 per_sample_weights, B, prepare)
        self._enforce_bounds_check_mode()
        return out

    def forward_into(self, out: torch.Tensor, out_offsets: torch.Tensor, row_stride: int, indices: torch.Tensor,
                     offsets: torch.Tensor, per_sample_weights: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Pooled lookup whose feature blocks land at `out[b * row_stride + out_offsets[f] + d]` of the
        given buffer (returned, marked dirty for autograd)."""
        indices, offsets, per_sample_weights, B = self._check_inputs(indices, offsets, per_sample_weights)
        if self._cache is not None:
            indices = self._cache.prefetch(self._real_layout(), indices, offsets, B, torch.is_grad_enabled())
        mode = self.overlap_backward_sort
        prepare = torch.is_grad_enabled() and (
            mode in (True, "1") or (mode == "auto" and indices.numel() <= self.overlap_backward_sort_max_ids))
        out = _FusedLookupInto.apply(out, self.placeholder_autograd_tensor, self, indices, offsets,
                                     per_sample_weights, B, out_offsets, int(row_stride), prepare)
        self._enforce_bounds_check_mode()
        return out

    def lookup_no_autograd(self, indices: torch.Tensor, offsets: torch.Tensor,
                           per_sample_weights: Optional[torch.Tensor] = None, into=None):
        """The training forward without an autograd node: returns (output, LookupRecord).  `into` = (buffer,
        per-feature offsets, row stride) as in forward_into.  The caller owes exactly one
        backward_no_autograd(record, grad) per call (the side-stream sort of the backward is already running)."""
        indices, offsets, per_sample_weights, B = self._check_inputs(indices, offsets, per_sample_weights)
        if self._cache is not None:
            indices = self._cache.prefetch(self._real_layout(), indices, offsets, B, True)
        rec = LookupRecord(indices, offsets, per_sample_weights, B, (into[1], int(into[2])) if into is not None else None)
        out = self._forward_impl(indices, offsets, per_sample_weights, B, into=into)
        mode = self.overlap_backward_sort
        if mode in (True, "1") or (mode == "auto" and indices.numel() <= self.overlap_backward_sort_max_ids_explicit):
            rec.prepared = self._prepare_or_defer(rec, indices, offsets, B, per_sample_weights is not None)
        self._enforce_bounds_check_mode()
        return out, rec

    def backward_no_autograd(self, rec: "LookupRecord", grad_out: torch.Tensor) -> None:
        """Coalesced gradient + fused optimizer update for the lookup `rec` describes (what _FusedLookup.backward does)."""
        self.iter += 1
        self._backward_impl(grad_out, rec.indices, rec.offsets, rec.per_sample_weights, rec.B, self._optimizer_struct(),
                            prepared=rec.prepared, layout=rec.layout)
        rec.prepared = None
        if self._cache is not None:
            self._cache.after_backward()


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
        state0 = module._dense_grad_ptrs(grad_w)
        opt = OptimizerArgs(_OPT_DENSE_GRAD, 0.0, 0.0, 0.0, 0.0, 0.0, 1)
        module._backward_impl(grad_out, indices, offsets, psw, ctx.B, opt, state0_override=state0,
                              state0_aligned=grad_w.data_ptr() % 16 == 0)
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

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # inside a sharded collection the tables travel as `embedding_bags.<table>.weight` (loaded by the owner)
        if getattr(self, "_owned_by_sharded_module", False):
            return
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def _dense_grad_ptrs(self, grad_w: torch.Tensor) -> torch.Tensor:
        """Per-feature base addresses inside a dense gradient buffer, computed on the device (a
        host-built table would cost a blocking H2D copy in every backward)."""
        rel = getattr(self, "_dense_rel", None)
        if rel is None or rel.device != grad_w.device:
            rel = torch.tensor([4 * self.weights_offsets[t] for t in self.feature_table_map], dtype=torch.int64).to(grad_w.device)
            self._dense_rel = rel
        return rel + grad_w.data_ptr()

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

    def lookup_no_autograd(self, indices: torch.Tensor, offsets: torch.Tensor,
                           per_sample_weights: Optional[torch.Tensor] = None, into=None):
        """Forward without an autograd node: (output, LookupRecord); see the fused class."""
        indices, offsets, per_sample_weights, B = self._check_inputs(indices, offsets, per_sample_weights)
        rec = LookupRecord(indices, offsets, per_sample_weights, B, (into[1], int(into[2])) if into is not None else None)
        out = self._forward_impl(indices, offsets, per_sample_weights, B, into=into)
        # the gradient-independent half of the backward (linearize + sort: 5 dependent launches, ~25 us of latency for the
        # ~90 K ids of the replicated tiny tables) starts now on the module's side stream, as the fused module's does
        mode = self.overlap_backward_sort
        if mode in (True, "1") or (mode == "auto" and indices.numel() <= self.overlap_backward_sort_max_ids_explicit):
            rec.prepared = self._prepare_or_defer(rec, indices, offsets, B, per_sample_weights is not None)
        return out, rec

    def backward_no_autograd(self, rec: "LookupRecord", grad_out: torch.Tensor,
                             into: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The dense gradient of `.weights` for the lookup `rec` describes (what _DenseLookup.backward returns).
        `into`: a persistent float32 buffer shaped like `.weights` (e.g. this parameter's slice of a flat gradient buffer)
        that receives the gradient instead of a fresh tensor — no allocation, and the per-feature address table is built
        once per buffer instead of by a kernel per step."""
        if into is not None:
            if (into.shape != self.weights.shape or into.dtype != torch.float32 or not into.is_contiguous()
                    or into.device != self.weights.device):
                raise RuntimeError("backward_no_autograd: `into` must be a contiguous float32 tensor shaped like .weights")
            grad_w = into.zero_()
            cached = getattr(self, "_dense_ptrs_cache", None)
            if cached is None or cached[0] != grad_w.data_ptr():
                cached = self._dense_ptrs_cache = (grad_w.data_ptr(), self._dense_grad_ptrs(grad_w))
            ptrs = cached[1]
        else:
            grad_w = torch.zeros_like(self.weights)
            ptrs = self._dense_grad_ptrs(grad_w)
        opt = OptimizerArgs(_OPT_DENSE_GRAD, 0.0, 0.0, 0.0, 0.0, 0.0, 1)
        self._backward_impl(grad_out, rec.indices, rec.offsets, rec.per_sample_weights, rec.B, opt,
                            state0_override=ptrs, prepared=rec.prepared, layout=rec.layout,
                            state0_aligned=grad_w.data_ptr() % 16 == 0)
        rec.prepared = None
        return grad_w
