"""ShardedEmbeddingBagCollection for MI355X: table-wise + row-wise sharding of an
EmbeddingBagCollection over the ranks of one node.

Interface follows torchrec/distributed/embeddingbag.py:226-486 (input_dist -> compute ->
output_dist, `compute_and_output_dist`, `fused_optimizer`, KeyedTensor result) and
EmbeddingBagCollectionSharder (:489-515).  The data path is re-designed for xGMI / HIP:

  reference (per sharding TYPE, embeddingbag.py:331-402)      this build (once, for all types)
  --------------------------------------------------------    ------------------------------------------
  kjt.permute + split                                          one gather of ids into send order
  RW: block_bucketize + 2-phase lengths/values a2a + D2H sync  ids of row-wise features go to every rank as
  TW: 2-phase lengths/values a2a + D2H sync                    GLOBAL rows; the kernels get each shard's row
  recat permute_2D on the receiver                             window and skip other shards' rows silently
                                                               (ids outside the TABLE still count as bounds
                                                               errors); with host-known pooling factors every
                                                               size is static: ONE a2a, no sync, no recat
                                                               (TBE consumes [src][feature][sample])
  one TBE per (type, group) + cat                              ONE TBE per rank (row-wise shards + table-wise
                                                               tables), output written a2a-ready
  TW: a2a + split/cat; RW: ring reduce-scatter; cat            ONE a2a, then tbe_pooled_exchange_unpack (copy
                                                               TW columns, sum RW partials in rank order)
  backward: recat copy + a2a / all-gather, grads / W           tbe_pooled_exchange_pack (x 1/W fused) + ONE a2a

Row-wise input dist has a second, selectable form (`rw_input_dist`): "bucketize" = the reference's — ids of row-wise
features bucketized by row block (block_bucketize_sparse_features, embedding_sharding.py:121-184), every rank receives only
its own block's ids as LOCAL rows, lengths exchange + one D2H read of the counts + values exchange.  "windows" (above) is
sync-free but sends, linearizes and sorts W x the row-wise ids; "auto" picks windows for a host-known pooling factor <= 2
and bucketize for longer or data-dependent bags (tools/rankbench.py, DESIGN.md §4 have the numbers).

Tiny tables can be DATA_PARALLEL (replicated): a dense-gradient TBE looks them up for the local
batch and writes straight into its columns of the output matrix (no exchange, no cat); their
gradient is all-reduced by DDP like any dense parameter (sharding/dp_sharding.py in the reference).

Forward issues the lookup and the a2a before the caller's dense work and waits afterwards
(`forward()` returns an awaitable), so the exchange overlaps the bottom MLP; autograd replays
the same overlap in reverse for the gradient exchange.
"""
import os
from typing import Any, Callable, Dict, Iterator, List, Optional, Tuple

import torch
import torch.distributed as dist
from torch import nn

from ..modules.embedding_configs import EmbeddingBagConfig, pooling_type_to_pooling_mode
from ..modules.embedding_modules import EmbeddingBagCollection
from ..profiling import label
from ..sparse.jagged_tensor import KeyedJaggedTensor, KeyedTensor
from . import _device_ops  # noqa: F401  (registers torch.ops.tbe_hip.*)
from .planner import rw_block_size, rw_shard_rows
from .types import Awaitable, LazyAwaitable, NoWait, ParameterSharding, ShardingEnv, ShardingType

GRADIENT_DIVISION = True  # torchrec/distributed/comm_ops.py:35-40
# Rehearsal switch: run the id / pooled all-to-all and the exchange kernels even at world_size 1
# (a one-rank RCCL group), so the N > 1 data path can be exercised on a one-GPU box.
FORCE_EXCHANGE = os.environ.get("TORCHREC_AMD_FORCE_EXCHANGE", "0") == "1"


def set_gradient_division(val: bool) -> None:
    global GRADIENT_DIVISION
    GRADIENT_DIVISION = val


def _gather_rows(mat: torch.Tensor, order: List[int], order_dev: torch.Tensor) -> torch.Tensor:
    """Rows `order` of a [keys, B * L] id / weight matrix as one contiguous [len(order), B * L] block (`order_dev`: the same
    list as a device int32 tensor).  On the GPU one launch of csrc/sparse_ops.hip copy_rows_kernel (16-byte vectors):
    torch.index_select picks a per-element kernel for this shape (87 us for 26 x 8192 ids, rocprof on MI355X) and a cat of
    row slices is no better (76 us)."""
    if not order:
        return mat.new_empty((0, mat.shape[1]))
    if order == list(range(order[0], order[0] + len(order))):
        return mat[order[0]:order[0] + len(order)]
    if os.environ.get("TORCHREC_AMD_GATHER") == "cat":
        runs, start = [], 0
        for i in range(1, len(order) + 1):
            if i == len(order) or order[i] != order[i - 1] + 1:
                runs.append(mat[order[start]:order[i - 1] + 1])
                start = i
        return torch.cat(runs, dim=0)
    if mat.is_cuda and mat.is_contiguous() and (mat.shape[1] * mat.element_size()) % 16 == 0 and mat.data_ptr() % 16 == 0:
        from . import _device_ops  # noqa: F401  (registers torch.ops.tbe_hip.*)
        return torch.ops.tbe_hip.copy_rows(mat, order_dev)
    return mat.index_select(0, order_dev.long())


def _default_dp_tbe_factory(specs, ftm, pooling_mode, device):
    from fbgemm_gpu.split_table_batched_embeddings_ops import DenseTableBatchedEmbeddingBagsCodegen

    with torch.cuda.device(device):
        return DenseTableBatchedEmbeddingBagsCodegen(list(specs), feature_table_map=ftm, pooling_mode=pooling_mode)


def _default_tbe_factory(specs, ftm, pooling_mode, device, fused_params):
    from fbgemm_gpu.split_table_batched_embeddings_ops import (
        ComputeDevice, EmbeddingLocation, SplitTableBatchedEmbeddingBagsCodegen)

    # compute kernel -> table location (embedding_types.py:57-76 compute_kernel_to_embedding_location)
    loc = {"batched_fused_uvm": EmbeddingLocation.MANAGED, "batched_fused_uvm_caching": EmbeddingLocation.MANAGED_CACHING}
    return SplitTableBatchedEmbeddingBagsCodegen(
        embedding_specs=[(s[0], s[1], loc.get(s[2] if len(s) > 2 else "", EmbeddingLocation.DEVICE), ComputeDevice.CUDA)
                         for s in specs],
        feature_table_map=ftm, pooling_mode=pooling_mode, device=device, **fused_params)


def _placement(rank: int, me: int, device: torch.device) -> str:
    """`rank:R/device` placement string of a shard (torch.distributed._shard): this rank's shards sit on its own device,
    a peer's on the device with the peer's index (one process per GPU of one node)."""
    if device.type != "cuda":
        return f"rank:{rank}/cpu"
    return f"rank:{rank}/cuda:{device.index if rank == me else rank}"


def wrap_sharded(local: Optional[torch.Tensor], global_size: List[int], shards: List[Tuple[List[int], List[int], int]], pg,
                 me: int, device: torch.device):
    """A torch ShardedTensor over this rank's `local` shard (None if it holds none) of a tensor of `global_size` cut into
    `shards` = [(offsets, sizes, rank), ...] — what the reference puts into state_dict() and the fused optimizer state for
    every sharded table when a process group exists (embedding_kernel.py:63-122, batched_embedding_kernel.py:166-246)."""
    from torch.distributed._shard.metadata import ShardMetadata as TorchShardMetadata
    from torch.distributed._shard.sharded_tensor import Shard, ShardedTensor, ShardedTensorMetadata, TensorProperties

    metas = [TorchShardMetadata(shard_offsets=list(o), shard_sizes=list(z), placement=_placement(r, me, device))
             for o, z, r in shards]
    mine = [Shard(local, m) for m, (_, _, r) in zip(metas, shards) if r == me and local is not None]
    dtype = local.dtype if local is not None else torch.float32
    meta = ShardedTensorMetadata(shards_metadata=metas, size=torch.Size(global_size),
                                 tensor_properties=TensorProperties(dtype=dtype, requires_grad=False))
    return ShardedTensor._init_from_local_shards_and_global_metadata(mine, meta, process_group=pg)


def unwrap_local(value):
    """A tensor out of a state_dict value: the local shard of a ShardedTensor, the tensor itself otherwise."""
    if hasattr(value, "local_shards") and hasattr(value, "metadata"):
        shards = value.local_shards()
        if len(shards) != 1:
            raise ValueError(f"expected exactly one local shard, found {len(shards)}")
        return shards[0].tensor
    return value


class _LocalTable:
    def __init__(self, cfg: EmbeddingBagConfig, local_rows: int, row_offset: int, row_wise: bool,
                 compute_kernel: str = "batched_fused") -> None:
        self.cfg, self.local_rows, self.row_offset, self.row_wise = cfg, local_rows, row_offset, row_wise
        self.compute_kernel = compute_kernel


class SparseFeaturesDist:
    """What input_dist hands to compute (embedding_types.py `SparseFeatures`, after the a2a):
    ids in [src rank][local feature][sample] order + offsets for the local TBE."""

    def __init__(self, values, offsets, weights, batch_size: int, dp=None) -> None:
        self.values, self.offsets, self.weights, self.batch_size = values, offsets, weights, batch_size
        self.dp = dp  # (values, offsets, weights) of the data-parallel features, local batch

    def record_stream(self, stream) -> None:
        for t in (self.values, self.offsets, self.weights) + (tuple(self.dp) if self.dp is not None else ()):
            if t is not None and t.is_cuda:
                t.record_stream(stream)


class _InputDistAwaitable(LazyAwaitable):
    def __init__(self, fn: Callable[[], SparseFeaturesDist]) -> None:
        super().__init__()
        self._fn = fn

    def _wait_impl(self) -> SparseFeaturesDist:
        return self._fn()


class _ExchangeReq(torch.autograd.Function):
    """Forward: start the pooled all-to-all (async).  Backward: wait for the gradient all-to-all
    that `_ExchangeWait.backward` started.  (Req/Wait split as comm_ops.py:462-605.)"""

    @staticmethod
    def forward(ctx, emb, state):
        ctx.state = state
        state.start_forward(emb)
        return state.recv_fwd

    @staticmethod
    def backward(ctx, _unused):
        return ctx.state.finish_backward(), None


class _ExchangeWait(torch.autograd.Function):
    @staticmethod
    def forward(ctx, recv, state):
        ctx.state = state
        return state.finish_forward()

    @staticmethod
    def backward(ctx, grad_out):
        ctx.state.start_backward(grad_out)
        return ctx.state.dummy_grad(), None


class _ExchangeState:
    """One step's pooled exchange: buffers, split sizes and in-flight work handles."""

    def __init__(self, owner: "ShardedEmbeddingBagCollection", B: int) -> None:
        self.o, self.B = owner, B
        self.lay = owner._exchange_layout(B)
        self.recv_fwd: Optional[torch.Tensor] = None
        self.work = None
        self.grad_recv: Optional[torch.Tensor] = None
        self.bwd_work = None
        self._dest: Optional[torch.Tensor] = None
        self._half_bwd_work: List[Any] = []
        self._half_send_keepalive: List[torch.Tensor] = []
        self.static = None

    def start_forward(self, emb: torch.Tensor, allow_static: bool = False) -> None:
        o, lay = self.o, self.lay
        # (the persistent buffers serve the explicit train step only: an eval forward between two steps must not receive
        # into the buffer that holds the next step's prefetched embeddings)
        static = o._static_exchange if (allow_static and o._static_exchange is not None
                                        and o._static_exchange["B"] == self.B) else None
        self.static = static  # persistent receive / send buffers: the owner's graphs unpack / pack (set_graph_exchange)
        self.recv_fwd = (static["recv_fwd"] if static is not None
                         else torch.empty(lay["recv_numel"], dtype=torch.float32, device=emb.device))
        with label("## alltoall_fwd_single ##"):  # comm_ops.py:489
            self.work = dist.all_to_all_single(self.recv_fwd, emb.reshape(-1), output_split_sizes=lay["recv_splits"],
                                               input_split_sizes=lay["send_splits"], group=o._pg, async_op=True)

    def output_destination(self) -> torch.Tensor:
        """The [B, sum D] tensor finish_forward() unpacks into: the consumer's own buffer (e.g. the static input of a
        HIP-graph segment: no copy downstream) or a fresh one.  Available before the all-to-all is done, so that the
        replicated tables' lookup can fill its columns meanwhile."""
        if self._dest is None:
            o = self.o
            buf = o._output_buffer
            self._dest = (o._alias_output_buffer(self.B) if buf is not None and buf.numel() == self.B * o._D_total
                          else torch.empty((self.B, o._D_total), dtype=torch.float32, device=self.recv_fwd.device))
        return self._dest

    def wait_forward(self) -> None:
        """Static-exchange mode: the current stream waits for the pooled all-to-all; the owner's forward graph unpacks."""
        with label("## alltoall_fwd_wait ##"):
            self.work.wait()
        self.work = None

    def start_backward_packed(self) -> None:
        """Static-exchange mode: the owner's backward graph has packed the gradient into the persistent send buffer."""
        o, lay, st = self.o, self.lay, self.static
        self.grad_recv = st["grad_recv"]
        with label("## alltoall_bwd_single ##"):  # comm_ops.py:591
            self.bwd_work = dist.all_to_all_single(self.grad_recv, st["send_bwd"], output_split_sizes=lay["send_splits"],
                                                   input_split_sizes=lay["recv_splits"], group=o._pg, async_op=True)

    def finish_forward(self) -> torch.Tensor:
        with label("## alltoall_fwd_wait ##"):
            self.work.wait()
        self.work = None
        lay = self.lay
        if self._dest is not None or (self.o._output_buffer is not None
                                      and self.o._output_buffer.numel() == self.B * self.o._D_total):
            dest, self._dest = self.output_destination(), None
            return torch.ops.tbe_hip.pooled_exchange_unpack_into(
                self.recv_fwd, lay["feat_out_col"], lay["feat_src"], lay["feat_slab_col"], lay["slab_offset"],
                lay["slab_stride"], self.B, self.o._D_total, self.o._vec_ok, 1.0, dest)
        return torch.ops.tbe_hip.pooled_exchange_unpack(
            self.recv_fwd, lay["feat_out_col"], lay["feat_src"], lay["feat_slab_col"], lay["slab_offset"],
            lay["slab_stride"], self.B, self.o._D_total, self.o._vec_ok, 1.0)

    def dummy_grad(self) -> torch.Tensor:
        # gradient placeholder for recv_fwd: its real gradient travels through the all-to-all
        return self.recv_fwd.new_zeros(1).expand(self.recv_fwd.shape)

    def start_backward(self, grad_out: torch.Tensor) -> None:
        o, lay = self.o, self.lay
        scale = 1.0 / o._world_size if GRADIENT_DIVISION else 1.0
        send = torch.ops.tbe_hip.pooled_exchange_pack(
            grad_out, lay["feat_out_col"], lay["feat_src"], lay["feat_slab_col"], lay["slab_offset"],
            lay["slab_stride"], lay["recv_numel"], o._vec_ok, scale)
        self.grad_recv = torch.empty(lay["send_numel"], dtype=torch.float32, device=grad_out.device)
        with label("## alltoall_bwd_single ##"):  # comm_ops.py:591
            self.bwd_work = dist.all_to_all_single(self.grad_recv, send, output_split_sizes=lay["send_splits"],
                                                   input_split_sizes=lay["recv_splits"], group=o._pg, async_op=True)
        self._send_keepalive = send

    def finish_backward(self) -> torch.Tensor:
        with label("## alltoall_bwd_wait ##"):
            if self.bwd_work is not None:
                self.bwd_work.wait()
            for w in self._half_bwd_work:
                w.wait()
        self.bwd_work = None
        self._half_bwd_work = []
        self._send_keepalive = None
        return self.grad_recv.view(self.o._world_size * self.B, self.o._D_local)

    # ---- the same exchange in two half-batches (rows [0, B/2) and [B/2, B) of every rank's local batch) -------------
    # Each half is its own all-to-all (list form: the pieces are row ranges of the slabs, contiguous but not adjacent),
    # laid out exactly like a batch of B/2: unpack / pack run with the B/2 layout on the half's rows.  The caller runs
    # half 0's dense work while half 1's embeddings are on the links, and half 1's while half 0's gradients are
    # (models/dlrm.py explicit step): on point-to-point xGMI links the exchange is otherwise exposed.
    def start_forward_halves(self, emb: torch.Tensor) -> None:
        o, W, B = self.o, self.o._world_size, self.B
        if B % 2:
            raise RuntimeError("half-batch exchange: the per-rank batch must be even")
        Bh = B // 2
        layh = self.lay_half = o._exchange_layout(Bh)
        src = emb.view(W, B, o._D_local)
        self._half_recv, self._half_work = [], []
        for h in range(2):
            recv = torch.empty(layh["recv_numel"], dtype=torch.float32, device=emb.device)
            with label("## alltoall_fwd_single ##"):  # comm_ops.py:489
                work = dist.all_to_all(list(recv.split(layh["recv_splits"])), [src[d, h * Bh:(h + 1) * Bh] for d in range(W)],
                                       group=o._pg, async_op=True)
            self._half_recv.append(recv)
            self._half_work.append(work)
        self.recv_fwd = self._half_recv[0]  # (device of the destination)
        self._emb_keepalive = emb

    def finish_forward_half(self, h: int) -> torch.Tensor:
        """Waits for half h and unpacks it into rows [h B/2, (h + 1) B/2) of the destination; returns those rows."""
        with label("## alltoall_fwd_wait ##"):
            self._half_work[h].wait()
        self._half_work[h] = None
        layh, Bh, o = self.lay_half, self.B // 2, self.o
        rows = self.output_destination()[h * Bh:(h + 1) * Bh]
        torch.ops.tbe_hip.pooled_exchange_unpack_into(
            self._half_recv[h], layh["feat_out_col"], layh["feat_src"], layh["feat_slab_col"], layh["slab_offset"],
            layh["slab_stride"], Bh, o._D_total, o._vec_ok, 1.0, rows)
        self._half_recv[h] = None
        if h == 1:
            self._emb_keepalive = None
        return rows

    def start_backward_half(self, h: int, grad_rows: torch.Tensor) -> None:
        """grad_rows: [B/2, sum D] gradient of rows [h B/2, (h + 1) B/2) of the pooled output, contiguous."""
        o, W, B = self.o, self.o._world_size, self.B
        Bh, layh = B // 2, self.lay_half
        scale = 1.0 / W if GRADIENT_DIVISION else 1.0
        send = torch.ops.tbe_hip.pooled_exchange_pack(
            grad_rows, layh["feat_out_col"], layh["feat_src"], layh["feat_slab_col"], layh["slab_offset"],
            layh["slab_stride"], layh["recv_numel"], o._vec_ok, scale)
        if self.grad_recv is None:
            self.grad_recv = torch.empty(self.lay["send_numel"], dtype=torch.float32, device=grad_rows.device)
        dst = self.grad_recv.view(W, B, o._D_local)
        with label("## alltoall_bwd_single ##"):  # comm_ops.py:591
            work = dist.all_to_all([dst[s, h * Bh:(h + 1) * Bh] for s in range(W)], list(send.split(layh["recv_splits"])),
                                   group=o._pg, async_op=True)
        self._half_bwd_work.append(work)
        self._half_send_keepalive.append(send)


class _OutputAwaitable(LazyAwaitable):
    def __init__(self, fn: Callable[[], KeyedTensor]) -> None:
        super().__init__()
        self._fn = fn

    def _wait_impl(self) -> KeyedTensor:
        return self._fn()


class EmbeddingFusedOptimizer:
    """Optimizer facade over the TBE's in-backward optimizer (torchrec/distributed/batched_embedding_kernel.py:53-257):
    `step()` / `zero_grad()` only push the learning rate; parameters are the local table shards.  Keys follow the
    reference: parameter `<prefix><table>.weight`, its state `{"<table>.momentum1": ..., "<table>.momentum2": ...}`
    nested under the parameter key (:241-249).  Tensors are views of the module's storage; for MANAGED_CACHING
    tables the HBM row cache is written back before they are handed out."""

    def __init__(self, emb_module, table_names: List[str], key_prefix: str = "", wrap=None) -> None:
        self._wrap = wrap  # (table name, local tensor) -> ShardedTensor | tensor; None = plain tensors
        self._emb_module = emb_module
        self._table_names = list(table_names)
        self._key_prefix = key_prefix
        self._save_param_groups = False
        self.param_groups = [{"params": [], "lr": emb_module.optimizer_args.learning_rate}]
        self._refresh()

    def _refresh(self) -> None:
        """(Re)reads the weight / state views: split_* write the row cache back and empty it first."""
        m, pre = self._emb_module, self._key_prefix
        self._params = {f"{pre}{n}.weight": w for n, w in zip(self._table_names, m.split_embedding_weights())}
        self._state = {f"{pre}{n}.weight": {f"{n}.momentum{i + 1}": s for i, s in enumerate(st)}
                       for n, st in zip(self._table_names, m.split_optimizer_states())}
        self.param_groups[0]["params"] = list(self._params.values())

    def _cached(self) -> bool:
        return getattr(self._emb_module, "_cache", None) is not None

    @property
    def params(self) -> Dict[str, torch.Tensor]:
        if self._cached():
            self._refresh()
        return self._params

    @property
    def state(self) -> Dict[str, Dict[str, torch.Tensor]]:
        if self._cached():
            self._refresh()
        return self._state

    def zero_grad(self, set_to_none: bool = False) -> None:
        self._emb_module.set_learning_rate(self.param_groups[0]["lr"])

    def step(self, closure: Any = None) -> None:
        self._emb_module.set_learning_rate(self.param_groups[0]["lr"])

    def save_param_groups(self, save: bool) -> None:
        self._save_param_groups = save

    def state_dict(self) -> Dict[str, Any]:
        w = self._wrap
        out: Dict[str, Any] = {"state": {k: {kk: (w(kk.rsplit(".", 1)[0], vv) if w is not None else vv) for kk, vv in v.items()}
                                         for k, v in self.state.items()}}
        if self._save_param_groups:
            out["param_groups"] = [{"params": sorted(self._params.keys()), "lr": self.param_groups[0]["lr"]}]
        return out

    def load_state_dict(self, state_dict: Dict[str, Any]) -> None:
        mine = self.state
        new = state_dict["state"]
        if set(new.keys()) != set(mine.keys()):
            raise ValueError(f"fused optimizer state keys differ: {sorted(mine.keys())} vs {sorted(new.keys())}")
        with torch.no_grad():
            for k, st in new.items():
                if set(st.keys()) != set(mine[k].keys()):
                    raise ValueError(f"fused optimizer state of {k}: {sorted(mine[k].keys())} vs {sorted(st.keys())}")
                for name, t in st.items():
                    mine[k][name].copy_(unwrap_local(t))
        if "param_groups" in state_dict and state_dict["param_groups"]:
            self.param_groups[0]["lr"] = state_dict["param_groups"][0].get("lr", self.param_groups[0]["lr"])
            self._emb_module.set_learning_rate(self.param_groups[0]["lr"])


class ShardedEmbeddingBagCollection(nn.Module):
    def __init__(
        self,
        module: EmbeddingBagCollection,
        table_name_to_parameter_sharding: Dict[str, ParameterSharding],
        env: ShardingEnv,
        fused_params: Optional[Dict[str, Any]] = None,
        device: Optional[torch.device] = None,
        tbe_factory: Optional[Callable] = None,
        dp_tbe_factory: Optional[Callable] = None,
        rw_input_dist: Optional[str] = None,
    ) -> None:
        super().__init__()
        self._env = env
        rw_input_dist = rw_input_dist or os.environ.get("TORCHREC_AMD_RW_INPUT_DIST", "auto")
        if rw_input_dist not in ("auto", "windows", "bucketize"):
            raise ValueError("rw_input_dist must be auto, windows or bucketize")
        self._rw_input_dist = rw_input_dist
        W, me = env.world_size, env.rank
        self._world_size, self._rank = W, me
        self._exchange = W > 1 or (FORCE_EXCHANGE and env.process_group is not None)
        self._device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        # the exchanges' group: the environment's, or — for an RCCL group whose collective stream is not high-priority, i.e. a
        # launcher written for CUDA — a second communicator whose stream is, clear of the compute stream's hardware queue
        from .comm import exchange_group

        self._pg = exchange_group(env.process_group, self._device) if (self._exchange and self._device.type == "cuda") \
            else env.process_group
        self._is_weighted = module.is_weighted
        cfgs = module.embedding_bag_configs
        self._embedding_bag_configs = cfgs
        # tables of different pooling types (and dims, and placements) share ONE lookup per rank: pooling is
        # per-feature metadata of the kernels, not a reason for a second module + cat (the reference groups by
        # pooling / data type / compute kernel: embedding_sharding.py:393-490, embedding_lookup.py:219-253)
        # ---- global feature list, in the collection's output order -----------------------------
        self._feature_names: List[str] = []
        g_table: List[int] = []
        for t, c in enumerate(cfgs):
            for f in c.feature_names:
                self._feature_names.append(f)
                g_table.append(t)
        Fg = len(self._feature_names)
        g_dim = [cfgs[t].embedding_dim for t in g_table]
        self._lengths_per_embedding = g_dim
        self._D_total = sum(g_dim)
        # ---- who holds what --------------------------------------------------------------------
        kind: List[int] = []  # per table: -1 row-wise, else owning rank
        for c in cfgs:
            ps = table_name_to_parameter_sharding[c.name]
            if ps.sharding_type == ShardingType.DATA_PARALLEL.value:
                kind.append(-2)
            elif ps.sharding_type == ShardingType.ROW_WISE.value:
                kind.append(-1)
            elif ps.sharding_type == ShardingType.TABLE_WISE.value:
                kind.append(int(ps.ranks[0]))
            else:
                raise NotImplementedError(f"sharding type {ps.sharding_type} is outside the MI355X hot path "
                                          "(table_wise / row_wise)")
        self._table_kind = kind
        # local feature list of every rank: row-wise features first (same columns on every rank)
        rw_feats = [g for g in range(Fg) if kind[g_table[g]] == -1]
        self._dp_feats = [g for g in range(Fg) if kind[g_table[g]] == -2]
        self._sharded_feats = [g for g in range(Fg) if kind[g_table[g]] != -2]
        local_feats = [rw_feats + [g for g in range(Fg) if kind[g_table[g]] == r] for r in range(W)]
        self._local_feats = local_feats
        self._D_local_per_rank = [sum(g_dim[g] for g in lf) for lf in local_feats]
        self._D_local = self._D_local_per_rank[me]
        self._F_local = len(local_feats[me])
        self._send_feature_order = [g for lf in local_feats for g in lf]
        self._send_feats_per_rank = [len(lf) for lf in local_feats]
        # bucketized row-wise input dist: row-wise features (bucketized, one block per destination) and the table-wise
        # features in destination order travel as separate pieces of one exchange
        self._rw_feats = rw_feats
        self._tw_send_order = [g for r in range(W) for g in local_feats[r] if kind[g_table[g]] != -1]
        self._tw_per_rank = [sum(1 for g in local_feats[r] if kind[g_table[g]] != -1) for r in range(W)]
        self._rw_block_sizes = torch.tensor([rw_block_size(cfgs[g_table[g]].num_embeddings, W) for g in rw_feats],
                                            dtype=torch.int64, device=self._device if self._device.type != "meta" else "cpu")
        self._rw_mean = any(pooling_type_to_pooling_mode(cfgs[g_table[g]].pooling) == 1 for g in rw_feats)
        if self._rw_mean and rw_input_dist == "bucketize":
            raise NotImplementedError(
                "rw_input_dist='bucketize' with MEAN-pooled row-wise tables: a rank would divide its partial sum by the number "
                "of ids in ITS row block, not by the bag length; use 'windows' (or 'auto', which does) for such collections")
        # exchange descriptors (batch-independent part)
        feat_src, feat_slab_col = [0] * Fg, [0] * Fg
        for r in range(W):
            col = 0
            for g in local_feats[r]:
                if kind[g_table[g]] == -1:
                    feat_src[g], feat_slab_col[g] = -1, col
                elif kind[g_table[g]] == r:
                    feat_src[g], feat_slab_col[g] = r, col
                col += g_dim[g]
        for g in self._dp_feats:
            feat_src[g] = -2
        out_col = [0]
        for d in g_dim:
            out_col.append(out_col[-1] + d)
        self._out_col = out_col
        dev = self._device
        self._feat_out_col = torch.tensor(out_col, dtype=torch.int32, device=dev)
        self._feat_src = torch.tensor(feat_src, dtype=torch.int32, device=dev)
        self._feat_slab_col = torch.tensor(feat_slab_col, dtype=torch.int32, device=dev)
        self._slab_stride = torch.tensor(self._D_local_per_rank, dtype=torch.int32, device=dev)
        self._vec_ok = all(d % 4 == 0 for d in g_dim)
        self._layout_cache: Dict[int, Dict[str, Any]] = {}
        # state_dict() / fused-optimizer state as torch ShardedTensors whenever a process group exists (the reference's
        # behaviour); False hands out the plain local shards
        self.sharded_tensor_state = True
        self._kjt_cache: Dict[Tuple, Any] = {}
        self._output_buffer: Optional[torch.Tensor] = None  # see set_output_buffer
        self.half_batch_exchange = False  # see set_half_batch_exchange
        self._replicated_grad_sink: Optional[torch.Tensor] = None  # see set_replicated_grad_sink
        self._static_exchange: Optional[Dict[str, Any]] = None  # see set_graph_exchange
        self._weights_epoch = 0  # bumped by everything that rewrites tables outside a train step (see ExplicitLookupStep.epoch)
        # ---- local tables + TBE ----------------------------------------------------------------
        self._local_tables: List[_LocalTable] = []
        local_table_index: Dict[int, int] = {}
        for g in local_feats[me]:
            t = g_table[g]
            if t in local_table_index:
                continue
            c = cfgs[t]
            if kind[t] == -1:
                rows = rw_shard_rows(c.num_embeddings, W)[me]
                self._local_tables.append(_LocalTable(c, rows, me * rw_block_size(c.num_embeddings, W), True,
                                                      table_name_to_parameter_sharding[c.name].compute_kernel))
            else:
                self._local_tables.append(_LocalTable(c, c.num_embeddings, 0, False,
                                                      table_name_to_parameter_sharding[c.name].compute_kernel))
            local_table_index[t] = len(self._local_tables) - 1
        ftm_local = [local_table_index[g_table[g]] for g in local_feats[me]]
        # row-wise shards see un-bucketized GLOBAL ids: (first global row, global rows) per local feature
        win_first = [self._local_tables[i].row_offset for i in ftm_local]
        win_global = [self._local_tables[i].cfg.num_embeddings for i in ftm_local]
        self._has_rw = any(lt.row_wise for lt in self._local_tables)
        fused_params = dict(fused_params or {})
        factory = tbe_factory or _default_tbe_factory
        self._emb_module = None
        self._row_windows = None
        self._rw_mode_active = "windows"
        if self._local_tables:
            self._emb_module = factory(
                [(max(lt.local_rows, 0), lt.cfg.embedding_dim, lt.compute_kernel) for lt in self._local_tables],
                ftm_local * W, pooling_type_to_pooling_mode(self._local_tables[ftm_local[0]].cfg.pooling), dev, fused_params)
            if self._exchange:
                self._emb_module.set_a2a_output_layout(W)
            self._row_windows = (win_first * W, win_global * W) if self._has_rw else None
            self._rw_mode_active = "windows"
            if self._has_rw:
                self._emb_module.set_row_windows(*self._row_windows)
            local_pooling = [pooling_type_to_pooling_mode(self._local_tables[i].cfg.pooling) for i in ftm_local]
            if len(set(local_pooling)) > 1:
                self._emb_module.set_feature_pooling(local_pooling * W)
            self._init_parameters()
            self._optim = EmbeddingFusedOptimizer(self._emb_module, [lt.cfg.name for lt in self._local_tables],
                                                  key_prefix="embedding_bags.", wrap=self._wrap)
        else:
            self._optim = None
        # global-column addressing of the sharded features for the world_size == 1 "write into one buffer" path
        self._sharded_out_off = torch.tensor([out_col[g] for g in local_feats[me]], dtype=torch.int64, device=dev)
        # ---- data-parallel (replicated) tables -------------------------------------------------------
        self._dp_module = None
        self._dp_table_ids: List[int] = []
        if self._dp_feats:
            for g in self._dp_feats:
                if g_table[g] not in self._dp_table_ids:
                    self._dp_table_ids.append(g_table[g])
            dp_ftm = [self._dp_table_ids.index(g_table[g]) for g in self._dp_feats]
            dpf = dp_tbe_factory or _default_dp_tbe_factory
            dp_pooling = [pooling_type_to_pooling_mode(cfgs[g_table[g]].pooling) for g in self._dp_feats]
            self._dp_module = dpf([(cfgs[t].num_embeddings, cfgs[t].embedding_dim) for t in self._dp_table_ids], dp_ftm,
                                  dp_pooling[0], dev)
            if len(set(dp_pooling)) > 1:
                self._dp_module.set_feature_pooling(dp_pooling)
            self._dp_module._owned_by_sharded_module = True  # its weights load / save as embedding_bags.<t>.weight
            for t, w in zip(self._dp_table_ids, self._dp_module.split_embedding_weights()):
                w.uniform_(cfgs[t].get_weight_init_min(), cfgs[t].get_weight_init_max())
            self._dp_out_off = torch.tensor([out_col[g] for g in self._dp_feats], dtype=torch.int64, device=dev)

    def defer_backward_sort(self, on: bool = True) -> None:
        """Lets the caller choose when the fused lookup's backward sort starts (launch_deferred_backward_sort)."""
        if self._emb_module is not None:
            self._emb_module.defer_backward_sort = bool(on)

    def launch_deferred_backward_sort(self) -> bool:
        m = self._emb_module
        return bool(m is not None and hasattr(m, "launch_deferred_backward_sort") and m.launch_deferred_backward_sort())

    def set_output_buffer(self, buf: Optional[torch.Tensor]) -> None:
        """A persistent float32 buffer of B_local * sum(D) elements that receives the pooled output of every
        step with that batch size (instead of a fresh allocation) — e.g. the static input of a HIP-graph
        segment, so that the segment reads the embeddings in place.  The caller owns the aliasing: the
        output of step i is overwritten by step i + 1."""
        self._output_buffer = buf

    def set_replicated_grad_sink(self, buf: Optional[torch.Tensor]) -> None:
        """A persistent float32 buffer shaped like the replicated tables' parameter (`_dp_module.weights`): the explicit
        step's backward writes their dense gradient there and hands it out as `.grad` (DLRMTrain: the parameter's slice of
        the flat gradient buffer that is all-reduced as a whole).  Requires zero_grad(set_to_none=True) between steps."""
        if buf is not None and (self._dp_module is None or buf.shape != self._dp_module.weights.shape):
            raise ValueError("set_replicated_grad_sink: the buffer must be shaped like the replicated tables' parameter")
        self._replicated_grad_sink = buf

    def set_graph_exchange(self, batch_size: Optional[int]):
        """Static-exchange mode for compute_explicit() at this per-rank batch size (None = off): the pooled all-to-alls
        use PERSISTENT receive / send buffers and the step neither unpacks nor packs — the owner captures those two kernels
        into its own HIP graphs (the returned callables: `unpack()` writes the sharded features' columns of the output
        buffer from the receive buffer, `pack(grad)` fills the send buffer from the [B, sum D] gradient), so that they cost
        no eager launch in a stretch of the step where the host is what the GPU waits for.  One step at a time uses the
        buffers, which the explicit step guarantees (also with the next lookup prefetched: its all-to-all is ordered
        behind this step's unpack)."""
        if batch_size is None or not self._exchange:
            self._static_exchange = None
            return None
        B = int(batch_size)
        lay = self._exchange_layout(B)
        dev = self._device
        st = {"B": B, "recv_fwd": torch.zeros(lay["recv_numel"], dtype=torch.float32, device=dev),
              "send_bwd": torch.zeros(lay["recv_numel"], dtype=torch.float32, device=dev),
              "grad_recv": torch.zeros(lay["send_numel"], dtype=torch.float32, device=dev)}
        self._static_exchange = st
        scale = 1.0 / self._world_size if GRADIENT_DIVISION else 1.0

        def unpack() -> None:
            torch.ops.tbe_hip.pooled_exchange_unpack_into(
                st["recv_fwd"], lay["feat_out_col"], lay["feat_src"], lay["feat_slab_col"], lay["slab_offset"],
                lay["slab_stride"], B, self._D_total, self._vec_ok, 1.0, self._alias_output_buffer(B))

        def pack(grad: torch.Tensor) -> None:
            torch.ops.tbe_hip.pooled_exchange_pack_into(
                grad.view(B, self._D_total), lay["feat_out_col"], lay["feat_src"], lay["feat_slab_col"], lay["slab_offset"],
                lay["slab_stride"], self._vec_ok, scale, st["send_bwd"])

        return unpack, pack

    def set_half_batch_exchange(self, on: bool) -> None:
        """compute_explicit() then exchanges the pooled embeddings (and their gradients) as two half-batches
        (ExplicitLookupStep.finish_half / start_backward_half): the owner of the step interleaves the halves' dense work
        with the other half's exchange."""
        self.half_batch_exchange = bool(on)

    def _alias_output_buffer(self, B: int) -> torch.Tensor:
        """A fresh [B, sum D] tensor over the output buffer's storage that is NOT an autograd view of it
        (later lookups write their column blocks into it in place, which autograd forbids for views
        created inside a custom Function)."""
        buf = self._output_buffer
        return torch.empty(0, dtype=torch.float32, device=buf.device).set_(
            buf.untyped_storage(), buf.storage_offset(), (B, self._D_total), (self._D_total, 1))

    # ---- parameters -----------------------------------------------------------------------------
    def _init_parameters(self) -> None:
        # U(-sqrt(1/N), sqrt(1/N)) per table (batched_embedding_kernel.py:530-544)
        for lt, w in zip(self._local_tables, self._emb_module.split_embedding_weights()):
            if w.numel():
                w.uniform_(lt.cfg.get_weight_init_min(), lt.cfg.get_weight_init_max())

    def reset_parameters_sharding_invariant(self, seed: int = 0, chunk_rows: int = 1 << 16) -> None:
        """U(-sqrt(1/N), sqrt(1/N)) per table like _init_parameters, but as a function of (seed, table, global row)
        only: every table is drawn in chunks of `chunk_rows` global rows, chunk c of table t from its own generator
        seeded with (seed, t, c), and a shard copies the rows it holds.  Any sharding of the collection — and the
        unsharded one — then starts from the same tables, which is what lets a world-size-N run be compared with a
        world-size-1 run (the reference's tests copy a global model's state_dict into the shards instead:
        test_model_parallel_base.py:92-122)."""
        self._weights_epoch += 1
        gen = torch.Generator(device=self._device)
        index = {c.name: t for t, c in enumerate(self._embedding_bag_configs)}
        targets = [(n, w, r0) for n, (w, r0) in self.local_shards().items()] + [(n, w, 0) for n, w in self.dp_tables().items()]
        with torch.no_grad():
            for name, w, row0 in targets:
                cfg = self._embedding_bag_configs[index[name]]
                lo, hi = cfg.get_weight_init_min(), cfg.get_weight_init_max()
                n, D = w.shape
                c = row0 // chunk_rows
                while c * chunk_rows < row0 + n:
                    g0 = c * chunk_rows
                    rows = min(chunk_rows, cfg.num_embeddings - g0)
                    gen.manual_seed((int(seed) * 1000003 + index[name]) * 1000003 + c)
                    block = torch.rand((rows, D), generator=gen, device=self._device, dtype=torch.float32)
                    a, b = max(g0, row0), min(g0 + rows, row0 + n)
                    w[a - row0:b - row0].copy_(block[a - g0:b - g0].mul_(hi - lo).add_(lo))
                    c += 1

    @property
    def fused_optimizer(self) -> Optional[EmbeddingFusedOptimizer]:
        return self._optim

    @property
    def embedding_bag_configs(self) -> List[EmbeddingBagConfig]:
        return self._embedding_bag_configs

    def local_shards(self) -> Dict[str, Tuple[torch.Tensor, int]]:
        """table name -> (local weight shard [rows_local, D], first global row of the shard)."""
        if self._emb_module is None:
            return {}
        return {lt.cfg.name: (w, lt.row_offset)
                for lt, w in zip(self._local_tables, self._emb_module.split_embedding_weights())}

    def _table_shards(self, name: str, cols: Optional[int] = None) -> Tuple[List[int], List[Tuple[List[int], List[int], int]]]:
        """(global size, [(offsets, sizes, rank)]) of a sharded table's weight (cols = D) or row-wise state (cols None)."""
        t = next(i for i, c in enumerate(self._embedding_bag_configs) if c.name == name)
        cfg, kind, W = self._embedding_bag_configs[t], self._table_kind[t], self._world_size
        if kind == -1:
            rows, off, out = rw_shard_rows(cfg.num_embeddings, W), 0, []
            for r in range(W):
                out.append(([off, 0], [rows[r], cols], r) if cols is not None else ([off], [rows[r]], r))
                off += rows[r]
        else:
            out = [([0, 0], [cfg.num_embeddings, cols], kind) if cols is not None else ([0], [cfg.num_embeddings], kind)]
        return ([cfg.num_embeddings, cols] if cols is not None else [cfg.num_embeddings]), out

    def _wrap(self, name: str, local: torch.Tensor):
        if not self.sharded_tensor_state or self._pg is None:
            return local
        size, shards = self._table_shards(name, local.shape[1] if local.dim() == 2 else None)
        if local.dim() == 2 and local.shape[1] != size[1]:
            return local
        return wrap_sharded(local, size, shards, self._pg, self._rank, self._device)

    def state_dict(self, destination=None, prefix: str = "", keep_vars: bool = False):
        """`embedding_bags.<table>.weight` for EVERY table this rank holds (embeddingbag.py:405-416): the local
        shard [rows_local, D] of a sharded table, the whole [rows, D] of a replicated one.  The tensors alias
        the modules' storage (host views, cache written back, for MANAGED_CACHING tables)."""
        destination = {} if destination is None else destination
        for name, (w, _) in self.local_shards().items():
            # with a process group the value is a ShardedTensor over the shard, as the reference's (sharded_tensor_state)
            destination[f"{prefix}embedding_bags.{name}.weight"] = self._wrap(name, w if keep_vars else w.detach())
        for name, w in self.dp_tables().items():
            destination[f"{prefix}embedding_bags.{name}.weight"] = w if keep_vars else w.detach()
        return destination

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        """Accepts, per table, either the local shard or the WHOLE table (rows are then cut at this rank's row
        offset, as the reference's tests load a global model into shards: test_model_parallel_base.py:92-122)."""
        self._weights_epoch += 1  # a lookup prefetched before this load read the old tables
        targets = {n: (w, r0, False) for n, (w, r0) in self.local_shards().items()}
        targets.update({n: (w, 0, True) for n, w in self.dp_tables().items()})
        cfg = {c.name: c for c in self._embedding_bag_configs}
        for name, (w, row0, _) in targets.items():
            key = f"{prefix}embedding_bags.{name}.weight"
            if key not in state_dict:
                if strict:
                    missing_keys.append(key)
                continue
            src = unwrap_local(state_dict[key])
            if tuple(src.shape) == tuple(w.shape):
                pass
            elif src.dim() == 2 and src.shape[0] == cfg[name].num_embeddings and src.shape[1] == w.shape[1]:
                src = src[row0:row0 + w.shape[0]]
            else:
                error_msgs.append(f"size mismatch for {key}: {tuple(src.shape)} vs local {tuple(w.shape)} / global "
                                  f"({cfg[name].num_embeddings}, {w.shape[1]})")
                continue
            with torch.no_grad():
                w.copy_(src)
        # the replicated tables' dense module is a registered child: nn.Module.load_state_dict will visit it after this
        # method; hand it its own (just restored) storage under the key it expects, so that strict loading of a
        # reference-shaped checkpoint (embedding_bags.<table>.weight only) does not report it missing
        if self._dp_module is not None:
            for k, v in nn.Module.state_dict(self._dp_module).items():
                state_dict.setdefault(f"{prefix}_dp_module.{k}", v)
        if strict:
            mine = {f"{prefix}embedding_bags.{n}.weight" for n in cfg}
            for k in state_dict.keys():
                if k.startswith(prefix + "embedding_bags.") and k not in mine:
                    unexpected_keys.append(k)

    def dp_tables(self) -> Dict[str, torch.Tensor]:
        """table name -> replicated weight [rows, D] (a view of the dense TBE's parameter)."""
        if self._dp_module is None:
            return {}
        return {self._embedding_bag_configs[t].name: w
                for t, w in zip(self._dp_table_ids, self._dp_module.split_embedding_weights())}

    def named_parameters(self, prefix: str = "", recurse: bool = True) -> Iterator[Tuple[str, nn.Parameter]]:
        # sharded tables are fused (updated inside backward, batched_embedding_kernel.py:655-658);
        # replicated tables are ordinary dense parameters (all-reduced by DDP, stepped by the dense optimizer)
        if self._dp_module is not None:
            yield (prefix + ("." if prefix else "") + "_dp_module.weights", self._dp_module.weights)

    # ---- layouts --------------------------------------------------------------------------------
    def _exchange_layout(self, B: int) -> Dict[str, Any]:
        lay = self._layout_cache.get(B)
        if lay is None:
            W = self._world_size
            offs, o = [], 0
            for r in range(W):
                offs.append(o)
                o += B * self._D_local_per_rank[r]
            lay = {
                "slab_offset": torch.tensor(offs, dtype=torch.int64, device=self._device),
                "slab_stride": self._slab_stride,
                "feat_out_col": self._feat_out_col, "feat_src": self._feat_src, "feat_slab_col": self._feat_slab_col,
                "recv_splits": [B * d for d in self._D_local_per_rank], "recv_numel": o,
                "send_splits": [B * self._D_local] * W, "send_numel": W * B * self._D_local,
            }
            self._layout_cache[B] = lay
        return lay

    # ---- input dist -----------------------------------------------------------------------------
    def _send_perm(self, keys: List[str]) -> torch.Tensor:
        ck = ("perm", tuple(keys))
        hit = self._kjt_cache.get(ck)
        if hit is None:
            pos = {k: i for i, k in enumerate(keys)}
            missing = [n for n in self._feature_names if n not in pos]
            if missing:
                raise KeyError(f"KeyedJaggedTensor is missing features {missing[:3]}...")
            order = [pos[self._feature_names[g]] for g in self._send_feature_order]
            hit = (order, torch.tensor(order, dtype=torch.int32, device=self._device))
            self._kjt_cache[ck] = hit
        return hit

    def _dp_inputs(self, features: KeyedJaggedTensor):
        """ids of the replicated features for the LOCAL batch (no communication)."""
        if self._dp_module is None:
            return None
        keys = features.keys()
        ck = ("dp", tuple(keys))
        hit = self._kjt_cache.get(ck)
        if hit is None:
            pos = {k: i for i, k in enumerate(keys)}
            order = [pos[self._feature_names[g]] for g in self._dp_feats]
            hit = (order, torch.tensor(order, dtype=torch.int32, device=self._device))
            self._kjt_cache[ck] = hit
        order, order_t = hit
        B = features.stride()
        fixed = features.fixed_lengths()
        weights = features.weights_or_none() if self._is_weighted else None
        if fixed is not None and len(set(fixed)) == 1 and fixed[0] > 0:
            L = fixed[0]
            nkeys = len(keys)
            v = _gather_rows(features.values().view(nkeys, B * L), order, order_t).reshape(-1)
            w = _gather_rows(weights.view(nkeys, B * L), order, order_t).reshape(-1) if weights is not None else None
            ck2 = ("dpoff", B, L)
            offs = self._kjt_cache.get(ck2)
            if offs is None:
                offs = torch.arange(len(order) * B + 1, dtype=torch.int64, device=self._device) * L
                self._kjt_cache[ck2] = offs
            return v, offs, w
        sub = features.permute(order, None)
        return sub.values(), sub.offsets().long(), (sub.weights_or_none() if weights is not None else None)

    # ---- row-wise input dist: row windows (sync-free, W x the ids) or bucketized (the reference's) --------------------
    AUTO_WINDOWS_MAX_POOLING = 2  # "auto": host-known pooling factor <= this -> windows; longer / data-dependent -> bucketize

    def _pick_rw_mode(self, features: KeyedJaggedTensor) -> str:
        if not self._rw_feats or self._rw_input_dist == "windows" or self._rw_mean:
            return "windows"
        if self._rw_input_dist == "bucketize":
            return "bucketize"
        fixed = features.fixed_lengths()
        if fixed is not None and len(set(fixed)) == 1 and 0 < fixed[0] <= self.AUTO_WINDOWS_MAX_POOLING:
            return "windows"
        return "bucketize"

    def _set_rw_mode(self, mode: str) -> None:
        """windows: the lookup sees GLOBAL ids of row-wise features and masks with its shard's row window;
        bucketize: it sees the LOCAL rows of its own block only (no window)."""
        if mode == self._rw_mode_active:
            return
        if self._emb_module is not None and self._row_windows is not None:
            if mode == "windows":
                self._emb_module.set_row_windows(*self._row_windows)
            else:
                self._emb_module.set_row_windows(None)
        self._rw_mode_active = mode

    def _bucketize_perm(self, keys: List[str]):
        ck = ("bkt", tuple(keys))
        hit = self._kjt_cache.get(ck)
        if hit is None:
            pos = {k: i for i, k in enumerate(keys)}
            missing = [n for n in self._feature_names if n not in pos]
            if missing:
                raise KeyError(f"KeyedJaggedTensor is missing features {missing[:3]}...")
            rw_pos = [pos[self._feature_names[g]] for g in self._rw_feats]
            tw_pos = [pos[self._feature_names[g]] for g in self._tw_send_order]
            dev = self._device
            tw_first = [0]
            for n in self._tw_per_rank:
                tw_first.append(tw_first[-1] + n)
            hit = (rw_pos, torch.tensor(rw_pos, dtype=torch.int32, device=dev), tw_pos,
                   torch.tensor(tw_pos, dtype=torch.int32, device=dev), tw_first)
            self._kjt_cache[ck] = hit
        return hit

    def _input_dist_bucketized(self, features: KeyedJaggedTensor, dp_in) -> Awaitable[SparseFeaturesDist]:
        """The reference's row-wise input dist (bucketize_kjt_before_all2all + KJTAllToAll: embedding_sharding.py:121-184,
        dist_data.py:137-524) folded into this collection's ONE exchange: the piece for destination r is
        [bucket r of the row-wise features | the table-wise features r owns], lengths first (static sizes), then ONE D2H
        read of the piece / receive counts (the reference reads them in two places: dist_data.py:396-398,
        jagged_tensor.py:502-509), then the ids (and per-sample weights).  The receiver's layout is the usual
        [src rank][local feature][sample]; row-wise ids arrive as LOCAL rows of this rank's block."""
        W, B = self._world_size, features.stride()
        rw_pos, rw_pos_t, tw_pos, tw_pos_t, tw_first = self._bucketize_perm(features.keys())
        n_rw = len(rw_pos)
        weighted = self._is_weighted and features.weights_or_none() is not None
        rw = features.permute(rw_pos, rw_pos_t)
        with label("## bucketize_kjt_before_all2all ##"):
            blocks = self._rw_block_sizes.to(device=rw.values().device, dtype=rw.values().dtype)
            bl, bi, bw, _, _ = torch.ops.fbgemm.block_bucketize_sparse_features(
                lengths=rw.lengths(), indices=rw.values(), bucketize_pos=False, sequence=False, block_sizes=blocks,
                my_size=W, weights=rw.weights_or_none() if weighted else None)
        tw = features.permute(tw_pos, tw_pos_t) if tw_pos else None
        tw_len = tw.lengths() if tw is not None else None
        pieces = []
        for r in range(W):
            pieces.append(bl[r * n_rw * B:(r + 1) * n_rw * B])
            if tw is not None and self._tw_per_rank[r]:
                pieces.append(tw_len[tw_first[r] * B:tw_first[r + 1] * B].to(bl.dtype))
        send_l = torch.cat(pieces)
        # piece boundaries inside the bucketized / table-wise id arrays, as device prefix sums
        off_b = torch.ops.fbgemm.asynchronous_complete_cumsum(bl)[::n_rw * B].to(torch.int64)  # [W + 1]
        if tw is not None:
            ck = ("twb", B)
            idx_t = self._kjt_cache.get(ck)
            if idx_t is None:
                idx_t = torch.tensor([f * B for f in tw_first], dtype=torch.int64, device=self._device)
                self._kjt_cache[ck] = idx_t
            off_t = tw.offsets().to(torch.int64).index_select(0, idx_t)  # [W + 1]
        else:
            off_t = torch.zeros(W + 1, dtype=torch.int64, device=off_b.device)
        len_in = [(n_rw + n) * B for n in self._tw_per_rank]
        len_out = [self._F_local * B] * W
        if self._exchange:
            recv_l = torch.empty(sum(len_out), dtype=send_l.dtype, device=send_l.device)
            with label("## all2all_data:lengths ##"):  # dist_data.py:366
                dist.all_to_all_single(recv_l, send_l, len_out, len_in, group=self._pg)
        else:
            recv_l = send_l
        with label("## all2all_data:split length for a2a ##"):  # dist_data.py:388-398: the D2H read
            host = torch.cat([off_b, off_t, recv_l.view(W, -1).sum(dim=1).to(torch.int64)]).cpu().tolist()
        ob, ot, val_out = host[:W + 1], host[W + 1:2 * W + 2], host[2 * W + 2:]
        val_in = [(ob[r + 1] - ob[r]) + (ot[r + 1] - ot[r]) for r in range(W)]

        def assemble(b_arr, t_arr):
            parts = []
            for r in range(W):
                parts.append(b_arr[ob[r]:ob[r + 1]])
                if t_arr is not None and ot[r + 1] > ot[r]:
                    parts.append(t_arr[ot[r]:ot[r + 1]])
            return torch.cat(parts) if parts else b_arr[:0]

        send_v = assemble(bi, tw.values() if tw is not None else None)
        send_w = assemble(bw, tw.weights() if tw is not None else None) if weighted else None
        if self._exchange:
            recv_v = torch.empty(sum(val_out), dtype=send_v.dtype, device=send_v.device)
            with label("## all2all_data:indices ##"):  # dist_data.py:190
                wk = dist.all_to_all_single(recv_v, send_v, val_out, val_in, group=self._pg, async_op=True)
            recv_w, wk2 = None, None
            if send_w is not None:
                recv_w = torch.empty(sum(val_out), dtype=send_w.dtype, device=send_w.device)
                with label("## all2all_data:weights ##"):  # dist_data.py:213
                    wk2 = dist.all_to_all_single(recv_w, send_w, val_out, val_in, group=self._pg, async_op=True)
        else:
            recv_v, recv_w, wk, wk2 = send_v, send_w, None, None

        def finish() -> SparseFeaturesDist:
            if wk is not None:
                wk.wait()
            if wk2 is not None:
                wk2.wait()
            offsets = torch.ops.fbgemm.asynchronous_complete_cumsum(recv_l).long()
            return SparseFeaturesDist(recv_v, offsets, recv_w, B, dp_in)

        return _InputDistAwaitable(finish)

    def input_dist(self, features: KeyedJaggedTensor) -> Awaitable[SparseFeaturesDist]:
        W, B = self._world_size, features.stride()
        dp_in = self._dp_inputs(features)
        mode = self._pick_rw_mode(features)
        self._set_rw_mode(mode)
        if mode == "bucketize":
            return self._input_dist_bucketized(features, dp_in)
        order, order_t = self._send_perm(features.keys())
        fixed = features.fixed_lengths()
        weights = features.weights_or_none() if self._is_weighted else None
        if fixed is not None and len(set(fixed)) == 1 and fixed[0] > 0:
            L = fixed[0]
            nkeys = len(features.keys())
            if order == list(range(nkeys)):  # already in send order: no gather
                send_v = features.values().view(nkeys, B * L)
                send_w = weights.view(nkeys, B * L) if weights is not None else None
            else:
                send_v = _gather_rows(features.values().view(nkeys, B * L), order, order_t)
                send_w = _gather_rows(weights.view(nkeys, B * L), order, order_t) if weights is not None else None
            in_splits = [n * B * L for n in self._send_feats_per_rank]
            out_splits = [self._F_local * B * L] * W
            if self._exchange:
                recv_v = torch.empty(sum(out_splits), dtype=send_v.dtype, device=send_v.device)
                with label("## all2all_data:indices ##"):  # dist_data.py:190
                    wk = dist.all_to_all_single(recv_v, send_v.view(-1), out_splits, in_splits, group=self._pg, async_op=True)
                recv_w, wk2 = None, None
                if send_w is not None:
                    recv_w = torch.empty(sum(out_splits), dtype=send_w.dtype, device=send_w.device)
                    with label("## all2all_data:weights ##"):  # dist_data.py:213
                        wk2 = dist.all_to_all_single(recv_w, send_w.view(-1), out_splits, in_splits, group=self._pg,
                                                     async_op=True)
            else:
                recv_v, recv_w, wk, wk2 = send_v.reshape(-1), (send_w.reshape(-1) if send_w is not None else None), None, None

            def finish() -> SparseFeaturesDist:
                if wk is not None:
                    wk.wait()
                if wk2 is not None:
                    wk2.wait()
                ck = ("off", B, L)
                offsets = self._kjt_cache.get(ck)
                if offsets is None:
                    offsets = torch.arange(W * self._F_local * B + 1, dtype=torch.int64, device=self._device) * L
                    self._kjt_cache[ck] = offsets
                return SparseFeaturesDist(recv_v, offsets, recv_w, B, dp_in)

            return _InputDistAwaitable(finish)
        # ---- data-dependent pooling factors: lengths a2a, D2H of the value counts, values a2a
        #      (the reference's 2-phase KJTAllToAll, dist_data.py:137-524) -----------------------
        sent = features.permute(order, None)
        lengths = sent.lengths()
        lpk = sent.length_per_key()
        n_per_rank = self._send_feats_per_rank
        len_in = [n * B for n in n_per_rank]
        len_out = [self._F_local * B] * W
        val_in, k = [], 0
        for n in n_per_rank:
            val_in.append(sum(lpk[k:k + n]))
            k += n
        if self._exchange:
            recv_l = torch.empty(sum(len_out), dtype=lengths.dtype, device=lengths.device)
            with label("## all2all_data:lengths ##"):  # dist_data.py:366
                dist.all_to_all_single(recv_l, lengths, len_out, len_in, group=self._pg)
            with label("## all2all_data:split length for a2a ##"):  # dist_data.py:388
                val_out = recv_l.view(W, -1).sum(dim=1).cpu().tolist()  # host sync (dist_data.py:396-398)
            recv_v = torch.empty(sum(val_out), dtype=sent.values().dtype, device=lengths.device)
            with label("## all2all_data:indices ##"):
                wk = dist.all_to_all_single(recv_v, sent.values(), val_out, val_in, group=self._pg, async_op=True)
            recv_w, wk2 = None, None
            if weights is not None:
                recv_w = torch.empty(sum(val_out), dtype=weights.dtype, device=lengths.device)
                with label("## all2all_data:weights ##"):
                    wk2 = dist.all_to_all_single(recv_w, sent.weights(), val_out, val_in, group=self._pg, async_op=True)
        else:
            recv_l, recv_v, recv_w, wk, wk2 = lengths, sent.values(), sent.weights_or_none() if weights is not None else None, None, None

        def finish_var() -> SparseFeaturesDist:
            if wk is not None:
                wk.wait()
            if wk2 is not None:
                wk2.wait()
            offsets = torch.ops.fbgemm.asynchronous_complete_cumsum(recv_l).long()
            return SparseFeaturesDist(recv_v, offsets, recv_w, B, dp_in)

        return _InputDistAwaitable(finish_var)

    # ---- compute + output dist ------------------------------------------------------------------
    def _dp_fill(self, out: torch.Tensor, dist_input: SparseFeaturesDist) -> torch.Tensor:
        if self._dp_module is None:
            return out
        v, offs, w = dist_input.dp
        return self._dp_module.forward_into(out, self._dp_out_off, self._D_total, v, offs, w)

    def compute_and_output_dist(self, dist_input: SparseFeaturesDist) -> Awaitable[KeyedTensor]:
        B = dist_input.batch_size
        keys, lpe = self._feature_names, self._lengths_per_embedding
        if not self._exchange:
            buf = self._output_buffer
            use_buf = buf is not None and buf.numel() == B * self._D_total
            if self._dp_module is None and not use_buf:
                emb = self._emb_module(dist_input.values, dist_input.offsets, dist_input.weights)
                return NoWait(KeyedTensor(keys, lpe, emb))
            # both lookups write their column blocks of ONE [B, sum D] matrix
            out = (self._alias_output_buffer(B) if use_buf
                   else torch.empty((B, self._D_total), dtype=torch.float32, device=self._device))
            if self._emb_module is not None:
                out = self._emb_module.forward_into(out, self._sharded_out_off, self._D_total, dist_input.values,
                                                    dist_input.offsets, dist_input.weights)
            return NoWait(KeyedTensor(keys, lpe, self._dp_fill(out, dist_input)))
        if self._emb_module is not None:
            with label("## tbe_lookup ##"):
                emb = self._emb_module(dist_input.values, dist_input.offsets, dist_input.weights)
        else:
            emb = torch.zeros((self._world_size * B, 0), dtype=torch.float32, device=self._device,
                              requires_grad=True)
        state = _ExchangeState(self, B)
        recv = _ExchangeReq.apply(emb, state)
        return _OutputAwaitable(
            lambda: KeyedTensor(keys, lpe, self._dp_fill(_ExchangeWait.apply(recv, state), dist_input)))

    def explicit_step_supported(self, batch_size: int) -> bool:
        """Whether compute_explicit() can serve a batch of this (per-rank) size: a fused module and an output buffer of
        exactly that batch — the configuration of the HIP-graph train step (with or without the exchange)."""
        buf = self._output_buffer
        return (self._emb_module is not None and buf is not None
                and buf.numel() == batch_size * self._D_total and hasattr(self._emb_module, "lookup_no_autograd")
                and (self._dp_module is None or hasattr(self._dp_module, "lookup_no_autograd")))

    def compute_explicit(self, dist_input: SparseFeaturesDist, prefetched: bool = False) -> "ExplicitLookupStep":
        """compute_and_output_dist for a caller that runs the backward ITSELF (no autograd nodes): lookup + start of the
        pooled all-to-all now, `finish()` = wait + unpack (+ replicated tables) into the output buffer,
        `start_backward(grad)` = pack + gradient all-to-all (+ the replicated tables' backward), `finish_backward()` =
        the fused backward.  Check explicit_step_supported() first.  `prefetched`: the call is made at the END of the previous
        step, before its dense optimizer — everything that reads dense parameters (the replicated tables) waits for finish()."""
        if not self.explicit_step_supported(dist_input.batch_size):
            raise RuntimeError("compute_explicit: not available for this configuration (explicit_step_supported())")
        return ExplicitLookupStep(self, dist_input, halves=self.half_batch_exchange, early_replicated_lookup=not prefetched)

    def forward(self, features: KeyedJaggedTensor) -> Awaitable[KeyedTensor]:
        return self.compute_and_output_dist(self.input_dist(features).wait())


class ExplicitLookupStep:
    """One step of a sharded collection driven without autograd (ShardedEmbeddingBagCollection.compute_explicit).
    With the exchange: lookup in all-to-all layout -> pooled all-to-all -> unpack into the output buffer; without
    (one rank): the lookup writes its column blocks of the output buffer directly."""

    def __init__(self, owner: "ShardedEmbeddingBagCollection", dist_input: SparseFeaturesDist, halves: bool = False,
                 early_replicated_lookup: bool = True) -> None:
        self.o, self.d = owner, dist_input
        self.halves = bool(halves and owner._exchange)  # two half-batch exchanges (finish_half / start_backward_half)
        self.epoch = owner._weights_epoch  # a prefetched step is only good while nobody has rewritten the tables
        self.dp_rec = None
        self.state: Optional[_ExchangeState] = None
        self._grad: Optional[torch.Tensor] = None
        self._early_dp = False
        self._dp_sort_pending = False
        if owner._exchange:
            # order of the HOST calls: lookup kernel, pooled all-to-all, THEN the backward's side-stream sort (6 launches,
            # ~60 us of host time): the all-to-all is on the step's critical path, the sort is not
            m = owner._emb_module
            can_defer = hasattr(m, "launch_deferred_backward_sort") and not getattr(m, "defer_backward_sort", False)
            if can_defer:
                m.defer_backward_sort = True
            try:
                with label("## tbe_lookup ##"):
                    emb, self.rec = m.lookup_no_autograd(dist_input.values, dist_input.offsets, dist_input.weights)
            finally:
                if can_defer:
                    m.defer_backward_sort = False
            self.state = _ExchangeState(owner, dist_input.batch_size)
            if self.halves:
                self.state.start_forward_halves(emb)
            else:
                self.state.start_forward(emb, allow_static=True)
            # the replicated tables' lookup does not depend on the exchange: it fills its columns of the destination
            # while the all-to-all (on its own hardware queue) is in flight, instead of after the wait
            # (NOT when this step is prefetched at the end of the previous one: the replicated tables are dense parameters,
            # the previous step's dense optimizer has not run yet — finish() does the lookup then)
            self._early_dp = owner._dp_module is not None and early_replicated_lookup
            if self._early_dp:
                v, offs, w = dist_input.dp
                with label("## tbe_lookup ##"):
                    _, self.dp_rec = owner._dp_module.lookup_no_autograd(
                        v, offs, w, into=(self.state.output_destination(), owner._dp_out_off, owner._D_total))
            if can_defer:
                m.launch_deferred_backward_sort()
            self._out = None
        else:
            out = owner._alias_output_buffer(dist_input.batch_size)
            with label("## tbe_lookup ##"):
                self._out, self.rec = owner._emb_module.lookup_no_autograd(
                    dist_input.values, dist_input.offsets, dist_input.weights,
                    into=(out, owner._sharded_out_off, owner._D_total))

    def finish(self) -> torch.Tensor:
        """[B_local, sum D] pooled embeddings in the collection's key order, inside the output buffer."""
        self._late_dp_lookup()
        if self.state is not None and self.state.static is not None:
            self.state.wait_forward()  # the owner's forward graph unpacks the persistent receive buffer
            out = self.state.output_destination()
        else:
            out = self.state.finish_forward() if self.state is not None else self._out
        self._launch_late_dp_sort()
        return out

    def _late_dp_lookup(self) -> None:
        """The replicated tables' lookup of a step that could not do it early: ahead of the wait for the exchange."""
        o = self.o
        if o._dp_module is not None and not self._early_dp and self.dp_rec is None:
            v, offs, w = self.d.dp
            dest = self.state.output_destination() if self.state is not None else self._out
            # its backward's side-stream sort (5 launches of host time) goes behind the unpack launch: the host is what the
            # GPU waits for at this point of the step
            m = o._dp_module
            can_defer = hasattr(m, "launch_deferred_backward_sort") and not getattr(m, "defer_backward_sort", False)
            if can_defer:
                m.defer_backward_sort = True
            try:
                with label("## tbe_lookup ##"):
                    _, self.dp_rec = m.lookup_no_autograd(v, offs, w, into=(dest, o._dp_out_off, o._D_total))
            finally:
                if can_defer:
                    m.defer_backward_sort = False
            self._dp_sort_pending = can_defer

    def _launch_late_dp_sort(self) -> None:
        if self._dp_sort_pending:
            self._dp_sort_pending = False
            self.o._dp_module.launch_deferred_backward_sort()

    def finish_half(self, h: int) -> torch.Tensor:
        """Half-batch mode: rows [h B/2, (h + 1) B/2) of the pooled output, complete (the replicated tables' columns were
        filled for the whole batch while the exchange was in flight)."""
        self._late_dp_lookup()
        rows = self.state.finish_forward_half(h)
        self._launch_late_dp_sort()
        return rows

    def start_backward_half(self, h: int, grad_rows: torch.Tensor, grad_out: Optional[torch.Tensor] = None) -> None:
        """Half-batch mode: packs and starts the gradient all-to-all of half h.  With the LAST half pass `grad_out`, the
        whole [B_local, sum D] gradient (both halves written): the replicated tables' backward runs on it meanwhile."""
        self.state.start_backward_half(h, grad_rows)
        if grad_out is not None:
            self._dp_backward(grad_out)

    def start_backward(self, grad_out: torch.Tensor) -> None:
        """grad_out: [B_local, sum D], contiguous.  Packs and starts the gradient all-to-all (if any), and runs the
        replicated tables' backward (their dense gradient lands in `.grad` of the module's weights) meanwhile."""
        if self.state is not None and self.state.static is not None:
            self.state.start_backward_packed()  # the owner's backward graph has filled the persistent send buffer
        elif self.state is not None:
            self.state.start_backward(grad_out)
        else:
            self._grad = grad_out
        self._dp_backward(grad_out)

    def _dp_backward(self, grad_out: torch.Tensor) -> None:
        o = self.o
        if self.dp_rec is not None:
            w = o._dp_module.weights
            sink = o._replicated_grad_sink
            if sink is not None and w.grad is None:
                # the owner's persistent gradient buffer for the replicated tables (its slice of a flat, all-reduced
                # buffer): written in place — no allocation, no address-table kernel, no copy into the flat buffer later
                w.grad = o._dp_module.backward_no_autograd(self.dp_rec, grad_out, into=sink)
            else:
                g = o._dp_module.backward_no_autograd(self.dp_rec, grad_out)
                if w.grad is None:
                    w.grad = g
                else:
                    w.grad.add_(g)
            self.dp_rec = None

    def discard(self) -> None:
        """Drops a step that will not be used (a prefetched lookup made stale by a load_state_dict): drains its exchange."""
        st = self.state
        if st is not None:
            for w in [st.work] + list(getattr(st, "_half_work", [])):
                if w is not None:
                    w.wait()
            st.work = None
        self.rec = self.dp_rec = None

    def finish_backward(self) -> None:
        grad = self.state.finish_backward() if self.state is not None else self._grad
        self._grad = None
        with label("## tbe_backward_fused_optimizer ##"):
            self.o._emb_module.backward_no_autograd(self.rec, grad)


class EmbeddingBagCollectionSharder:
    """Builds the sharded module from an EmbeddingBagCollection + per-table plan
    (torchrec/distributed/embeddingbag.py:489-515)."""

    def __init__(self, fused_params: Optional[Dict[str, Any]] = None, tbe_factory: Optional[Callable] = None,
                 dp_tbe_factory: Optional[Callable] = None, rw_input_dist: Optional[str] = None) -> None:
        self.fused_params = fused_params
        self.tbe_factory = tbe_factory
        self.dp_tbe_factory = dp_tbe_factory
        self.rw_input_dist = rw_input_dist  # "auto" | "windows" | "bucketize" (ShardedEmbeddingBagCollection)

    def shard(self, module: EmbeddingBagCollection, params: Dict[str, ParameterSharding], env: ShardingEnv,
              device: Optional[torch.device] = None) -> ShardedEmbeddingBagCollection:
        return ShardedEmbeddingBagCollection(module, params, env, self.fused_params, device, self.tbe_factory,
                                             self.dp_tbe_factory, self.rw_input_dist)

    @property
    def module_type(self):
        return EmbeddingBagCollection
