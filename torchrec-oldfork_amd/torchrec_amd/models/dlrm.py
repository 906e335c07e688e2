"""DLRM dense side + model wrapper (torchrec/models/dlrm.py:36-406,
examples/dlrm/modules/dlrm_train.py).  Module and parameter names follow the reference so
state_dict keys are interchangeable (`dense_arch.model._mlp.0._linear.weight`,
`over_arch.model.1.bias`, ...).  The dense MLPs and the dot interaction are the only MFMA
users of the path (rocBLAS/hipBLASLt fp32 GEMMs)."""
import os
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from ..modules.mlp import MLP, LinearOut
from ..profiling import label
from ..sparse.jagged_tensor import KeyedJaggedTensor, KeyedTensor


class SparseArch(nn.Module):
    """models/dlrm.py:36-113: pooled embeddings -> [B, F, D]."""

    def __init__(self, embedding_bag_collection: nn.Module) -> None:
        super().__init__()
        self.embedding_bag_collection = embedding_bag_collection
        cfgs = embedding_bag_collection.embedding_bag_configs
        assert cfgs, "Embedding bag collection cannot be empty!"
        self.D: int = cfgs[0].embedding_dim
        self._sparse_feature_names: List[str] = [n for c in cfgs for n in c.feature_names]
        self.F: int = len(self._sparse_feature_names)

    def start(self, features: KeyedJaggedTensor):
        """Issues the lookup (and, when sharded, the pooled all-to-all) and returns without waiting."""
        return self.embedding_bag_collection(features)

    def finish(self, pending) -> torch.Tensor:
        sparse_features = pending.wait() if hasattr(pending, "wait") else pending
        B = sparse_features.values().shape[0]
        if sparse_features.keys() == self._sparse_feature_names:
            return sparse_features.values().reshape(B, self.F, self.D)
        sparse: Dict[str, torch.Tensor] = sparse_features.to_dict()
        return torch.cat([sparse[n] for n in self._sparse_feature_names], dim=1).reshape(B, self.F, self.D)

    def forward(self, features: KeyedJaggedTensor) -> torch.Tensor:
        return self.finish(self.start(features))

    @property
    def sparse_feature_names(self) -> List[str]:
        return self._sparse_feature_names


# Where the embedding backward's side-stream sort starts: "lookup" (default) = right behind the lookup kernel, beside the
# bottom MLP (measured best at N = 1: 7.62 M samples/s); "head" = behind the interaction forward, beside the over-arch
# GEMMs (7.58 M: the GEMMs fill every CU, the sort's launches trickle in).  DESIGN.md §3.
_SORT_PLACEMENT = os.environ.get("TORCHREC_AMD_SORT_PLACEMENT", "lookup")


# HIP-graph + flat-gradient mode: run forward and backward of a step by hand instead of through the autograd engine
# (DLRMTrain._explicit_step); TORCHREC_AMD_EXPLICIT_STEP=0 keeps the autograd path.
_EXPLICIT_STEP = os.environ.get("TORCHREC_AMD_EXPLICIT_STEP", "1") != "0"
# flat mode: capture the head segment's weight-gradient GEMMs into a second backward graph (0: one graph as before)
_DEFER_WGRAD = os.environ.get("TORCHREC_AMD_DEFER_WGRAD", "1") != "0"
# Two half-batches per step when the pooled embeddings cross links (DLRMTrain.capture_hip_graphs(half_batches=)):
# "auto" = on for per-rank batches of at least this many samples — N = 2 at the global batch of 65 536, where the exchange
# takes ~1.5 ms per pass over the one link pair and only half of the forward one stays exposed.  Measured with emulated
# link times: 7.01 -> 6.30 ms per step at 32 768 per rank; a tie at 16 384 (3.40 / 3.38) and a loss at 8192
# (1.86 -> 2.01): two half-size passes cost 0.2 - 0.3 ms more kernel time (GEMMs at half M, kernels at their latency floor).
# "1" / "0" force it.
# Flat-gradient graph mode under a pipeline that prefetches the next lookup: the weight / bias gradients of the head's FIRST
# this-many Linear layers are captured into a third graph that the explicit step replays AFTER it has started the NEXT step's
# lookup + pooled all-to-all (the rest stays behind this step's gradient all-to-all).  The forward all-to-all then has the
# same kind of cover the gradient all-to-all always had, instead of the bottom MLP's forward only.  0 = off.
# "auto" (default): 2 layers at per-rank batches of 16 384 .. 32 767 (N = 4 at the global batch of 65 536), none otherwise.
# Why not everywhere: the late layers' gradients are all-reduced at the very end of the step, in front of the dense
# optimizer and the next bottom MLP, and that all-reduce takes the same time at every batch size while the GEMMs that are
# supposed to hide the all-to-all shrink with it.  With emulated link AND all-reduce times 0 / 2 / 3 late
# layers measure 2.08 / 2.32 / 2.33 ms at 8192 per rank, 3.71 / 3.50 / 3.60 at 16 384; without the all-reduce time
# 1.87 / 1.71 / 1.72 and 3.38 / 3.18 / 3.11.  A number forces it.
_WGRAD_LATE_LAYERS = os.environ.get("TORCHREC_AMD_WGRAD_LATE_LAYERS", "auto")


def _late_layers(batch: int, halves: bool) -> int:
    if _WGRAD_LATE_LAYERS != "auto":
        return int(_WGRAD_LATE_LAYERS)
    return 2 if (not halves and 16384 <= batch < 32768) else 0


# Where the split-off graph of those layers runs: "late" (default) = behind the next step's prefetched lookup, as above;
# "early" = FIRST in the backward window, and every piece of the flat gradient is all-reduced as soon as it exists, the
# largest first (replicated tables, the split-off head layers, the other head layers; the bottom MLP's small slice last).
# Opt-in: bit-identical, four collectives per step instead of two, and with emulated link + all-reduce times within noise of
# no split at all (2.02 vs 2.05 ms at 8192 per rank; 3.71 vs 3.49 for "late" at 16 384).
_WGRAD_SPLIT_MODE = os.environ.get("TORCHREC_AMD_WGRAD_SPLIT_MODE", "late")


# whole-batch explicit step with an exchange: unpack / pack captured into the head segment's graphs instead of two eager
# launches (persistent receive / send buffers).  Opt-in: bit-identical, the two launch gaps (8 + 7 us) do disappear, and the
# step gets no faster — 1.732 vs 1.695 ms at 8192 per rank, 2.849 vs 2.854 at 16 384 (the host, freed earlier, starts the
# next input dist earlier, and its kernels then share the chip with the head's forward GEMMs: + 50 us of GEMM time)
_GRAPH_EXCHANGE = os.environ.get("TORCHREC_AMD_GRAPH_EXCHANGE", "0") == "1"
_HALF_BATCHES = os.environ.get("TORCHREC_AMD_HALF_BATCHES", "auto")
_HALF_BATCH_MIN = int(os.environ.get("TORCHREC_AMD_HALF_BATCH_MIN", "32768"))


def _pad64(n: int) -> int:
    return (n + 63) // 64 * 64


def _sort_hooks(ebc: nn.Module):
    """(defer, launch) callables of a collection that lets its caller place the backward sort, or (None, None)."""
    if _SORT_PLACEMENT != "head":
        return None, None
    inner = getattr(ebc, "sharded", ebc)  # (a wrapper object that carries the collection as `.sharded`, if any)
    if hasattr(inner, "defer_backward_sort") and hasattr(inner, "launch_deferred_backward_sort"):
        return inner.defer_backward_sort, inner.launch_deferred_backward_sort
    return None, None


class DenseArch(nn.Module):
    def __init__(self, in_features: int, layer_sizes: List[int], device: Optional[torch.device] = None) -> None:
        super().__init__()
        self.model = MLP(in_features, layer_sizes, bias=True, activation="relu", device=device)

    def forward(self, features: torch.Tensor) -> torch.Tensor:
        return self.model(features)


class _FusedDotInteraction(torch.autograd.Function):
    """One HIP kernel each way (csrc/dlrm_interaction.hip, fp32 MFMA) instead of the reference's
    cat + bmm + index + cat chain and its autograd backward (models/dlrm.py:206-219).

    `pad_rows`: the result [B, D + P] is a view of a [B, S] buffer with S = D + P rounded up to a multiple of 4 (479 ->
    480 floats): rows start 16-B aligned, so the kernel stores 16 B per lane and the GEMMs of the next layer read aligned
    rows (lda = S, K = D + P; tools/gemm_probe.py).  Values and shapes are those of the dense result."""

    @staticmethod
    def forward(ctx, dense, sparse, pad_rows=False, grad_sinks=None):
        from fbgemm_gpu import _lib
        from fbgemm_gpu._lib import check, ptr, stream_ptr

        dense, sparse = dense.contiguous(), sparse.contiguous()
        B, F, D = sparse.shape
        if dense.shape != (B, D) or sparse.device != dense.device:
            raise RuntimeError(f"dot interaction: dense {tuple(dense.shape)} does not match sparse {tuple(sparse.shape)} "
                               "(same batch, same embedding dim, same device required)")
        width = D + (F + 1) * F // 2
        stride = (D + ((F + 1) * F // 2 + 3) // 4 * 4) if pad_rows else width
        buf = torch.empty((B, stride), dtype=torch.float32, device=dense.device)
        with torch.cuda.device(dense.device):
            check(_lib.load().tbe_dlrm_interaction_forward_f32(ptr(dense), ptr(sparse), B, F, D, ptr(buf), stride,
                                                               stream_ptr(dense.device)),
                  "tbe_dlrm_interaction_forward_f32")
        ctx.save_for_backward(dense, sparse)
        # (d dense, d sparse) buffers of the caller instead of fresh allocations: a captured backward then writes e.g. its
        # half of a whole-batch gradient buffer in place (DLRMTrain.capture_hip_graphs(half_batches=True))
        ctx.grad_sinks = grad_sinks
        return buf if stride == width else buf[:, :width]

    @staticmethod
    def backward(ctx, grad_out):
        from fbgemm_gpu import _lib
        from fbgemm_gpu._lib import check, ptr, stream_ptr

        dense, sparse = ctx.saved_tensors
        B, F, D = sparse.shape
        width = D + (F + 1) * F // 2
        if grad_out.shape != (B, width):
            raise RuntimeError(f"dot interaction backward: grad_out {tuple(grad_out.shape)} has the wrong shape")
        if grad_out.dtype != torch.float32 or grad_out.stride(1) != 1 or grad_out.stride(0) < width:
            grad_out = grad_out.float().contiguous()  # padded rows (stride >= width) are read in place
        if ctx.grad_sinks is not None:
            gd, gs = ctx.grad_sinks
            if (gd.shape != dense.shape or gs.shape != sparse.shape or not gd.is_contiguous() or not gs.is_contiguous()
                    or gd.dtype != torch.float32 or gs.dtype != torch.float32 or gd.device != dense.device):
                raise RuntimeError("dot interaction backward: gradient sinks do not match the inputs")
        else:
            gd, gs = torch.empty_like(dense), torch.empty_like(sparse)
        with torch.cuda.device(dense.device):
            check(_lib.load().tbe_dlrm_interaction_backward_f32(ptr(dense), ptr(sparse), ptr(grad_out), grad_out.stride(0),
                                                                B, F, D, ptr(gd), ptr(gs), stream_ptr(dense.device)),
                  "tbe_dlrm_interaction_backward_f32")
        return gd, gs, None, None


def _fused_interaction_ok(dense: torch.Tensor, sparse: torch.Tensor) -> bool:
    F, D = sparse.shape[1], sparse.shape[2]
    return (dense.is_cuda and dense.dtype == torch.float32 and sparse.dtype == torch.float32
            and 1 <= F <= 27 and D in (16, 32, 64, 128))


class InteractionArch(nn.Module):
    """models/dlrm.py:193-219: [dense | upper-triangle of (dense,sparse)x(dense,sparse)^T]."""

    def __init__(self, num_sparse_features: int) -> None:
        super().__init__()
        self.F = num_sparse_features
        self.fused = True
        # 16-B aligned output rows (a strided [B, D + P] view of a [B, 480] buffer at F = 26, D = 128); the first
        # over-arch layer keeps the alignment for its input gradient (modules/mlp.py).  Off by default: measured on
        # MI355X it buys nothing — the recorded GEMM choices run the 479-wide layer as fast at lda = 479 as at 480
        # (0.481 / 0.464 / 0.439 ms vs 0.479 / 0.464 / 0.445 ms) and the interaction kernels are not store-bound.
        self.pad_rows = os.environ.get("TORCHREC_AMD_PAD_INTERACTION", "0") == "1"
        self.grad_sinks = None  # (d dense, d sparse) buffers the fused backward writes into; read at forward time
        self.register_buffer("triu_indices", torch.triu_indices(self.F + 1, self.F + 1, offset=1), persistent=False)

    def forward(self, dense_features: torch.Tensor, sparse_features: torch.Tensor) -> torch.Tensor:
        if self.F <= 0:
            return dense_features
        if self.fused and _fused_interaction_ok(dense_features, sparse_features):
            return _FusedDotInteraction.apply(dense_features, sparse_features, self.pad_rows, self.grad_sinks)
        # generic shapes: the reference's formulation on torch ops
        combined = torch.cat((dense_features.unsqueeze(1), sparse_features), dim=1)
        interactions = torch.bmm(combined, combined.transpose(1, 2))
        flat = interactions[:, self.triu_indices[0], self.triu_indices[1]]
        return torch.cat((dense_features, flat), dim=1)


class OverArch(nn.Module):
    def __init__(self, in_features: int, layer_sizes: List[int], device: Optional[torch.device] = None) -> None:
        super().__init__()
        if len(layer_sizes) <= 1:
            raise ValueError("OverArch must have multiple layers.")
        self.model = nn.Sequential(
            MLP(in_features, layer_sizes[:-1], bias=True, activation="relu", device=device),
            LinearOut(layer_sizes[-2], layer_sizes[-1], bias=True, device=device))

    def forward(self, features: torch.Tensor) -> torch.Tensor:
        return self.model(features)


class DLRM(nn.Module):
    def __init__(self, embedding_bag_collection: nn.Module, dense_in_features: int,
                 dense_arch_layer_sizes: List[int], over_arch_layer_sizes: List[int],
                 dense_device: Optional[torch.device] = None) -> None:
        super().__init__()
        cfgs = embedding_bag_collection.embedding_bag_configs
        assert len(cfgs) > 0, "At least one embedding bag is required"
        D = cfgs[0].embedding_dim
        assert all(c.embedding_dim == D for c in cfgs), "All EmbeddingBagConfigs must have the same dimension"
        if dense_arch_layer_sizes[-1] != D:
            raise ValueError(f"embedding_bag_collection dimension ({D}) and final dense arch layer size "
                             f"({dense_arch_layer_sizes[-1]}) must match.")
        self.sparse_arch = SparseArch(embedding_bag_collection)
        F = self.sparse_arch.F
        self.dense_arch = DenseArch(dense_in_features, dense_arch_layer_sizes, device=dense_device)
        self.inter_arch = InteractionArch(F)
        over_in = D + (F + 1) * F // 2
        self.over_arch = OverArch(over_in, over_arch_layer_sizes, device=dense_device)
        if dense_device is not None:
            self.inter_arch.to(dense_device)

    def forward(self, dense_features: torch.Tensor, sparse_features: KeyedJaggedTensor) -> torch.Tensor:
        # The reference runs dense_arch, then sparse_arch (models/dlrm.py:400-401).  Here the lookup and
        # the pooled all-to-all are issued first so that the exchange overlaps the bottom MLP on the
        # collective's own HIP stream; the two branches are independent, results are identical.
        # The embedding backward's sort (side stream, gradient-independent) is started AFTER the HBM-bound part of the
        # forward (lookup, interaction) and runs beside the over-arch GEMMs, which are MFMA-bound and leave the
        # vector units, LDS and memory queues it needs mostly idle.
        defer, launch = _sort_hooks(self.sparse_arch.embedding_bag_collection)
        if defer is not None:
            defer(self.training and torch.is_grad_enabled())
        pending = self.sparse_arch.start(sparse_features)
        embedded_dense = self.dense_arch(dense_features)
        embedded_sparse = self.sparse_arch.finish(pending)
        concatenated = self.inter_arch(dense_features=embedded_dense, sparse_features=embedded_sparse)
        if launch is not None:
            launch()
        return self.over_arch(concatenated)


# nn.BCEWithLogitsLoss (mean) as ONE kernel for loss + gradient (csrc/mlp_epilogue.hip bce_with_logits_kernel) instead of 8
# element-wise / reduce kernels forward and 5 backward; TORCHREC_AMD_FUSED_BCE=0 keeps torch's
_FUSED_BCE = os.environ.get("TORCHREC_AMD_FUSED_BCE", "1") != "0"


class _FusedBCEWithLogits(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels):
        from ..distributed import _device_ops  # noqa: F401  (registers torch.ops.tbe_hip.*)

        loss, dlogits = torch.ops.tbe_hip.bce_with_logits(logits, labels)
        ctx.save_for_backward(dlogits)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        (dlogits,) = ctx.saved_tensors
        return dlogits * grad_out, None


def bce_with_logits_mean(loss_fn: nn.Module, logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """`loss_fn(logits, labels.float())` for the train wrapper's nn.BCEWithLogitsLoss (examples/dlrm/modules/dlrm_train.py);
    on a HIP device, for the plain mean-reduced loss over float32 logits [B], one fused kernel (labels int64 or float)."""
    if (_FUSED_BCE and logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 1 and labels.shape == logits.shape
            and type(loss_fn) is nn.BCEWithLogitsLoss and loss_fn.reduction == "mean" and loss_fn.weight is None
            and loss_fn.pos_weight is None and logits.numel() > 0 and labels.dtype in (torch.float32, torch.int64)):
        return _FusedBCEWithLogits.apply(logits, labels)
    return loss_fn(logits, labels.float())


class _Head(nn.Module):
    """interaction + over arch + loss as ONE static-shape segment (inputs: bottom-MLP output, pooled
    embeddings [B, F, D], labels as they come: int64 or float) -> (loss, logits)."""

    def __init__(self, inter_arch: nn.Module, over_arch: nn.Module, loss_fn: nn.Module) -> None:
        super().__init__()
        self.inter_arch, self.over_arch, self.loss_fn = inter_arch, over_arch, loss_fn

    def forward(self, embedded_dense: torch.Tensor, embedded_sparse: torch.Tensor, labels: torch.Tensor):
        logits = self.over_arch(self.inter_arch(dense_features=embedded_dense, sparse_features=embedded_sparse)).squeeze(-1)
        # the logits leave the segment for metrics only: detached, so that the captured backward has ONE root (no zero
        # gradient buffer for a second output, no add of the two contributions to d logits)
        return bce_with_logits_mean(self.loss_fn, logits, labels), logits.detach()


class DLRMTrain(nn.Module):
    """examples/dlrm/modules/dlrm_train.py: BCEWithLogits wrapper used by the train pipeline."""

    def __init__(self, embedding_bag_collection: nn.Module, dense_in_features: int,
                 dense_arch_layer_sizes: List[int], over_arch_layer_sizes: List[int],
                 dense_device: Optional[torch.device] = None) -> None:
        super().__init__()
        self.model = DLRM(embedding_bag_collection, dense_in_features, dense_arch_layer_sizes,
                          over_arch_layer_sizes, dense_device)
        self.loss_fn = nn.BCEWithLogitsLoss()
        self._graphs = None  # (batch size, bottom-MLP segment, head segment)

    def capture_hip_graphs(self, batch_size: int, flat_grads: bool = False, process_group=None,
                           half_batches: Optional[bool] = None) -> None:
        """Captures the two collective-free dense segments of a train step — bottom MLP; interaction
        + top MLP + loss — as HIP graphs for this per-rank batch size (distributed/hip_graph.py).
        Steps with another batch size, eval mode or no_grad run eagerly as before.

        flat_grads: the backward graphs write the parameter gradients, already divided by the world
        size, into ONE flat buffer; the head's slice is all-reduced asynchronously as soon as its backward
        replay is enqueued, the rest (bottom segment + the model's remaining dense parameters, i.e. the
        replicated tiny tables) by `finish_dense_grads()`, which also attaches the slices as `.grad`.
        This replaces DistributedDataParallel altogether (its forward wrapper alone cost 0.43 ms of host
        time per step, plus 16 per-parameter bucket copies — the per-rank step of an N > 1 run is
        host-bound); DistributedModelParallel.init_data_parallel() broadcasts rank 0's values instead."""
        from ..distributed.hip_graph import GraphedSegment

        m = self.model
        p = next(m.dense_arch.parameters())
        dev, B = p.device, batch_size
        F, D = m.sparse_arch.F, m.sparse_arch.D
        dense_in = m.dense_arch.model._mlp[0]._in_size
        head = _Head(m.inter_arch, m.over_arch, self.loss_fn)
        flat_param = None
        if flat_grads:
            # the parameters move into ONE flat buffer too (same order as the flat gradient: head, bottom MLP, the
            # rest), before anything captures their addresses: the dense optimizer becomes one kernel (optim/flat.py)
            seg_head = [q for q in head.parameters() if q.requires_grad]
            seg_dense = [q for q in m.dense_arch.parameters() if q.requires_grad]
            seen = {id(q) for q in seg_head + seg_dense}
            seg_rest = [q for q in self.parameters() if q.requires_grad and id(q) not in seen]
            if all(q.dtype == torch.float32 and q.device == dev for q in seg_head + seg_dense + seg_rest):
                # every segment starts on a 256-B boundary (the head ends with a 1-element bias: without the padding the
                # bottom MLP's weights and the replicated tables would sit at addresses that are 4 mod 16)
                starts, total = [], 0
                for seg in (seg_head, seg_dense, seg_rest):
                    starts.append(total)
                    total = _pad64(total + sum(q.numel() for q in seg))
                flat_param = torch.zeros(total, dtype=torch.float32, device=dev)
                with torch.no_grad():
                    for seg, off in zip((seg_head, seg_dense, seg_rest), starts):
                        for q in seg:
                            view = flat_param[off:off + q.numel()].view_as(q)
                            view.copy_(q)
                            q.data = view
                            off += q.numel()
        ebc = m.sparse_arch.embedding_bag_collection
        ebc = getattr(ebc, "sharded", ebc)
        if half_batches is None:
            half_batches = _HALF_BATCHES == "1" or (_HALF_BATCHES == "auto" and B >= _HALF_BATCH_MIN)
        halves = bool(half_batches and flat_grads and B % 2 == 0 and getattr(ebc, "_exchange", False)
                      and hasattr(ebc, "set_half_batch_exchange") and _EXPLICIT_STEP)
        g_dense = GraphedSegment(m.dense_arch, [torch.randn(B, dense_in, device=dev)])
        if halves and hasattr(ebc, "set_graph_exchange"):
            ebc.set_graph_exchange(None)
        if halves:
            # whole-batch buffers the two head segments share by halves: pooled embeddings (the collection's output
            # buffer), labels, and the gradients w.r.t. the bottom-MLP output / the pooled embeddings
            Bh = B // 2
            Dd = g_dense.static_outputs[0].shape[1]
            half = {"pooled": torch.zeros(B, F, D, device=dev), "labels": torch.zeros(B, dtype=torch.int64, device=dev),
                    "gd": torch.zeros(B, Dd, device=dev), "gs": torch.zeros(B, F, D, device=dev), "Bh": Bh}
            heads = []
            for h in range(2):
                rows = slice(h * Bh, (h + 1) * Bh)
                m.inter_arch.grad_sinks = (half["gd"][rows], half["gs"][rows])
                try:
                    heads.append(GraphedSegment(
                        head, [g_dense.static_outputs[0][rows].detach().requires_grad_(True),
                               half["pooled"][rows].detach().requires_grad_(True), half["labels"][rows]],
                        input_buffers=[g_dense.static_outputs[0][rows], half["pooled"][rows], half["labels"][rows]],
                        pool=g_dense._pool))
                finally:
                    m.inter_arch.grad_sinks = None
            g_head = heads[1]
        else:
            # static-exchange mode (flat gradients + explicit step + a collection that exchanges): the unpack of the pooled
            # all-to-all's receive buffer is the FIRST kernel of the head's forward graph and the pack of the gradient
            # into the send buffer the LAST of its backward graph, instead of two eager launches per step
            hooks = None
            if (flat_grads and _EXPLICIT_STEP and _GRAPH_EXCHANGE and getattr(ebc, "_exchange", False)
                    and hasattr(ebc, "set_graph_exchange")):
                pooled = torch.zeros(B, F, D, device=dev)
                ebc.set_output_buffer(pooled)
                hooks = ebc.set_graph_exchange(B)
            elif hasattr(ebc, "set_graph_exchange"):
                ebc.set_graph_exchange(None)
            g_head = GraphedSegment(
                head, [g_dense.static_outputs[0].detach().requires_grad_(True),
                       torch.randn(B, F, D, device=dev).requires_grad_(True),
                       torch.randint(0, 2, (B,), device=dev)],  # int64, as the data loader delivers them
                input_buffers=[g_dense.static_outputs[0], pooled if hooks is not None else None, None], pool=g_dense._pool,
                pre_forward=hooks[0] if hooks is not None else None)
        head_sinks = dense_sinks = None
        scale = 1.0
        if flat_grads:
            import torch.distributed as dist

            world = dist.get_world_size(process_group) if process_group is not None else 1
            scale = 1.0 / world
            dense_pg = process_group
            # TORCHREC_AMD_FORCE_DENSE_REDUCE=1 (rehearsal of the N > 1 path on one rank): issue the dense all-reduces — and
            # create their communicator — although a one-rank group needs none
            reduce = world > 1 or (process_group is not None and os.environ.get("TORCHREC_AMD_FORCE_DENSE_REDUCE") == "1")
            if reduce and dist.get_backend(process_group) == "nccl" and os.environ.get("TORCHREC_AMD_DENSE_PG", "1") == "1":
                # the dense all-reduces get a communicator (and stream) of their own: on the collection's they would queue
                # behind a pooled all-to-all that is waiting for its links (the prefetched one, above all)
                from ..distributed.comm import new_rccl_group

                dense_pg = new_rccl_group(process_group)
            graphed = {id(q) for q in list(g_head._params) + list(g_dense._params)}
            # every other trainable dense parameter of the model (the replicated tiny tables of a sharded
            # collection): its gradient arrives through autograd and joins the flat buffer after backward
            extras = [q for q in self.parameters() if q.requires_grad and id(q) not in graphed]
            # same layout as the flat parameter buffer: [head | pad | bottom MLP | pad | rest | pad], segments 256-B aligned
            n_head = _pad64(sum(q.numel() for q in g_head._params))
            n_dense = _pad64(sum(q.numel() for q in g_dense._params))
            n_extra = _pad64(sum(q.numel() for q in extras))
            flat = torch.zeros(n_head + n_dense + n_extra, dtype=torch.float32, device=dev)

            def views(params, off):
                out = []
                for q in params:
                    out.append(flat[off:off + q.numel()].view_as(q))
                    off += q.numel()
                return out

            head_sinks, dense_sinks = views(g_head._params, 0), views(g_dense._params, n_head)
            extra_views = views(extras, n_head + n_dense)
            state = {"flat": flat, "flat_param": flat_param, "params": list(g_head._params) + list(g_dense._params) + extras,
                     "views": head_sinks + dense_sinks + extra_views, "extras": list(zip(extras, extra_views)),
                     "n_head": n_head, "n_dense": n_dense, "works": [], "pg": dense_pg, "world": world, "scale": scale,
                     "reduce": reduce}

            def reduce_head():  # the head's backward runs first: its slice overlaps the rest of backward
                state["fired"] += 1
                # with a late part (its gradients do not exist yet) what is left of the head's slice is two small layers:
                # they ride with the rest of the buffer (_start_rest_reduce) instead of paying for a collective of their own
                if state["reduce"] and not state.get("n_late", 0):
                    state["works"].append(dist.all_reduce(flat[:n_head], group=state["pg"], async_op=True))

            def reduce_late():
                if state["reduce"] and state.get("n_late", 0):
                    state["works"].append(dist.all_reduce(flat[:state["n_late"]], group=state["pg"], async_op=True))

            state["reduce_late"] = reduce_late

            def dense_done():
                state["fired"] += 1

            state["fired"] = 0
            if hasattr(ebc, "set_replicated_grad_sink"):
                dp = getattr(ebc, "_dp_module", None)
                sink = next((v for q, v in zip(extras, extra_views) if dp is not None and q is dp.weights), None)
                ebc.set_replicated_grad_sink(sink)
            g_head.after_backward = reduce_head
            g_dense.after_backward = dense_done
            object.__setattr__(self, "_flat_dense", state)
        late = 0
        if flat_grads and _DEFER_WGRAD and _EXPLICIT_STEP and getattr(ebc, "_exchange", False):
            # (weight, bias) of the over arch's first layers: the leading parameters of the head segment
            n_lin = sum(1 for q in g_head._params if q.dim() == 2)
            late = 2 * min(_late_layers(B, halves), max(n_lin - 1, 0))
            if late and not all(g_head._params[j].dim() == (2 if j % 2 == 0 else 1) for j in range(late)):
                late = 0  # not a plain (weight, bias) sequence: keep everything in the second graph
        # the embedding collection writes its pooled output straight into the head segment's static input
        if hasattr(ebc, "set_half_batch_exchange"):
            ebc.set_half_batch_exchange(halves)
        if halves:
            ebc.set_output_buffer(half["pooled"])
            # half 0's parameter gradients go to a buffer of their own and are added to half 1's (which sit in the flat
            # buffer) by one launch after both weight-gradient graphs
            tmp = torch.zeros(n_head, dtype=torch.float32, device=dev)
            tmp_sinks, off = [], 0
            for q in g_head._params:
                tmp_sinks.append(tmp[off:off + q.numel()].view_as(q))
                off += q.numel()
            half["tmp"], half["n_head_used"] = tmp, off
            for h, gh in enumerate(heads):
                gh.capture_backward(param_grad_sinks=tmp_sinks if h == 0 else head_sinks, sink_scale=scale,
                                    defer_wgrad=_DEFER_WGRAD, late_params=late)
                rows = slice(h * Bh, (h + 1) * Bh)
                if (gh.static_grad_inputs[0].data_ptr() != half["gd"][rows].data_ptr()
                        or gh.static_grad_inputs[1].data_ptr() != half["gs"][rows].data_ptr()):
                    raise RuntimeError("capture_hip_graphs(half_batches=True): the head segment's input gradients did not land "
                                       "in the shared whole-batch buffers (is the fused dot interaction in use?)")
            heads[1].after_backward = heads[0].after_backward = None
            half["heads"], half["reduce_head"] = heads, reduce_head
            state["n_late"] = sum(q.numel() for q in g_head._params[:late]) if heads[1].bwd_graph3 is not None else 0
            g_dense.capture_backward([half["gd"]], param_grad_sinks=dense_sinks, sink_scale=scale, defer_wgrad=_DEFER_WGRAD)
            object.__setattr__(self, "_half", half)
            object.__setattr__(self, "_graph_exchange", False)
        else:
            if hasattr(ebc, "set_output_buffer") and hooks is None:
                ebc.set_output_buffer(g_head.static_input(1).detach())
            # flat mode: the head's weight gradients go into a second graph that the explicit step replays after it has
            # started the embedding-gradient all-to-all (modules/mlp.py _DeferredWgrad)
            g_head.capture_backward(param_grad_sinks=head_sinks, sink_scale=scale, defer_wgrad=flat_grads and _DEFER_WGRAD,
                                    late_params=late,
                                    post_backward=(lambda gin: hooks[1](gin[1])) if hooks is not None else None)
            object.__setattr__(self, "_graph_exchange", hooks is not None)
            if flat_grads:
                state["n_late"] = sum(q.numel() for q in g_head._params[:late]) if g_head.bwd_graph3 is not None else 0
                state["split_mode"] = _WGRAD_SPLIT_MODE
            # the head's gradient w.r.t. the bottom-MLP output doubles as the bottom segment's grad_output buffer
            g_dense.capture_backward([g_head.static_grad_inputs[0]], param_grad_sinks=dense_sinks, sink_scale=scale,
                                     defer_wgrad=flat_grads and _DEFER_WGRAD)
            object.__setattr__(self, "_half", None)
        object.__setattr__(self, "_graphs", (B, g_dense, g_head))  # not sub-modules: state_dict keys unchanged
        # the explicit step fills d(loss) = 1 into the NEW head segment's grad-output buffer (zeros after capture)
        object.__setattr__(self, "_loss_grad_ready", False)

    def dense_optimizer(self, params, lr: float, momentum: float = 0.0, weight_decay: float = 0.0) -> torch.optim.Optimizer:
        """SGD for the dense parameters (what examples/dlrm/dlrm_main.py:536-540 builds with torch.optim.SGD): one
        kernel over the flat parameter / gradient buffers when capture_hip_graphs(flat_grads=True) has laid them out
        (optim/flat.py), torch.optim.SGD otherwise — same hyper-parameters, same arithmetic.  Call it AFTER
        capture_hip_graphs (a later capture moves the parameters into NEW flat buffers)."""
        params = list(params)
        st = getattr(self, "_flat_dense", None)
        if (st is not None and st.get("flat_param") is not None and {id(q) for q in st["params"]} <= {id(q) for q in params}
                and os.environ.get("TORCHREC_AMD_FLAT_SGD", "1") != "0"):
            from ..optim.flat import FlatSGD

            return FlatSGD(params, lr, flat_param=st["flat_param"], flat_grad=st["flat"], covered=st["params"],
                           grad_views=st["views"], momentum=momentum, weight_decay=weight_decay)
        return torch.optim.SGD(params, lr=lr, momentum=momentum, weight_decay=weight_decay)

    def flat_grad_parameters(self) -> List[nn.Parameter]:
        """Parameters whose gradients travel through the flat buffer (kept out of DDP)."""
        st = getattr(self, "_flat_dense", None)
        return list(st["params"]) if st is not None else []

    def _start_rest_reduce(self, st) -> None:
        """Graph-replayed step: folds the autograd / explicit gradients of the non-graphed dense parameters (the replicated
        tiny tables) into the flat buffer and starts the all-reduce of everything behind the head's slice (bottom MLP +
        those).  The explicit step calls this as soon as the bottom MLP's backward graph is enqueued — before the fused
        embedding backward and before the next step's prefetched lookup + pooled all-to-all, so that on the collective
        stream the (small) dense reduction is queued AHEAD of that all-to-all and overlaps the embedding update."""
        if st.get("rest_started"):
            return
        import torch.distributed as dist

        self._fold_extras(st)
        if st["reduce"]:
            if st.get("pieces_done"):  # "early" split mode: head and replicated tables are on their way, the bottom MLP is left
                lo, hi = st["n_head"], st["n_head"] + st["n_dense"]
            else:
                lo, hi = (st["n_late"] if st.get("n_late", 0) else st["n_head"]), st["flat"].numel()  # (see reduce_head)
            st["works"].append(dist.all_reduce(st["flat"][lo:hi], group=st["pg"], async_op=True))
        st["rest_started"] = True
        st["pieces_done"] = False

    @staticmethod
    def _fold_extras(st) -> None:
        if st.get("extras_folded"):
            return
        for q, v in st["extras"]:
            if q.grad is None:
                v.zero_()
            elif q.grad.data_ptr() != v.data_ptr():
                torch.mul(q.grad, st["scale"], out=v)
            elif st["scale"] != 1.0:
                v.mul_(st["scale"])  # autograd accumulated in place (zero_grad(set_to_none=False))
        st["extras_folded"] = True

    def finish_dense_grads(self) -> None:
        """After backward: folds the autograd gradients of the non-graphed dense parameters into the flat
        buffer, all-reduces the rest of it (bottom segment + those), waits, and attaches the slices as
        `.grad` (every rank then holds the rank-averaged gradient, as under DistributedDataParallel)."""
        from ..modules.mlp import _DeferredFinish, _WgradOverlap

        _WgradOverlap.join()  # weight gradients computed on the side stream (eager steps; modules/mlp.py)
        _DeferredFinish.flush()  # eager steps: every split-K / bias gradient of this backward finished by one launch
        st = getattr(self, "_flat_dense", None)
        if st is None:
            return
        replayed = st["fired"] == 2
        st["fired"] = 0
        import torch.distributed as dist

        if not replayed:
            # this step ran eagerly (another batch size, ...): every gradient came through autograd; fold all of
            # them into the flat buffer and reduce it as a whole
            for w in st["works"]:
                w.wait()
            st["works"].clear()
            for q, v in zip(st["params"], st["views"]):
                if q.grad is None:
                    v.zero_()
                elif q.grad.data_ptr() != v.data_ptr():
                    torch.mul(q.grad, st["scale"], out=v)
                elif st["scale"] != 1.0:
                    v.mul_(st["scale"])
            if st["reduce"]:
                st["works"].append(dist.all_reduce(st["flat"], group=st["pg"], async_op=True))
        else:
            self._start_rest_reduce(st)
        st["rest_started"] = False
        st["extras_folded"] = False
        for w in st["works"]:
            w.wait()
        st["works"].clear()
        for q, v in zip(st["params"], st["views"]):
            q.grad = v

    # ---- explicit train step (HIP-graph + flat-gradient mode): forward AND backward without the autograd engine --------
    def set_between_forward_and_backward(self, fn, prefetch=None) -> None:
        """A callable the explicit step runs once between its forward and its backward (the train pipeline starts the
        next batch's input dist there, as it does between `model(batch)` and `loss.backward()` otherwise).

        `prefetch`: a callable the explicit step runs right behind its LAST backward launch (the fused embedding
        backward): it returns (next batch's KeyedJaggedTensor, its ExplicitLookupStep) or None.  The next step's lookup
        and pooled all-to-all depend on the tables this step's embedding backward has just updated and on nothing else
        — not on the dense optimizer, the dense gradient all-reduce or the host's step-boundary work — so they are
        enqueued here and the GPU has work while the host does that (the step boundary was the largest idle gap of the
        per-rank step: 98 us of 1.99 ms, profiles/r02_rehearsal_b8192_explicit_gaps.txt)."""
        object.__setattr__(self, "_between", fn)
        object.__setattr__(self, "_prefetch", prefetch)

    @property
    def wants_lookup_prefetch(self) -> bool:
        """True when the captured head segment keeps part of its weight gradients for the window behind the next step's
        prefetched lookup (capture_hip_graphs, TORCHREC_AMD_WGRAD_LATE_LAYERS): the pipeline then prefetches by default."""
        g = getattr(self, "_graphs", None)
        st = getattr(self, "_flat_dense", None)
        return bool(g is not None and getattr(g[2], "bwd_graph3", None) is not None
                    and (st is None or st.get("split_mode", "late") == "late"))

    def take_backward_done(self) -> bool:
        """True (once) when the latest forward() already ran the backward: the caller must not call loss.backward()."""
        done = getattr(self, "_backward_done", False)
        object.__setattr__(self, "_backward_done", False)
        return done

    def _explicit_step(self, batch, g_dense, g_head):
        """One train step of the graphed configuration driven by hand: lookup -> all-to-all || bottom-MLP graph ->
        head graph (forward + backward replays) -> gradient all-to-all || bottom-MLP backward graph -> embedding
        backward.  Same kernels, same order of the dependent work as the autograd path; what disappears is the engine
        (its worker thread, ~12 Python Function nodes and their hand-offs: ~0.25 ms of host time per step, which is the
        bottleneck of the N > 1 per-rank step on a slow host).  Returns None if the collection cannot run it."""
        ebc = self.model.sparse_arch.embedding_bag_collection
        if not hasattr(ebc, "compute_explicit"):
            return None
        inner = getattr(ebc, "sharded", ebc)
        if list(inner._feature_names) != list(self.model.sparse_arch.sparse_feature_names):
            return None
        kjt = batch.sparse_features
        pre = getattr(self, "_prefetched", None)
        object.__setattr__(self, "_prefetched", None)
        if pre is not None and pre[0] is kjt and getattr(pre[1], "epoch", 0) != getattr(inner, "_weights_epoch", 0):
            # the tables were rewritten (load_state_dict) after the prefetch: look up again, on the same distributed ids
            pre[1].discard()
            pre = (kjt, inner.compute_explicit(pre[1].d))
            object.__setattr__(self, "prefetched_lookups", getattr(self, "prefetched_lookups", 0) - 1)
        if pre is not None and pre[0] is kjt:
            step = pre[1]  # looked up (and its all-to-all started) at the end of the previous step
            object.__setattr__(self, "prefetched_lookups", getattr(self, "prefetched_lookups", 0) + 1)
        else:
            if pre is not None:
                raise RuntimeError("DLRMTrain: a lookup was prefetched for another batch than the one this step received; the "
                                   "owner of the pipeline must feed the batches in the order it announced them")
            piped = getattr(ebc, "_pipelined", None)  # a train pipeline's forward on this collection (queued input dist)
            step = (piped.compute_explicit(kjt) if piped is not None
                    else (ebc.compute_explicit(ebc.input_dist(kjt).wait()) if ebc.explicit_step_supported(kjt.stride()) else None))
        if step is None:
            return None
        B = g_dense.static_inputs[0].shape[0]
        half = getattr(self, "_half", None)
        if half is not None:
            if not getattr(step, "halves", False):
                raise RuntimeError("DLRMTrain: the head segments were captured for half-batch exchanges, the collection did "
                                   "not start one")
            return self._explicit_step_halves(batch, step, g_dense, half, B)
        with torch.no_grad():
            # bottom MLP (graph) while the pooled all-to-all is in flight
            if batch.dense_features.data_ptr() != g_dense.static_inputs[0].data_ptr():
                g_dense.static_inputs[0].copy_(batch.dense_features)
            g_dense.fwd_graph.replay()
            pooled = step.finish()  # written into the head segment's static input by the collection
            if pooled.data_ptr() != g_head.static_inputs[1].data_ptr():
                g_head.static_inputs[1].copy_(pooled.view_as(g_head.static_inputs[1]))
            g_head.static_inputs[2].copy_(batch.labels)
            g_head.fwd_graph.replay()
            loss, logits = g_head.static_outputs[0], g_head.static_outputs[1]
            between = getattr(self, "_between", None)
            if between is not None:
                between()
            # backward: d(loss) = 1
            with label("## backward ##"):  # train_pipeline.py:546
                ones = getattr(self, "_loss_grad_ready", False)
                if not ones:
                    g_head.static_grad_outputs[0].fill_(1.0)
                    object.__setattr__(self, "_loss_grad_ready", True)
                g_head.bwd_graph.replay()  # ends with the gradient of the pooled embeddings
                step.start_backward(g_head.static_grad_inputs[1].view(B, -1))  # pack + gradient all-to-all (+ replicated tables)
                st = getattr(self, "_flat_dense", None)
                early = (getattr(g_head, "bwd_graph3", None) is not None and st is not None
                         and st.get("split_mode") == "early")
                if early:
                    # every piece of the flat gradient is all-reduced as soon as it exists, the largest first: the replicated
                    # tables' (written by start_backward above), the head's largest layers', the head's other layers' — and
                    # the bottom MLP's small slice last (_start_rest_reduce), behind its backward graphs
                    import torch.distributed as dist

                    self._fold_extras(st)
                    if st["reduce"]:
                        st["works"].append(dist.all_reduce(st["flat"][st["n_head"] + st["n_dense"]:], group=st["pg"], async_op=True))
                    g_head.bwd_graph3.replay()
                    st["reduce_late"]()
                if getattr(g_head, "bwd_graph2", None) is not None:
                    g_head.bwd_graph2.replay()  # the head's weight gradients, while the all-to-all is in flight
                if early:
                    st["fired"] += 1
                    if st["reduce"]:
                        st["works"].append(dist.all_reduce(st["flat"][st["n_late"]:st["n_head"]], group=st["pg"], async_op=True))
                    st["pieces_done"] = True
                elif g_head.after_backward is not None:
                    g_head.after_backward()  # all-reduce of the head's slice of the flat gradient
                g_dense.bwd_graph.replay()  # its grad_output buffer IS the head's gradient w.r.t. the bottom-MLP output
                if getattr(g_dense, "bwd_graph2", None) is not None:
                    g_dense.bwd_graph2.replay()  # its weight / bias gradients, finished into the flat buffer
                if g_dense.after_backward is not None:
                    g_dense.after_backward()
                st = getattr(self, "_flat_dense", None)
                if st is not None and st["fired"] == 2:
                    self._start_rest_reduce(st)  # ahead of the embedding update and of the next step's all-to-all
                step.finish_backward()
            prefetch = getattr(self, "_prefetch", None)
            if prefetch is not None:
                object.__setattr__(self, "_prefetched", prefetch())  # (kjt, ExplicitLookupStep) of the NEXT batch, or None
            if getattr(g_head, "bwd_graph3", None) is not None and not early:
                # the late part of the head's weight gradients: behind the NEXT step's lookup + pooled all-to-all when the
                # owner prefetches (its cover on the links), right here otherwise
                g_head.bwd_graph3.replay()
                if st is not None:
                    st["reduce_late"]()
        object.__setattr__(self, "_backward_done", True)
        object.__setattr__(self, "explicit_steps", getattr(self, "explicit_steps", 0) + 1)  # for tests / bench.py's line
        return loss.detach(), (loss.detach(), logits.detach(), batch.labels.detach())

    def _explicit_step_halves(self, batch, step, g_dense, half, B: int):
        """The explicit step with the exchange in two half-batches (capture_hip_graphs(half_batches=True)):

            lookup (whole batch) -> all-to-all half 0, half 1 -> bottom MLP
            half 0: wait, unpack, head forward, head backward (input gradients), pack, gradient all-to-all 0
            half 1: the same — its embeddings crossed the links during half 0's work, half 0's gradients cross during this
            weight gradients of both halves, bottom-MLP backward (gradient all-to-all 1 in flight), embedding backward."""
        heads, Bh = half["heads"], half["Bh"]
        with torch.no_grad():
            if batch.dense_features.data_ptr() != g_dense.static_inputs[0].data_ptr():
                g_dense.static_inputs[0].copy_(batch.dense_features)
            g_dense.fwd_graph.replay()
            half["labels"].copy_(batch.labels)
            if not getattr(self, "_loss_grad_ready", False):
                for gh in heads:
                    gh.static_grad_outputs[0].fill_(0.5)  # loss = (mean of half 0 + mean of half 1) / 2
                object.__setattr__(self, "_loss_grad_ready", True)
            gs_all = half["gs"].view(B, -1)
            between = getattr(self, "_between", None)
            with label("## backward ##"):  # train_pipeline.py:546 (the halves' forwards are inside: they alternate)
                for h, gh in enumerate(heads):
                    step.finish_half(h)  # into the rows of the pooled buffer this segment reads
                    gh.fwd_graph.replay()
                    if h == 0 and between is not None:
                        between()
                    gh.bwd_graph.replay()  # ends with the gradients of this half's pooled embeddings / bottom-MLP output
                    step.start_backward_half(h, gs_all[h * Bh:(h + 1) * Bh], grad_out=gs_all if h == 1 else None)
                for gh in heads:
                    if getattr(gh, "bwd_graph2", None) is not None:
                        gh.bwd_graph2.replay()  # weight gradients, while the gradient all-to-alls are in flight
                n, nl = half["n_head_used"], self._flat_dense.get("n_late", 0)
                st = self._flat_dense
                st["flat"][nl:n].add_(half["tmp"][nl:n])
                half["reduce_head"]()  # all-reduce of the head's slice of the flat gradient (without the late part)
                g_dense.bwd_graph.replay()
                if getattr(g_dense, "bwd_graph2", None) is not None:
                    g_dense.bwd_graph2.replay()
                if g_dense.after_backward is not None:
                    g_dense.after_backward()
                if st["fired"] == 2:
                    self._start_rest_reduce(st)
                step.finish_backward()
            prefetch = getattr(self, "_prefetch", None)
            if prefetch is not None:
                object.__setattr__(self, "_prefetched", prefetch())  # (kjt, ExplicitLookupStep) of the NEXT batch, or None
            if nl:
                for gh in heads:
                    gh.bwd_graph3.replay()  # late weight gradients: behind the next step's first half-batch all-to-all
                st["flat"][:nl].add_(half["tmp"][:nl])
                st["reduce_late"]()
            loss = (heads[0].static_outputs[0] + heads[1].static_outputs[0]) * 0.5
            logits = torch.cat([heads[0].static_outputs[1], heads[1].static_outputs[1]])
        object.__setattr__(self, "_backward_done", True)
        object.__setattr__(self, "explicit_steps", getattr(self, "explicit_steps", 0) + 1)
        object.__setattr__(self, "half_batch_steps", getattr(self, "half_batch_steps", 0) + 1)  # for tests / bench.py's line
        return loss.detach(), (loss.detach(), logits.detach(), batch.labels.detach())

    def forward(self, batch) -> Tuple[torch.Tensor, Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
        g = self._graphs
        if (g is not None and self.training and torch.is_grad_enabled()
                and batch.dense_features.shape[0] == g[0] and batch.dense_features.is_cuda):
            _, g_dense, g_head = g
            if (getattr(self, "_flat_dense", None) is not None and getattr(self, "_between", None) is not None
                    and _EXPLICIT_STEP):
                # only under an owner that asked for it (set_between_forward_and_backward) and will skip loss.backward()
                out = self._explicit_step(batch, g_dense, g_head)
                object.__setattr__(self, "_between", None)
                object.__setattr__(self, "_prefetch", None)
                if out is not None:
                    return out
            if getattr(self, "_half", None) is not None:
                raise RuntimeError("DLRMTrain: graphs captured with half_batches=True serve the explicit step only (run the "
                                   "model under TrainPipelineSparseDist, or capture with half_batches=False)")
            if getattr(self, "_graph_exchange", False):
                raise RuntimeError("DLRMTrain: these graphs unpack / pack the pooled exchange themselves and serve the explicit "
                                   "step only (run the model under TrainPipelineSparseDist, or set TORCHREC_AMD_GRAPH_EXCHANGE=0)")
            defer, launch = _sort_hooks(self.model.sparse_arch.embedding_bag_collection)
            if defer is not None:
                defer(True)
            pending = self.model.sparse_arch.start(batch.sparse_features)
            embedded_dense = g_dense(batch.dense_features)
            embedded_sparse = self.model.sparse_arch.finish(pending)
            if launch is not None:
                launch()  # beside the head segment (interaction + over arch), after the lookup and the exchange
            object.__setattr__(self, "_loss_grad_ready", False)  # autograd writes whatever d(loss) the caller backpropagates
            loss, logits = g_head(embedded_dense, embedded_sparse, batch.labels.to(g_head.static_inputs[2].dtype))
            return loss, (loss.detach(), logits.detach(), batch.labels.detach())
        logits = self.model(batch.dense_features, batch.sparse_features).squeeze(-1)
        loss = bce_with_logits_mean(self.loss_fn, logits, batch.labels)
        return loss, (loss.detach(), logits.detach(), batch.labels.detach())
