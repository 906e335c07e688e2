"""Forward + backward of a dense sub-module as two HIP graphs.

Why: at the per-rank batch of the 8-GPU configuration (8 192 samples) the DLRM step is bound by
host launch overhead, not by the GPU: the bottom MLP, the dot interaction, the top MLP and the loss
are ~70 small launches per direction.  Those segments have static shapes and contain no collective,
so each is captured once (hipStreamBeginCapture through torch.cuda.graph) and replayed with one
launch per direction.  The embedding lookup, the all-to-alls and the gradient all-reduce stay
outside the graphs; autograd sees one Function per segment, so DistributedDataParallel's gradient
hooks keep firing on the segment's parameters exactly as before.

The reference has no counterpart (it launches every op eagerly through
torchrec/distributed/train_pipeline.py:520-552); this is the MI355X-side replacement for a
tracing compiler: explicit capture of the launch-bound inner segments.

Static memory: a segment reads its inputs from, and writes its outputs / input gradients to, fixed
buffers.  A producer may write straight into `static_input(i)`; when the tensor handed to the
segment already IS the static buffer no copy is made.  Outputs are overwritten by the next replay
(the train loop consumes them within the step).  Parameter gradients are handed to autograd as
views of static buffers, so `zero_grad(set_to_none=True)` (the default of this package's
optimizers) is required.
"""
from typing import List, Optional, Sequence, Tuple

import torch
from torch import nn


class _Replay(torch.autograd.Function):
    @staticmethod
    def forward(ctx, seg, *flat):
        for i in range(seg.n_inputs):
            s, x = seg.static_inputs[i], flat[i]
            if x.data_ptr() != s.data_ptr():
                s.copy_(x)
        seg.fwd_graph.replay()
        ctx.seg = seg
        ctx.set_materialize_grads(False)  # an unused output (e.g. logits) costs no zero-fill + copy
        outs = tuple(o.detach() for o in seg.static_outputs)
        ctx.mark_non_differentiable(*[o for o, s in zip(outs, seg.static_outputs) if not s.requires_grad])
        return outs

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *grads):
        seg = ctx.seg
        for g, s in zip(grads, seg.static_grad_outputs):
            if s is not None and g is not None and g.data_ptr() != s.data_ptr():
                s.copy_(g)
        seg.bwd_graph.replay()
        if getattr(seg, "bwd_graph2", None) is not None:
            seg.bwd_graph2.replay()  # deferred weight gradients (capture_backward(defer_wgrad=True))
        if getattr(seg, "bwd_graph3", None) is not None:
            seg.bwd_graph3.replay()  # ... the late part of them (late_params)
        if seg.param_grad_sinks is not None:
            # parameter gradients were written (scaled) into the caller's flat buffer inside the graph;
            # autograd gets none, the owner of the buffer is told they are ready
            n_in = seg.n_inputs
            out = (None,) + tuple(g.detach() if g is not None else None for g in seg.static_grad_inputs[:n_in])
            if seg.after_backward is not None:
                seg.after_backward()
            return out + (None,) * (len(seg.static_grad_inputs) - n_in)
        return (None,) + tuple(g.detach() if g is not None else None for g in seg.static_grad_inputs)


class GraphedSegment(nn.Module):
    """`module(*inputs)` with static shapes, replayed from HIP graphs (training mode only)."""

    def __init__(self, module: nn.Module, sample_inputs: Sequence[torch.Tensor],
                 input_buffers: Optional[Sequence[Optional[torch.Tensor]]] = None, warmup: int = 3,
                 pool=None, pre_forward=None) -> None:
        """`pre_forward`: a callable captured at the START of the forward graph (and run in the warm-up) — e.g. the kernel
        that unpacks a persistent receive buffer into one of the static inputs, so that it costs no eager launch.
        Its counterpart is capture_backward(post_backward=)."""
        super().__init__()
        self.module = module
        self._pre_forward = pre_forward
        self.n_inputs = len(sample_inputs)
        input_buffers = list(input_buffers) if input_buffers is not None else [None] * self.n_inputs
        self.static_inputs: List[torch.Tensor] = []
        for x, buf in zip(sample_inputs, input_buffers):
            s = buf if buf is not None else x.detach().clone()
            s = s.detach().requires_grad_(x.requires_grad)
            self.static_inputs.append(s)
        self._params: Tuple[nn.Parameter, ...] = tuple(p for p in module.parameters() if p.requires_grad)
        self._pool = pool if pool is not None else torch.cuda.graph_pool_handle()
        self.fwd_graph = torch.cuda.CUDAGraph()
        self.bwd_graph: Optional[torch.cuda.CUDAGraph] = None
        self.static_grad_outputs: List[Optional[torch.Tensor]] = []
        self.static_grad_inputs: List[Optional[torch.Tensor]] = []
        self.param_grad_sinks: Optional[List[torch.Tensor]] = None
        self.after_backward = None  # callable run right after the backward replay (param_grad_sinks mode)
        dev = self.static_inputs[0].device
        self._stream = torch.cuda.Stream(dev)
        # ---- warm-up (lazy workspaces, GEMM heuristics) on the capture stream ---------------------
        torch.cuda.synchronize(dev)
        with torch.cuda.stream(self._stream):
            for _ in range(warmup):
                outs = self._call()
                need = [o for o in outs if o.requires_grad]
                if need:
                    torch.autograd.grad(need, self._grad_targets(), [torch.ones_like(o) for o in need], allow_unused=True)
            del outs, need
        self._stream.synchronize()
        with torch.cuda.graph(self.fwd_graph, pool=self._pool, stream=self._stream, capture_error_mode="thread_local"):
            self.static_outputs = self._call()
        self._pool_handle = self._pool

    def _call(self) -> Tuple[torch.Tensor, ...]:
        if self._pre_forward is not None:
            with torch.no_grad():
                self._pre_forward()
        out = self.module(*self.static_inputs)
        return tuple(out) if isinstance(out, (tuple, list)) else (out,)

    def _grad_targets(self) -> List[torch.Tensor]:
        return [x for x in self.static_inputs if x.requires_grad] + list(self._params)

    def static_input(self, i: int) -> torch.Tensor:
        return self.static_inputs[i]

    def capture_backward(self, grad_output_buffers: Optional[Sequence[Optional[torch.Tensor]]] = None,
                         param_grad_sinks: Optional[Sequence[torch.Tensor]] = None, sink_scale: float = 1.0,
                         defer_wgrad: bool = False, late_params: int = 0, post_backward=None) -> None:
        """Captures d(outputs)/d(inputs, parameters).  `grad_output_buffers[i]` lets a downstream
        segment's static input-gradient buffer double as this segment's output-gradient buffer.
        `param_grad_sinks[j]` (one per parameter, in `parameters()` order): the graph itself writes
        `sink_scale * d/d(param j)` there — e.g. views of one flat buffer that is all-reduced as a whole —
        and autograd receives no parameter gradients from this segment.
        `defer_wgrad`: the weight-gradient GEMMs of the segment's Linear layers (modules/mlp.py) and the writes into the
        sinks go into a SECOND graph, `bwd_graph2`; `bwd_graph` then ends with the input gradients, and an owner that
        drives the step itself can start what depends on them (the embedding-gradient all-to-all) before it replays the
        second graph.  `_Replay.backward` replays both back to back.
        `late_params` (with defer_wgrad and sinks): the gradients of the FIRST `late_params` parameters (in `parameters()`
        order: the first layers of an MLP) go into a THIRD graph, `bwd_graph3`, the rest stay in `bwd_graph2` — an owner
        can then put one part behind the gradient all-to-all and the other behind the NEXT step's prefetched forward
        all-to-all (models/dlrm.py).  `_Replay.backward` replays all three back to back.
        `post_backward(input_grads)`: a callable captured at the END of the first backward graph, behind the input
        gradients — e.g. the kernel that packs the pooled-embedding gradient into a persistent send buffer."""
        from ..modules.mlp import _DeferredWgrad

        outs = self.static_outputs
        bufs = list(grad_output_buffers) if grad_output_buffers is not None else [None] * len(outs)
        self.static_grad_outputs = [
            (bufs[i] if bufs[i] is not None else torch.zeros_like(o)) if o.requires_grad else None
            for i, o in enumerate(outs)]
        need = [o for o in outs if o.requires_grad]
        targets = self._grad_targets()
        n_in = sum(1 for x in self.static_inputs if x.requires_grad)
        if param_grad_sinks is not None and len(param_grad_sinks) != len(self._params):
            raise ValueError("one gradient sink per parameter")

        def write_sinks(pg_, partials=None, lo: int = 0, hi: Optional[int] = None) -> None:
            """pg_[j]: complete gradient of parameter j or None; partials[j] (optional): [chunks, *shape] whose sum over
            dim 0 is the gradient (split-K slices, row-block sums: modules/mlp.py _DeferredWgrad).  [lo, hi): the
            parameters this call finishes."""
            hi = len(param_grad_sinks) if hi is None else hi
            sinks = list(param_grad_sinks)[lo:hi]
            pg_ = list(pg_)[lo:hi]
            partials = list(partials)[lo:hi] if partials is not None else [None] * len(sinks)
            consecutive = all(s_.is_contiguous() for s_ in sinks) and all(
                sinks[i + 1].data_ptr() == sinks[i].data_ptr() + sinks[i].numel() * sinks[i].element_size()
                for i in range(len(sinks) - 1))
            fp32 = all(s_.dtype == torch.float32 for s_ in sinks) and all(
                (g is None or g.dtype == torch.float32) for g in list(pg_) + list(partials))
            if consecutive and sinks and fp32 and sinks[0].is_cuda:
                # the sinks are consecutive slices of ONE flat buffer: ONE launch sums every parameter's chunks (1 for a
                # complete gradient, 0 = no gradient: zeros) and writes the scaled result — instead of a reduce kernel
                # per split-K layer, a column-sum launch per bias, a cat and a mul (csrc/mlp_epilogue.hip)
                from . import _device_ops  # noqa: F401  (registers torch.ops.tbe_hip.*)
                total = sum(s_.numel() for s_ in sinks)
                flat_slice = torch.as_strided(sinks[0], (total,), (1,), sinks[0].storage_offset())
                rows, keep, off = [], [], 0
                for g, part, sink in zip(pg_, partials, sinks):
                    n = sink.numel()
                    if part is not None and g is not None:  # a parameter used twice: fold the complete part in
                        part = torch.cat([part.reshape(part.shape[0], -1), g.reshape(1, -1)])
                    if part is not None:
                        src = part.contiguous().view(part.shape[0], -1)
                        if src.shape[1] != n:
                            raise RuntimeError("capture_backward: partial gradient does not match its parameter")
                        rows.append([src.data_ptr(), src.shape[0], n, off])
                        keep.append(src)
                    elif g is not None:
                        src = g.contiguous().view(-1)
                        rows.append([src.data_ptr(), 1, n, off])
                        keep.append(src)
                    else:
                        rows.append([flat_slice.data_ptr(), 0, n, off])
                    off += n
                # the segment table was allocated BEFORE the capture, outside the graphs' memory pool: anything persistent
                # inside the pool can sit on addresses an EARLIER graph of the pool uses for its transient tensors and would be
                # overwritten by that graph's every replay (a table placed in the pool produced exactly that: a wild source
                # pointer and a memory aperture violation in the first replayed step).  Its contents — addresses that are
                # final only now — are uploaded ONCE, right after the capture has ended (below).
                table = sink_tables.pop()[:len(rows)]
                torch.ops.tbe_hip.multi_chunk_sum(table, len(rows), max(r[2] for r in rows), flat_slice, float(sink_scale))
                self._sink_tables.append((table, rows, keep))
            else:
                for g, part, sink in zip(pg_, partials, sinks):
                    if part is not None:
                        g = part.sum(dim=0).view_as(sink) if g is None else g + part.sum(dim=0).view_as(sink)
                    if g is None:
                        sink.zero_()
                    else:
                        torch.mul(g, sink_scale, out=sink)
            self.param_grad_sinks = list(param_grad_sinks)

        self.bwd_graph = torch.cuda.CUDAGraph()
        self.bwd_graph2: Optional[torch.cuda.CUDAGraph] = None
        self.bwd_graph3: Optional[torch.cuda.CUDAGraph] = None
        self._sink_tables = []  # (device segment table, its rows, the source tensors kept alive)
        # ordinary allocations: NOT in the graphs' pool (see write_sinks); one per write_sinks call
        sink_tables = ([torch.zeros((len(param_grad_sinks), 4), dtype=torch.int64, device=param_grad_sinks[0].device)
                        for _ in range(2)] if param_grad_sinks else [])
        late_params = int(late_params) if (defer_wgrad and param_grad_sinks is not None) else 0
        if not 0 <= late_params <= len(self._params):
            raise ValueError("late_params out of range")
        if defer_wgrad and self.static_inputs[0].is_cuda:
            # the deferred weight gradients run batched GEMMs (modules/mlp.py compute_partials) that the warm-up never
            # ran: the BLAS handle of this stream must exist before the capture (creating it inside one fails with
            # HIPBLAS_STATUS_INTERNAL_ERROR and invalidates the capture)
            with torch.cuda.stream(self._stream):
                probe = torch.zeros(2, 8, 8, device=self.static_inputs[0].device)
                torch.bmm(probe.transpose(1, 2), probe)
                torch.mm(probe[0].t(), probe[1])
                del probe
        self._stream.synchronize()
        stash = []
        with_partials = bool(defer_wgrad and param_grad_sinks is not None)
        with torch.cuda.graph(self.bwd_graph, pool=self._pool, stream=self._stream, capture_error_mode="thread_local"):
            _DeferredWgrad.pending = [] if defer_wgrad else None
            _DeferredWgrad.partials_ok = with_partials
            try:
                grads = list(torch.autograd.grad(need, targets, [g for g in self.static_grad_outputs if g is not None],
                                                 allow_unused=True))
            finally:
                stash, _DeferredWgrad.pending = (_DeferredWgrad.pending or []), None
                _DeferredWgrad.partials_ok = False
            if param_grad_sinks is not None and not stash:
                write_sinks(grads[n_in:])
            if post_backward is not None:
                with torch.no_grad():
                    post_backward(grads[:n_in])
        if stash:
            index = {id(p): n_in + j for j, p in enumerate(self._params)}

            def finish(graph, lo: int, hi: int) -> None:
                """Captures the stashed weight / bias gradients of parameters [lo, hi) and their write into the sinks."""
                partials = [None] * len(grads)
                with torch.cuda.graph(graph, pool=self._pool, stream=self._stream, capture_error_mode="thread_local"), \
                        torch.no_grad():  # X of a layer is an activation inside the forward's autograd graph
                    for entry in stash:
                        kind, prm = entry[0], entry[1]
                        k = index[id(prm)]
                        if not n_in + lo <= k < n_in + hi:
                            continue
                        if kind == "w":
                            _, _, gy, x, c = entry
                            part = _DeferredWgrad.compute_partials(gy, x, c) if with_partials else None
                            gw = None if with_partials else _DeferredWgrad.compute(gy, x, c)
                        else:
                            part, gw = entry[2], None
                        if part is not None:
                            part = part.reshape(part.shape[0], -1)
                            partials[k] = part if partials[k] is None else torch.cat([partials[k], part])
                        else:
                            grads[k] = gw if grads[k] is None else grads[k] + gw
                    if param_grad_sinks is not None:
                        write_sinks(grads[n_in:], partials[n_in:], lo, hi)
                    else:
                        for k, part in enumerate(partials):
                            if part is not None:  # not reached today (partials are only stashed with sinks); kept consistent
                                full = part.sum(dim=0).view_as(self._params[k - n_in])
                                grads[k] = full if grads[k] is None else grads[k] + full

            self.bwd_graph2 = torch.cuda.CUDAGraph()
            finish(self.bwd_graph2, late_params, len(self._params))
            if late_params:
                self.bwd_graph3 = torch.cuda.CUDAGraph()
                finish(self.bwd_graph3, 0, late_params)
            self._wgrad_stash = stash  # dY / X of the first graph stay allocated: the later graphs read them
        for table, rows, _ in self._sink_tables:  # outside every capture: fill the segment tables the graphs read
            table.copy_(torch.tensor(rows, dtype=torch.int64))
        if self._sink_tables:
            torch.cuda.synchronize(self._sink_tables[0][0].device)
        it = iter(grads)
        self.static_grad_inputs = [next(it) if x.requires_grad else None for x in self.static_inputs]
        self.static_grad_inputs += list(it)  # parameter gradients, in self._params order

    def forward(self, *inputs: torch.Tensor):
        if self.bwd_graph is None:
            raise RuntimeError("GraphedSegment.capture_backward() has not run")
        outs = _Replay.apply(self, *inputs, *self._params)
        return outs if len(outs) > 1 else outs[0]
