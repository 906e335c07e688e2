"""Perceptron / MLP with the reference's module and parameter names
(torchrec/modules/mlp.py:14-170) so state_dict keys match (`_mlp.{i}._linear.weight`)."""
from typing import Callable, List, Optional, Union

import torch
from torch import nn


class _WgradOverlap:
    """Weight-gradient GEMMs on a side stream (eager steps only).

    dW = dY^T X of a Linear is needed by nobody until the optimizer runs, while the chain the backward waits for —
    dX = dY W, the ReLU-backward pass, the interaction backward, the embedding backward — alternates MFMA-bound GEMMs
    with HBM-bound passes.  With this switch on, `_LinearSplitKWgrad.backward` enqueues dW on a second stream (ordered
    after dY by an event), accumulates it into `weight.grad` there and hands autograd no weight gradient; the GEMM then
    shares the chip with whatever the main stream runs next.  The OWNER of the train loop must call `join()` after
    backward and before the optimizer reads the gradients (TrainPipelineSparseDist does, through
    DLRMTrain.finish_dense_grads) — hence opt-in.  Not for modules wrapped in DistributedDataParallel (its hooks
    never see these gradients) and not inside a HIP-graph capture (the graphed segments keep their own order)."""

    stream: Optional[torch.cuda.Stream] = None
    on: bool = False
    pending: List[torch.cuda.Event] = []

    @classmethod
    def enable(cls, device: torch.device) -> None:
        """Switch on for backward passes started from now on (the caller brackets ONE backward: enable, backward,
        join / disable)."""
        if cls.stream is None or cls.stream.device != device:
            import os

            # numerically larger = lower priority; the runtime clamps to what the device offers
            from fbgemm_gpu._streams import side_stream

            cls.stream = side_stream(device, priority=int(os.environ.get("TORCHREC_AMD_WGRAD_PRIORITY", "0")))
        cls.on = True

    @classmethod
    def disable(cls) -> None:
        cls.join()
        cls.on = False

    @classmethod
    def join(cls) -> None:
        """The current stream waits for every weight gradient enqueued so far."""
        if cls.pending:
            cur = torch.cuda.current_stream()
            for ev in cls.pending:
                cur.wait_event(ev)
            cls.pending = []

    @classmethod
    def active_for(cls, t: torch.Tensor) -> bool:
        return (cls.on and cls.stream is not None and t.is_cuda and t.device == cls.stream.device
                and not torch.cuda.is_current_stream_capturing())


class _DeferredWgrad:
    """Weight gradients set aside while a backward is being CAPTURED (distributed/hip_graph.py capture_backward(defer_wgrad=
    True)): the capture of the input-gradient chain ends first, and the weight-gradient GEMMs are captured into a second
    graph that the owner replays AFTER it has started the embedding-gradient all-to-all — dW is needed by nobody until the
    optimizer, the all-to-all by the embedding backward (models/dlrm.py explicit step).  `pending` is a list while such a
    capture is running, else None; entries are ("w", weight parameter, dY, X, split-K chunks) or — with `partials_ok`, i.e.
    when the segment's parameter gradients go into a flat buffer (param_grad_sinks) — ("p", parameter, partial sums
    [chunks, *parameter shape]): bias gradients as the row-block sums of the ReLU-backward kernel, the one-output layer's
    weight gradient as the row-block sums of its column-sum kernel.  The capture then finishes EVERY gradient of the segment
    — split-K chunk sums, row-block sums, complete gradients — with one `multi_chunk_sum` launch into the flat buffer."""

    pending: Optional[list] = None
    partials_ok: bool = False

    @classmethod
    def compute(cls, gy: torch.Tensor, x: torch.Tensor, c: int) -> torch.Tensor:
        B = x.shape[0]
        if c > 1 and B % c == 0:
            return torch.bmm(gy.view(c, B // c, -1).transpose(1, 2), x.view(c, B // c, -1)).sum(dim=0)
        return gy.t() @ x

    @classmethod
    def compute_partials(cls, gy: torch.Tensor, x: torch.Tensor, c: int) -> torch.Tensor:
        """[chunks, out, in] whose sum over dim 0 is dY^T X (the batched GEMM of `compute` without its reduction)."""
        B = x.shape[0]
        if c > 1 and B % c == 0:
            return torch.bmm(gy.view(c, B // c, -1).transpose(1, 2), x.view(c, B // c, -1))
        return (gy.t() @ x).unsqueeze(0)

    @classmethod
    def stashing(cls, t: torch.Tensor) -> bool:
        return cls.pending is not None and t.is_cuda and torch.cuda.is_current_stream_capturing()


class _DeferredFinish:
    """EAGER steps: finish every dense gradient of one backward with ONE launch.

    The split-K weight gradient of a layer ends in a reduce kernel over its batch slices, the bias gradient in a second-stage
    column-sum launch: 16 launches per DLRM step whose combined work is ~50 MB (≈ 140 us per step at batch 65 536).  With this
    switch on, `backward` returns NO gradient for those parameters and records (parameter, partial sums [chunks, numel]);
    `flush()` adds every parameter's chunks in fixed order with one `multi_chunk_sum` launch (segment table by value in the
    kernel arguments) and attaches the results as `.grad` itself (accumulating into an existing `.grad`), the way
    `_WgradOverlap` does — handing autograd an unfinished tensor does not work: AccumulateGrad clones a gradient somebody
    else still references.  Only for an owner that calls flush() after backward and before anything reads the gradients —
    TrainPipelineSparseDist does, through DLRMTrain.finish_dense_grads — and enables it for the duration of one backward;
    never under DistributedDataParallel (its hooks wait for gradients autograd never delivers) and never inside a HIP-graph
    capture (captures have their own stash: _DeferredWgrad)."""

    on: bool = False
    pending: list = []  # (parameter, partials [chunks, numel] fp32 contiguous)
    MAX_SEGMENTS = 32

    @classmethod
    def enable(cls) -> None:
        cls.on = True

    @classmethod
    def disable(cls) -> None:
        cls.flush()
        cls.on = False

    @classmethod
    def active_for(cls, t: torch.Tensor, param) -> bool:
        return (cls.on and param is not None and t.is_cuda and t.dtype == torch.float32 and param.dtype == torch.float32
                and not torch.cuda.is_current_stream_capturing())

    @classmethod
    def add(cls, param, partials: torch.Tensor) -> None:
        cls.pending.append((param, partials.contiguous().view(partials.shape[0], -1)))
        if len(cls.pending) >= cls.MAX_SEGMENTS:
            cls.flush()

    @classmethod
    def flush(cls) -> None:
        if not cls.pending:
            return
        import ctypes

        from fbgemm_gpu import _lib
        from fbgemm_gpu._lib import check, stream_ptr

        items, cls.pending = cls.pending, []
        dev = items[0][1].device
        table = (ctypes.c_int64 * (4 * len(items)))()
        outs = []
        for i, (prm, part) in enumerate(items):
            if part.shape[1] != prm.numel() or part.device != dev:
                raise RuntimeError("_DeferredFinish: partial sums do not match their parameter")
            dst = torch.empty(prm.shape, dtype=torch.float32, device=dev)
            outs.append(dst)
            table[4 * i:4 * i + 4] = [part.data_ptr(), part.shape[0], dst.numel(), dst.data_ptr() // 4]
        with torch.cuda.device(dev):
            check(_lib.load().tbe_multi_chunk_sum_host_table_f32(table, len(items), max(d.numel() for d in outs), None, 1.0,
                                                                 stream_ptr(dev)), "tbe_multi_chunk_sum_host_table_f32")
        with torch.no_grad():
            for (prm, _), dst in zip(items, outs):
                if prm.grad is None:
                    prm.grad = dst
                else:
                    prm.grad.add_(dst)


class _LinearSplitKWgrad(torch.autograd.Function):
    """y = x W^T + b with the weight gradient computed as a batched GEMM over batch chunks.

    dW = dY^T X has a tiny output (out x in, e.g. 256 x 512) and a huge reduction dimension (the
    batch, 65 536): a plain GEMM launches ~50 workgroups on a 256-CU chip.  Splitting the batch
    into `chunks` slices turns it into one batched GEMM with `chunks` x more workgroups plus a
    small sum — the split-K the BLAS heuristic does not pick for this shape.  fp32 throughout;
    only the summation order of the batch reduction changes."""

    @staticmethod
    def forward(ctx, x, weight, bias, chunks, fuse_relu):
        ctx.chunks = chunks
        ctx.has_bias = bias is not None
        ctx.fuse_relu = fuse_relu
        ctx.weight_param = weight if (weight.is_leaf and weight.requires_grad) else None
        ctx.bias_param = bias if (bias is not None and bias.is_leaf and bias.requires_grad) else None
        if fuse_relu:
            # bias + ReLU in the GEMM epilogue (hipBLASLt): no separate activation kernel
            out = torch._addmm_activation(bias, x, weight.t(), use_gelu=False)
            ctx.save_for_backward(x, weight, out)
            return out
        ctx.save_for_backward(x, weight)
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, gy):
        gb = None
        gb_deferred = False
        if ctx.fuse_relu:
            x, weight, out = ctx.saved_tensors
            if ctx.has_bias and gy.dtype == torch.float32 and out.shape[1] % 4 == 0:
                # ReLU backward + bias gradient in one pass over [B, out] (csrc/mlp_epilogue.hip)
                from ..distributed import _device_ops  # noqa: F401  (registers torch.ops.tbe_hip.*)
                if (_DeferredWgrad.partials_ok and ctx.bias_param is not None and ctx.needs_input_grad[2]
                        and _DeferredWgrad.stashing(gy)):
                    # the row-block sums are finished later, together with every other gradient of the segment
                    gy, part = torch.ops.tbe_hip.relu_backward_bias_partials(gy, out)
                    _DeferredWgrad.pending.append(("p", ctx.bias_param, part))
                    gb_deferred = True
                elif ctx.needs_input_grad[2] and _DeferredFinish.active_for(gy, ctx.bias_param):
                    gy, part = torch.ops.tbe_hip.relu_backward_bias_partials(gy, out)
                    _DeferredFinish.add(ctx.bias_param, part)  # .grad attached by _DeferredFinish.flush()
                    gb_deferred = True
                else:
                    gy, gb = torch.ops.tbe_hip.relu_backward_bias_grad(gy, out)
            else:
                gy = torch.ops.aten.threshold_backward(gy.contiguous(), out, 0.0)
        else:
            x, weight = ctx.saved_tensors
            gy = gy.contiguous()
        gx = None
        if ctx.needs_input_grad[0]:
            if x.dim() == 2 and x.stride(1) == 1 and x.stride(0) > x.shape[1] and x.stride(0) % 4 == 0:
                # the input is a row-padded view (16-B aligned rows, e.g. the 479-wide interaction output inside a 480-wide
                # buffer): give its gradient the same layout, so that the consumer reads aligned rows too
                gx = gy.new_empty((x.shape[0], x.stride(0)))[:, :x.shape[1]]
                torch.mm(gy, weight, out=gx)
            else:
                gx = gy @ weight
        B = x.shape[0]
        c = ctx.chunks

        def wgrad():
            return _DeferredWgrad.compute(gy, x, c)

        w = ctx.weight_param
        if w is not None and ctx.needs_input_grad[1] and _DeferredWgrad.stashing(gy):
            _DeferredWgrad.pending.append(("w", w, gy, x, c))  # captured later, into the segment's second backward graph
            gw = None
        elif w is not None and ctx.needs_input_grad[1] and _WgradOverlap.active_for(gy):
            side, cur = _WgradOverlap.stream, torch.cuda.current_stream()
            ready = torch.cuda.Event()
            ready.record(cur)  # dY (and the step's earlier work on this stream) is complete for the side stream
            side.wait_event(ready)
            with torch.cuda.stream(side):
                gw = wgrad()
                if w.grad is None:
                    w.grad = gw
                else:
                    w.grad.add_(gw)
                done = torch.cuda.Event()
                done.record(side)
            for t in (gy, x):
                t.record_stream(side)  # freed by autograd on `cur` while the side stream may still read them
            gw.record_stream(cur)      # read by the optimizer on the main stream after join()
            _WgradOverlap.pending.append(done)
            gw = None
        elif ctx.needs_input_grad[1] and c > 1 and B % c == 0 and _DeferredFinish.active_for(gy, w):
            # [c, out, in]: the batched GEMM without its reduction; .grad attached by _DeferredFinish.flush()
            _DeferredFinish.add(w, _DeferredWgrad.compute_partials(gy, x, c))
            gw = None
        else:
            gw = wgrad()
        if gb is None and ctx.has_bias and not gb_deferred:
            gb = gy.sum(dim=0)
        return gx, gw, gb, None, None


class _LinearOneOutput(torch.autograd.Function):
    """y = x W^T + b for a Linear with ONE output feature.  Its weight gradient dW[0, c] = sum_b dy[b] x[b, c]
    is a GEMM with N = 1, which the BLAS libraries run at ~1 % of HBM speed (86 us for [65536, 256] with the
    best of all hipBLASLt / rocBLAS solutions); here it is one pass over x (csrc/mlp_epilogue.hip)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.weight_param = weight if (weight.is_leaf and weight.requires_grad) else None
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, gy):
        from ..distributed import _device_ops  # noqa: F401  (registers torch.ops.tbe_hip.*)

        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gy @ weight if ctx.needs_input_grad[0] else None
        if _DeferredWgrad.partials_ok and ctx.weight_param is not None and _DeferredWgrad.stashing(gy):
            part = torch.ops.tbe_hip.weighted_colsum_partials(x, gy.view(-1))  # [row blocks, in]
            _DeferredWgrad.pending.append(("p", ctx.weight_param, part.view(part.shape[0], 1, -1)))
            gw = None
        elif ctx.needs_input_grad[1] and _DeferredFinish.active_for(gy, ctx.weight_param):
            _DeferredFinish.add(ctx.weight_param, torch.ops.tbe_hip.weighted_colsum_partials(x, gy.view(-1)))
            gw = None
        else:
            gw = torch.ops.tbe_hip.weighted_colsum(x, gy.view(-1)).view(1, -1)
        gb = gy.sum(dim=0) if ctx.has_bias else None
        return gx, gw, gb


class LinearOut(nn.Linear):
    """nn.Linear (same parameters / state_dict keys) whose one-output case uses _LinearOneOutput on a HIP device."""

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        if (self.out_features == 1 and input.is_cuda and input.dim() == 2 and input.dtype == torch.float32
                and self.in_features % 4 == 0 and torch.is_grad_enabled() and self.weight.requires_grad):
            return _LinearOneOutput.apply(input, self.weight, self.bias)
        return super().forward(input)


def _wgrad_chunks(batch: int, out_f: int, in_f: int) -> int:
    """Chunks for the split-K weight gradient: enough (128 x 128)-tile workgroups to fill 256 CUs."""
    tiles = max(1, (out_f + 127) // 128) * max(1, (in_f + 127) // 128)
    c = 1
    while tiles * c < 256 and c < 32 and batch % (2 * c) == 0 and batch // (2 * c) >= 1024:
        c *= 2
    return c


class Perceptron(nn.Module):
    def __init__(self, in_size: int, out_size: int, bias: bool = True,
                 activation: Union[nn.Module, Callable[[torch.Tensor], torch.Tensor]] = torch.relu,
                 device: Optional[torch.device] = None) -> None:
        super().__init__()
        self._in_size, self._out_size = in_size, out_size
        self._linear = nn.Linear(in_size, out_size, bias=bias, device=device)
        self._activation_fn = activation

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        lin = self._linear
        if input.is_cuda and input.dim() == 2 and torch.is_grad_enabled() and lin.weight.requires_grad:
            c = _wgrad_chunks(input.shape[0], self._out_size, self._in_size)
            relu = self._activation_fn is torch.relu and lin.bias is not None
            if c > 1 or relu:
                y = _LinearSplitKWgrad.apply(input, lin.weight, lin.bias, c, relu)
                return y if relu else self._activation_fn(y)
        return self._activation_fn(lin(input))


class MLP(nn.Module):
    def __init__(self, in_size: int, layer_sizes: List[int], bias: bool = True,
                 activation: Union[str, Callable] = torch.relu, device: Optional[torch.device] = None) -> None:
        super().__init__()
        if activation == "relu":
            activation = torch.relu
        elif activation == "sigmoid":
            activation = torch.sigmoid
        sizes = [in_size] + list(layer_sizes)
        self._mlp = nn.Sequential(*[
            Perceptron(sizes[i], sizes[i + 1], bias=bias, activation=activation, device=device)
            for i in range(len(layer_sizes))])

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return self._mlp(input)
