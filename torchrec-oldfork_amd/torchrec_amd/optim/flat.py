"""SGD over parameters that live in ONE flat buffer, with their gradients in another.

The dense side of a DLRM has 16 parameters (plus the replicated tiny tables); torch.optim.SGD updates them with a
multi-tensor kernel that still costs 45 us of GPU time and ~60 us of host time per step at the 8-GPU per-rank batch,
where the whole step is 2 ms.  When `DLRMTrain.capture_hip_graphs(flat_grads=True)` has moved the parameters into one
flat buffer (their gradients already arrive in one: models/dlrm.py), the update is a single `flat_p.add_(flat_g, -lr)`.
Same arithmetic per element as torch.optim.SGD (p <- p - lr * g in fp32; with `weight_decay` g <- g + wd * p first, with
`momentum` buf <- momentum * buf + g and p <- p - lr * buf, the first step's buf = g; dampening 0, no Nesterov); a
parameter whose gradient is not the flat view (an eager step the owner did not fold: not the case under
TrainPipelineSparseDist) makes the whole step fall back to per-parameter updates on the same state.  ONE parameter group:
the learning rate, momentum and weight decay of `param_groups[0]` hold for every parameter (add_param_group raises).
The reference builds torch.optim.SGD for these parameters (examples/dlrm/dlrm_main.py:536-540).

Stale gradients: the flat path applies the WHOLE flat gradient buffer, so it is taken only when every covered parameter's
`.grad` IS its view of that buffer this step; a parameter that received no gradient has `.grad is None` (zero_grad with
set_to_none=True, this package's default) — per-parameter path, the parameter is skipped, as torch.optim.SGD skips it —
or a zeroed view (set_to_none=False).  The owner of the buffer (DLRMTrain.finish_dense_grads) zeroes the slices of
parameters without a gradient before it attaches the views, so nothing left over from an earlier step is re-applied.
"""
from typing import Iterable, List, Optional

import torch


class FlatSGD(torch.optim.Optimizer):
    def __init__(self, params: Iterable[torch.Tensor], lr: float, flat_param: Optional[torch.Tensor] = None,
                 flat_grad: Optional[torch.Tensor] = None, covered: Optional[List[torch.Tensor]] = None,
                 grad_views: Optional[List[torch.Tensor]] = None, momentum: float = 0.0, weight_decay: float = 0.0,
                 dampening: float = 0.0, nesterov: bool = False) -> None:
        if dampening != 0.0 or nesterov:
            raise NotImplementedError("FlatSGD: dampening / nesterov are not implemented (use torch.optim.SGD)")
        if lr < 0.0 or momentum < 0.0 or weight_decay < 0.0:
            raise ValueError("FlatSGD: negative lr / momentum / weight_decay")
        params = list(params)
        if params and isinstance(params[0], dict):
            raise ValueError("FlatSGD: one parameter group only (per-group settings would be ignored by the flat update)")
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self._flat_param, self._flat_grad = flat_param, flat_grad
        self._covered = list(covered) if covered is not None else []
        self._views = list(grad_views) if grad_views is not None else []
        if len(self._covered) != len(self._views):
            raise ValueError("FlatSGD: one gradient view per covered parameter")
        if (flat_param is None) != (flat_grad is None) or (flat_param is not None and flat_param.shape != flat_grad.shape):
            raise ValueError("FlatSGD: flat_param and flat_grad must both be given, with one shape")
        mine = {id(p) for g in self.param_groups for p in g["params"]}
        if any(id(q) not in mine for q in self._covered):
            raise ValueError("FlatSGD: the flat buffer covers parameters this optimizer was not given")
        cov = {id(q) for q in self._covered}
        self._others = [p for g in self.param_groups for p in g["params"] if id(p) not in cov]
        # momentum: ONE flat buffer for the covered parameters (the per-parameter fallback uses views of it, so the two
        # paths share their state), one tensor each for the others
        self._flat_buf: Optional[torch.Tensor] = None
        self._buf_views: List[torch.Tensor] = []
        self._other_bufs = {}

    def add_param_group(self, param_group) -> None:
        if getattr(self, "param_groups", None):
            raise ValueError("FlatSGD: one parameter group only")
        super().add_param_group(param_group)

    def _momentum_views(self) -> List[torch.Tensor]:
        if self._flat_buf is None:
            self._flat_buf = torch.zeros_like(self._flat_param)
            base = self._flat_param.data_ptr()
            self._buf_views = []
            for q in self._covered:
                off = (q.data_ptr() - base) // q.element_size()
                if q.data_ptr() < base or off + q.numel() > self._flat_param.numel() or not q.is_contiguous():
                    raise RuntimeError("FlatSGD: a covered parameter does not live in the flat parameter buffer")
                self._buf_views.append(self._flat_buf[off:off + q.numel()].view_as(q))
        return self._buf_views

    @staticmethod
    def _update(p, g, buf, lr: float, momentum: float, wd: float) -> None:
        if wd != 0.0:
            g = g.add(p, alpha=wd)
        if momentum != 0.0:
            buf.mul_(momentum).add_(g)  # buffers start at zero: the first step leaves buf = g, as torch.optim.SGD does
            g = buf
        p.add_(g, alpha=-lr)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if len(self.param_groups) != 1:
            raise RuntimeError("FlatSGD: one parameter group only")
        grp = self.param_groups[0]
        lr, momentum, wd = grp["lr"], grp["momentum"], grp["weight_decay"]
        flat_ok = self._flat_param is not None and all(
            q.grad is not None and q.grad.data_ptr() == v.data_ptr() for q, v in zip(self._covered, self._views))
        if flat_ok:
            if momentum != 0.0:
                self._momentum_views()
            self._update(self._flat_param, self._flat_grad, self._flat_buf, lr, momentum, wd)
        else:
            bufs = self._momentum_views() if (momentum != 0.0 and self._flat_param is not None) else [None] * len(self._covered)
            for p, b in zip(self._covered, bufs):
                if p.grad is not None:
                    self._update(p, p.grad, b, lr, momentum, wd)
        for p in self._others:
            if p.grad is None:
                continue
            b = None
            if momentum != 0.0:
                b = self._other_bufs.get(id(p))
                if b is None:
                    b = self._other_bufs[id(p)] = torch.zeros_like(p)
            self._update(p, p.grad, b, lr, momentum, wd)
        return loss
