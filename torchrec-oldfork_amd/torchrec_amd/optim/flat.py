"""Plain SGD over parameters that live in ONE flat buffer, with their gradients in another.

The dense side of a DLRM has 16 parameters (plus the replicated tiny tables); torch.optim.SGD updates them with a
multi-tensor kernel that still costs 45 us of GPU time and ~60 us of host time per step at the 8-GPU per-rank batch,
where the whole step is 2 ms.  When `DLRMTrain.capture_hip_graphs(flat_grads=True)` has moved the parameters into one
flat buffer (their gradients already arrive in one: models/dlrm.py), the update is a single `flat_p.add_(flat_g, -lr)`.
Same arithmetic per element as torch.optim.SGD without momentum / weight decay (p <- p - lr * g in fp32); a parameter
whose gradient is not the flat view (an eager step the owner did not fold: not the case under TrainPipelineSparseDist)
falls back to the per-parameter update.  The reference builds torch.optim.SGD for these parameters
(examples/dlrm/dlrm_main.py:536-540)."""
from typing import Iterable, List, Optional

import torch


class FlatSGD(torch.optim.Optimizer):
    def __init__(self, params: Iterable[torch.Tensor], lr: float, flat_param: Optional[torch.Tensor] = None,
                 flat_grad: Optional[torch.Tensor] = None, covered: Optional[List[torch.Tensor]] = None,
                 grad_views: Optional[List[torch.Tensor]] = None) -> None:
        super().__init__(list(params), dict(lr=lr))
        self._flat_param, self._flat_grad = flat_param, flat_grad
        self._covered = list(covered) if covered is not None else []
        self._views = list(grad_views) if grad_views is not None else []
        mine = {id(p) for g in self.param_groups for p in g["params"]}
        if any(id(q) not in mine for q in self._covered):
            raise ValueError("FlatSGD: the flat buffer covers parameters this optimizer was not given")
        cov = {id(q) for q in self._covered}
        self._others = [p for g in self.param_groups for p in g["params"] if id(p) not in cov]

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lr = self.param_groups[0]["lr"]
        flat_ok = self._flat_param is not None and all(
            q.grad is not None and q.grad.data_ptr() == v.data_ptr() for q, v in zip(self._covered, self._views))
        if flat_ok:
            self._flat_param.add_(self._flat_grad, alpha=-lr)
            rest = self._others
        else:
            rest = [p for g in self.param_groups for p in g["params"]]
        for p in rest:
            if p.grad is not None:
                p.add_(p.grad, alpha=-lr)
        return loss
