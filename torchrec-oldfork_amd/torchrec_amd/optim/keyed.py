"""Optimizer surface kept drop-in with torchrec/optim/keyed.py (KeyedOptimizer,
CombinedOptimizer, KeyedOptimizerWrapper): what examples/dlrm/dlrm_main.py:536-540 builds."""
from typing import Any, Callable, Dict, List, Mapping, Optional

import torch


class KeyedOptimizerWrapper:
    """params keyed by FQN + a factory for a torch optimizer (optim/keyed.py:310-338)."""

    def __init__(self, params: Mapping[str, torch.Tensor], optim_factory: Callable[[List[torch.Tensor]], torch.optim.Optimizer]) -> None:
        self.params = dict(params)
        self._optimizer = optim_factory(list(self.params.values()))
        self.param_groups = self._optimizer.param_groups
        self.state = self._optimizer.state

    def zero_grad(self, set_to_none: bool = True) -> None:
        self._optimizer.zero_grad(set_to_none=set_to_none)

    def step(self, closure: Any = None) -> None:
        self._optimizer.step(closure)

    def state_dict(self) -> Dict[str, Any]:
        return self._optimizer.state_dict()


class CombinedOptimizer:
    """Steps several (fused and dense) optimizers as one (optim/keyed.py:224-307)."""

    def __init__(self, optims: List[Any]) -> None:
        self._optims = [o for o in optims if o is not None]

    @property
    def optimizers(self) -> List[Any]:
        return self._optims

    @property
    def param_groups(self) -> List[Dict[str, Any]]:
        return [g for o in self._optims for g in o.param_groups]

    def zero_grad(self, set_to_none: bool = True) -> None:
        for o in self._optims:
            o.zero_grad(set_to_none=set_to_none)

    def step(self, closure: Any = None) -> None:
        for o in self._optims:
            o.step(closure)

    def state_dict(self) -> Dict[str, Any]:
        return {str(i): o.state_dict() for i, o in enumerate(self._optims)}
