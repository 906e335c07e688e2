"""Optimizer surface kept drop-in with torchrec/optim/keyed.py (KeyedOptimizer, CombinedOptimizer,
KeyedOptimizerWrapper): what examples/dlrm/dlrm_main.py:536-540 builds.  state_dict() is KEYED by parameter
name, as the reference's (optim/keyed.py:69-99): {"state": {param_key: {state_name: tensor}}, and
"param_groups" only after save_param_groups(True)}; load_state_dict copies INTO the live state tensors
(optim/keyed.py:104-186), so fused-optimizer state that lives inside a TBE module is restored in place."""
from typing import Any, Callable, Dict, List, Mapping, Tuple, Union

import torch


def _copy_state(dst: Dict[str, Any], src: Mapping[str, Any], where: str) -> None:
    if set(dst.keys()) != set(src.keys()):
        raise ValueError(f"optimizer state of {where}: keys differ: {sorted(dst.keys())} vs {sorted(src.keys())}")
    for k, v in src.items():
        if isinstance(dst[k], torch.Tensor):
            with torch.no_grad():
                dst[k].copy_(v)
        else:
            dst[k] = v


class KeyedOptimizerWrapper:
    """params keyed by FQN + a factory for a torch optimizer (optim/keyed.py:310-338)."""

    def __init__(self, params: Mapping[str, torch.Tensor], optim_factory: Callable[[List[torch.Tensor]], torch.optim.Optimizer]) -> None:
        self.params = dict(params)
        self._optimizer = optim_factory(list(self.params.values()))
        self.param_groups = self._optimizer.param_groups
        self.state = self._optimizer.state
        self._save_param_groups = False

    def zero_grad(self, set_to_none: bool = True) -> None:
        self._optimizer.zero_grad(set_to_none=set_to_none)

    def step(self, closure: Any = None) -> None:
        self._optimizer.step(closure)

    def save_param_groups(self, save: bool) -> None:
        self._save_param_groups = save

    def state_dict(self) -> Dict[str, Any]:
        key_of = {id(p): k for k, p in self.params.items()}
        out: Dict[str, Any] = {"state": {key_of[id(p)]: st for p, st in self.state.items()}}
        if self._save_param_groups:
            out["param_groups"] = [{"params": sorted(key_of[id(p)] for p in g["params"]),
                                    **{k: v for k, v in g.items() if k != "params"}} for g in self.param_groups]
        return out

    def load_state_dict(self, state_dict: Mapping[str, Any]) -> None:
        new = state_dict["state"]
        mine = {k: self.state[p] for k, p in self.params.items() if p in self.state}
        if set(new.keys()) != set(mine.keys()):
            raise ValueError(f"optimizer state keys differ: {sorted(mine.keys())} vs {sorted(new.keys())} "
                             "(run one step first so that the state exists: optim/keyed.py:104-120)")
        for k, st in new.items():
            _copy_state(mine[k], st, k)
        if "param_groups" in state_dict and self._save_param_groups:
            for g, ng in zip(self.param_groups, state_dict["param_groups"]):
                for kk, vv in ng.items():
                    if kk != "params":
                        g[kk] = vv


class CombinedOptimizer:
    """Steps several (fused and dense) optimizers as one (optim/keyed.py:224-307); an entry is an optimizer or a
    (key prefix, optimizer) pair; parameter keys of the combined state_dict are prefix + "." + key."""

    def __init__(self, optims: List[Union[Any, Tuple[str, Any]]]) -> None:
        self._optims: List[Tuple[str, Any]] = []
        for o in optims:
            if o is None:
                continue
            self._optims.append(o if isinstance(o, tuple) else ("", o))

    @staticmethod
    def _key(prefix: str, key: str) -> str:
        return f"{prefix}.{key}" if prefix else key

    @property
    def optimizers(self) -> List[Tuple[str, Any]]:
        return self._optims

    @property
    def params(self) -> Dict[str, torch.Tensor]:
        return {self._key(pre, k): v for pre, o in self._optims for k, v in o.params.items()}

    @property
    def param_groups(self) -> List[Dict[str, Any]]:
        return [g for _, o in self._optims for g in o.param_groups]

    def zero_grad(self, set_to_none: bool = True) -> None:
        for _, o in self._optims:
            o.zero_grad(set_to_none=set_to_none)

    def step(self, closure: Any = None) -> None:
        for _, o in self._optims:
            o.step(closure)

    def save_param_groups(self, save: bool) -> None:
        for _, o in self._optims:
            o.save_param_groups(save)

    def state_dict(self) -> Dict[str, Any]:
        state: Dict[str, Any] = {}
        groups: List[Dict[str, Any]] = []
        have_groups = False
        for pre, o in self._optims:
            sd = o.state_dict()
            for k, v in sd["state"].items():
                state[self._key(pre, k)] = v
            if "param_groups" in sd:
                have_groups = True
                for g in sd["param_groups"]:
                    groups.append({**g, "params": [self._key(pre, k) for k in g["params"]]})
        out: Dict[str, Any] = {"state": state}
        if have_groups:
            out["param_groups"] = groups
        return out

    def load_state_dict(self, state_dict: Mapping[str, Any]) -> None:
        state = state_dict["state"]
        used = set()
        gi = 0
        for pre, o in self._optims:
            mine = o.state_dict()
            sub = {}
            for k in mine["state"].keys():
                full = self._key(pre, k)
                if full not in state:
                    raise ValueError(f"optimizer state for {full} is missing")
                sub[k] = state[full]
                used.add(full)
            sub_sd: Dict[str, Any] = {"state": sub}
            if "param_groups" in mine and "param_groups" in state_dict:
                n = len(mine["param_groups"])
                sub_sd["param_groups"] = state_dict["param_groups"][gi:gi + n]
                gi += n
            o.load_state_dict(sub_sd)
        extra = set(state.keys()) - used
        if extra:
            raise ValueError(f"unexpected optimizer state keys: {sorted(extra)[:5]}")
