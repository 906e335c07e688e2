"""Perceptron / MLP with the reference's module and parameter names
(torchrec/modules/mlp.py:14-170) so state_dict keys match (`_mlp.{i}._linear.weight`)."""
from typing import Callable, List, Optional, Union

import torch
from torch import nn


class Perceptron(nn.Module):
    def __init__(self, in_size: int, out_size: int, bias: bool = True,
                 activation: Union[nn.Module, Callable[[torch.Tensor], torch.Tensor]] = torch.relu,
                 device: Optional[torch.device] = None) -> None:
        super().__init__()
        self._in_size, self._out_size = in_size, out_size
        self._linear = nn.Linear(in_size, out_size, bias=bias, device=device)
        self._activation_fn = activation

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return self._activation_fn(self._linear(input))


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
