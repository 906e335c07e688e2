"""KeyedJaggedTensor / JaggedTensor / KeyedTensor for the MI355X path.

Public surface follows torchrec/sparse/jagged_tensor.py:614-1081 (KeyedJaggedTensor),
:113-360 (JaggedTensor) and :1101-1247 (KeyedTensor): same constructor arguments, same
feature-major layout (`lengths[f * stride + b]`), same method names and results.

What is different by design: host metadata is carried explicitly.  A KJT whose pooling factors
are a host-known constant per key (`fixed_lengths`, e.g. Criteo: 1 id per feature) never needs
the D2H read the reference performs in `sync()` / `length_per_key()`
(jagged_tensor.py:502-509): lengths, offsets, length_per_key and all all-to-all split sizes are
then derived on the host.  Device index arithmetic goes through `torch.ops.fbgemm.*`
(HIP kernels of this repo).
"""
from typing import Dict, List, Optional, Tuple

import torch

import fbgemm_gpu  # noqa: F401  registers torch.ops.fbgemm.* (the reference loads the op library here: jagged_tensor.py:18-24)


def _host_cumsum(x: List[int]) -> List[int]:
    out = [0] * (len(x) + 1)
    for i, v in enumerate(x):
        out[i + 1] = out[i] + v
    return out


def _to_offsets(lengths: torch.Tensor) -> torch.Tensor:
    return torch.ops.fbgemm.asynchronous_complete_cumsum(lengths)


class JaggedTensor:
    """values + lengths/offsets of ONE key (torchrec/sparse/jagged_tensor.py:113-360)."""

    def __init__(self, values: torch.Tensor, weights: Optional[torch.Tensor] = None,
                 lengths: Optional[torch.Tensor] = None, offsets: Optional[torch.Tensor] = None) -> None:
        assert lengths is not None or offsets is not None, "Must provide lengths or offsets"
        self._values, self._weights, self._lengths, self._offsets = values, weights, lengths, offsets

    def values(self) -> torch.Tensor:
        return self._values

    def weights(self) -> torch.Tensor:
        assert self._weights is not None, "This JaggedTensor doesn't have weights."
        return self._weights

    def weights_or_none(self) -> Optional[torch.Tensor]:
        return self._weights

    def lengths(self) -> torch.Tensor:
        if self._lengths is None:
            self._lengths = self._offsets[1:] - self._offsets[:-1]
        return self._lengths

    def offsets(self) -> torch.Tensor:
        if self._offsets is None:
            self._offsets = _to_offsets(self._lengths)
        return self._offsets

    def to_dense(self) -> List[torch.Tensor]:
        offs = self.offsets().tolist()
        return [self._values[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)]


class KeyedJaggedTensor:
    def __init__(
        self,
        keys: List[str],
        values: torch.Tensor,
        weights: Optional[torch.Tensor] = None,
        lengths: Optional[torch.Tensor] = None,
        offsets: Optional[torch.Tensor] = None,
        stride: Optional[int] = None,
        length_per_key: Optional[List[int]] = None,
        offset_per_key: Optional[List[int]] = None,
        index_per_key: Optional[Dict[str, int]] = None,
        jt_dict: Optional[Dict[str, JaggedTensor]] = None,
        fixed_lengths: Optional[List[int]] = None,
    ) -> None:
        self._keys = list(keys)
        self._values = values
        self._weights = weights
        self._lengths = lengths
        self._offsets = offsets
        self._fixed_lengths = list(fixed_lengths) if fixed_lengths is not None else None
        if stride is None:
            if fixed_lengths is not None and len(keys) > 0:
                per_sample = sum(fixed_lengths)
                stride = values.numel() // per_sample if per_sample else 0
            elif len(keys) == 0:
                stride = 0
            elif offsets is not None and offsets.numel() > 0:
                stride = (offsets.numel() - 1) // len(keys)
            elif lengths is not None:
                stride = lengths.numel() // len(keys)
            else:
                stride = 0
        self._stride = int(stride)
        if self._fixed_lengths is not None and length_per_key is None:
            length_per_key = [l * self._stride for l in self._fixed_lengths]
        self._length_per_key = length_per_key
        self._offset_per_key = offset_per_key
        self._index_per_key = index_per_key
        self._jt_dict = jt_dict

    # ---- constructors (jagged_tensor.py:686-760) -------------------------------------------
    @staticmethod
    def from_lengths_sync(keys: List[str], values: torch.Tensor, lengths: torch.Tensor,
                          weights: Optional[torch.Tensor] = None) -> "KeyedJaggedTensor":
        kjt = KeyedJaggedTensor(keys=keys, values=values, weights=weights, lengths=lengths)
        return kjt.sync()

    @staticmethod
    def from_offsets_sync(keys: List[str], values: torch.Tensor, offsets: torch.Tensor,
                          weights: Optional[torch.Tensor] = None) -> "KeyedJaggedTensor":
        kjt = KeyedJaggedTensor(keys=keys, values=values, weights=weights, offsets=offsets)
        return kjt.sync()

    @staticmethod
    def from_fixed_lengths(keys: List[str], values: torch.Tensor, pooling_factors: List[int],
                           weights: Optional[torch.Tensor] = None) -> "KeyedJaggedTensor":
        """Every bag of key k holds exactly pooling_factors[k] ids (Criteo: 1).  All metadata is
        host-side; lengths/offsets tensors are materialised lazily and without a sync."""
        return KeyedJaggedTensor(keys=keys, values=values, weights=weights, fixed_lengths=pooling_factors)

    @staticmethod
    def empty(is_weighted: bool = False, device: Optional[torch.device] = None,
              values_dtype: Optional[torch.dtype] = None) -> "KeyedJaggedTensor":
        return KeyedJaggedTensor(
            keys=[], values=torch.empty(0, dtype=values_dtype or torch.int64, device=device),
            weights=torch.empty(0, device=device) if is_weighted else None,
            lengths=torch.empty(0, dtype=torch.int32, device=device), stride=0)

    # ---- accessors ----------------------------------------------------------------------
    def sync(self) -> "KeyedJaggedTensor":
        self.length_per_key()
        self.offset_per_key()
        return self

    def keys(self) -> List[str]:
        return self._keys

    def values(self) -> torch.Tensor:
        return self._values

    def weights(self) -> torch.Tensor:
        assert self._weights is not None, "This KeyedJaggedTensor doesn't have weights."
        return self._weights

    def weights_or_none(self) -> Optional[torch.Tensor]:
        return self._weights

    def stride(self) -> int:
        return self._stride

    def device(self) -> torch.device:
        return self._values.device

    def fixed_lengths(self) -> Optional[List[int]]:
        return self._fixed_lengths

    def lengths(self) -> torch.Tensor:
        if self._lengths is None:
            if self._fixed_lengths is not None:
                per_key = torch.tensor(self._fixed_lengths, dtype=torch.int32, device=self.device())
                self._lengths = per_key.repeat_interleave(self._stride)
            else:
                assert self._offsets is not None
                self._lengths = self._offsets[1:] - self._offsets[:-1]
        return self._lengths

    def offsets(self) -> torch.Tensor:
        if self._offsets is None:
            if self._fixed_lengths is not None and len(set(self._fixed_lengths)) == 1:
                n = len(self._keys) * self._stride + 1
                self._offsets = torch.arange(n, dtype=torch.int64, device=self.device()) * self._fixed_lengths[0]
            else:
                self._offsets = _to_offsets(self.lengths())
        return self._offsets

    def length_per_key(self) -> List[int]:
        if self._length_per_key is None:
            if len(self._keys) == 0:
                self._length_per_key = []
            elif self._offset_per_key is not None:
                o = self._offset_per_key
                self._length_per_key = [o[i + 1] - o[i] for i in range(len(o) - 1)]
            else:
                # the one unavoidable D2H read for data-dependent lengths
                # (jagged_tensor.py:502-509 does the same)
                self._length_per_key = (
                    self.lengths().view(len(self._keys), -1).sum(dim=1).cpu().tolist())
        return self._length_per_key

    def offset_per_key(self) -> List[int]:
        if self._offset_per_key is None:
            self._offset_per_key = _host_cumsum(self.length_per_key())
        return self._offset_per_key

    def _key_indices(self) -> Dict[str, int]:
        if self._index_per_key is None:
            self._index_per_key = {k: i for i, k in enumerate(self._keys)}
        return self._index_per_key

    # ---- structure ops (jagged_tensor.py:848-994) -----------------------------------------
    def split(self, segments: List[int]) -> List["KeyedJaggedTensor"]:
        out: List[KeyedJaggedTensor] = []
        start = 0
        opk = self.offset_per_key()
        lpk = self.length_per_key()
        for seg in segments:
            end = start + seg
            if seg == len(self._keys):
                out.append(self)
            elif seg == 0:
                out.append(KeyedJaggedTensor(
                    keys=[], values=self._values[:0],
                    weights=None if self._weights is None else self._weights[:0],
                    lengths=torch.empty(0, dtype=torch.int32, device=self.device()), stride=self._stride,
                    fixed_lengths=[] if self._fixed_lengths is not None else None))
            else:
                out.append(KeyedJaggedTensor(
                    keys=self._keys[start:end],
                    values=self._values[opk[start]:opk[end]],
                    weights=None if self._weights is None else self._weights[opk[start]:opk[end]],
                    lengths=None if self._fixed_lengths is not None
                    else self.lengths()[start * self._stride:end * self._stride],
                    stride=self._stride,
                    length_per_key=lpk[start:end],
                    fixed_lengths=self._fixed_lengths[start:end] if self._fixed_lengths is not None else None))
            start = end
        return out

    def permute(self, indices: List[int], indices_tensor: Optional[torch.Tensor] = None) -> "KeyedJaggedTensor":
        if indices_tensor is None:
            indices_tensor = torch.tensor(indices, dtype=torch.int32, device=self.device())
        lpk = self.length_per_key()
        keys = [self._keys[i] for i in indices]
        new_lpk = [lpk[i] for i in indices]
        lengths, values, weights = torch.ops.fbgemm.permute_2D_sparse_data(
            indices_tensor, self.lengths().view(len(self._keys), -1), self._values, self._weights, sum(new_lpk))
        return KeyedJaggedTensor(
            keys=keys, values=values, weights=weights, lengths=lengths.view(-1), stride=self._stride,
            length_per_key=new_lpk if keys else None,
            fixed_lengths=[self._fixed_lengths[i] for i in indices] if self._fixed_lengths is not None else None)

    def __getitem__(self, key: str) -> JaggedTensor:
        opk = self.offset_per_key()
        i = self._key_indices()[key]
        return JaggedTensor(
            values=self._values[opk[i]:opk[i + 1]],
            weights=None if self._weights is None else self._weights[opk[i]:opk[i + 1]],
            lengths=self.lengths()[i * self._stride:(i + 1) * self._stride])

    def to_dict(self) -> Dict[str, JaggedTensor]:
        if self._jt_dict is None:
            self._jt_dict = {k: self[k] for k in self._keys}
        return self._jt_dict

    # ---- device / stream plumbing (jagged_tensor.py:996-1050) -----------------------------
    def to(self, device: torch.device, non_blocking: bool = False) -> "KeyedJaggedTensor":
        mv = lambda t: None if t is None else t.to(device, non_blocking=non_blocking)  # noqa: E731
        return KeyedJaggedTensor(
            keys=self._keys, values=mv(self._values), weights=mv(self._weights), lengths=mv(self._lengths),
            offsets=mv(self._offsets), stride=self._stride, length_per_key=self._length_per_key,
            offset_per_key=self._offset_per_key, index_per_key=self._index_per_key,
            fixed_lengths=self._fixed_lengths)

    def record_stream(self, stream) -> None:
        for t in (self._values, self._weights, self._lengths, self._offsets):
            if t is not None and t.is_cuda:
                t.record_stream(stream)

    def pin_memory(self) -> "KeyedJaggedTensor":
        pm = lambda t: None if t is None else t.pin_memory()  # noqa: E731
        return KeyedJaggedTensor(
            keys=self._keys, values=pm(self._values), weights=pm(self._weights), lengths=pm(self._lengths),
            offsets=pm(self._offsets), stride=self._stride, length_per_key=self._length_per_key,
            offset_per_key=self._offset_per_key, fixed_lengths=self._fixed_lengths)

    def __str__(self) -> str:
        return f"KeyedJaggedTensor(keys={self._keys}, stride={self._stride}, values={tuple(self._values.shape)})"


class KeyedTensor:
    """Dense [B, sum(length_per_key)] tensor with named column blocks
    (torchrec/sparse/jagged_tensor.py:1101-1247)."""

    def __init__(self, keys: List[str], length_per_key: List[int], values: torch.Tensor, key_dim: int = 1) -> None:
        self._keys, self._length_per_key, self._values, self._key_dim = list(keys), list(length_per_key), values, key_dim
        self._offset_per_key = _host_cumsum(self._length_per_key)

    @staticmethod
    def from_tensor_list(keys: List[str], tensors: List[torch.Tensor], key_dim: int = 1, cat_dim: int = 1) -> "KeyedTensor":
        return KeyedTensor(keys, [t.shape[key_dim] for t in tensors], torch.cat(tensors, dim=cat_dim), key_dim)

    def keys(self) -> List[str]:
        return self._keys

    def values(self) -> torch.Tensor:
        return self._values

    def key_dim(self) -> int:
        return self._key_dim

    def length_per_key(self) -> List[int]:
        return self._length_per_key

    def offset_per_key(self) -> List[int]:
        return self._offset_per_key

    def __getitem__(self, key: str) -> torch.Tensor:
        i = self._keys.index(key)
        return self._values.narrow(self._key_dim, self._offset_per_key[i], self._length_per_key[i])

    def to_dict(self) -> Dict[str, torch.Tensor]:
        return {k: self[k] for k in self._keys}

    def record_stream(self, stream) -> None:
        if self._values.is_cuda:
            self._values.record_stream(stream)
