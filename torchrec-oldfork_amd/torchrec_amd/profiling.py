"""Profiler range labels of the path, with the reference's label strings.

The reference brackets every communication op and every pipeline stage with
`torch.autograd.profiler.record_function("## ... ##")` (torchrec/distributed/dist_data.py:67-388,
comm_ops.py:489-921, train_pipeline.py:120-147, 213, 504-550).  A `record_function` costs ~12 us of
host time per range even when nobody listens (measured on this stack), and a per-rank step of the 8-GPU
configuration is ~1.8 ms with ~20 ranges: always-on labels would be ~13 % of the step.  `label(name)` is therefore

  * a `record_function` while a torch profiler is attached (torch.profiler.profile / autograd.profiler: the same
    labels, the same place in the trace as the reference's) or when forced with mode "on",
  * a roctx range (`roctxRangePush/Pop` through torch.cuda.nvtx, ~0.5 us) in mode "roctx" — what
    `rocprofv3 --marker-trace -- python bench.py ...` collects,
  * a shared no-op context otherwise (one C-level check, ~0.1 us).

Mode: TORCHREC_AMD_PROFILE_LABELS = auto (default) | roctx | on | off, or set_profile_labels().
"""
import os

import torch
from torch.autograd.profiler import record_function

_MODES = ("auto", "roctx", "on", "off")
_mode = os.environ.get("TORCHREC_AMD_PROFILE_LABELS", "auto")
if _mode not in _MODES:
    raise ValueError(f"TORCHREC_AMD_PROFILE_LABELS must be one of {_MODES}")
_profiler_enabled = torch._C._autograd._profiler_enabled


class _Null:
    __slots__ = ()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class _Roctx:
    __slots__ = ("name",)

    def __init__(self, name: str) -> None:
        self.name = name

    def __enter__(self):
        torch.cuda.nvtx.range_push(self.name)
        return self

    def __exit__(self, *exc):
        torch.cuda.nvtx.range_pop()
        return False


_NULL = _Null()


def set_profile_labels(mode: str) -> None:
    global _mode
    if mode not in _MODES:
        raise ValueError(f"mode must be one of {_MODES}")
    _mode = mode


def profile_labels_mode() -> str:
    return _mode


def label(name: str):
    """Context manager for one labelled range (see the module docstring)."""
    m = _mode
    if m == "auto":
        return record_function(name) if _profiler_enabled() else _NULL
    if m == "roctx":
        return _Roctx(name)
    if m == "on":
        return record_function(name)
    return _NULL
