"""MI355X-native drop-in for the ``fbgemm_gpu`` surface that samiwilf/torchrec-oldfork calls.

Importing this package
  * loads the in-tree gfx950 C-ABI library (``include/tbe_hip.h``) — ImportError if it has not
    been built; there is no CPU fallback;
  * registers ``torch.ops.fbgemm.*`` for the HIP dispatch key (``_ops.py``).
Module surface: ``split_table_batched_embeddings_ops`` (TBE modules + enums),
``split_embedding_configs`` (SparseType, EmbOptimType).
"""
import os as _os
import warnings as _warnings


def _require_dmabuf_ipc() -> None:
    """RCCL (backend "nccl") shares device buffers between the rank processes of a node through HIP IPC; on this
    platform's driver only the dmabuf flavour works, which the HSA runtime selects when HSA_ENABLE_IPC_MODE_LEGACY=0
    is in the environment BEFORE it initialises (first HIP call of the process) — without it communicator set-up fails
    with `hipIpcGetMemHandle: invalid argument`.  A user who launches the reference's torchrun line
    (examples/dlrm/README.MD:17-28) on this package must not have to know that: importing the package sets it.  An
    explicit value in the environment wins; if HIP is already up the setting comes too late and we say so."""
    if "HSA_ENABLE_IPC_MODE_LEGACY" in _os.environ:
        return
    _os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    try:
        import torch as _torch

        if _torch.cuda.is_initialized():
            _warnings.warn("fbgemm_gpu (MI355X build): HIP was initialised before this package was imported, so "
                           "HSA_ENABLE_IPC_MODE_LEGACY=0 could not take effect; multi-process RCCL runs on this node need "
                           "it exported before the first HIP call (INTEGRATION.md, launch section)", RuntimeWarning)
    except Exception:  # pragma: no cover - torch always imports here
        pass


_require_dmabuf_ipc()

from . import _lib  # noqa: E402

_lib.load()

from . import _ops  # noqa: E402,F401  (registers torch.ops.fbgemm.*)
from . import split_embedding_configs  # noqa: E402,F401
from . import split_table_batched_embeddings_ops  # noqa: E402,F401

__all__ = ["split_embedding_configs", "split_table_batched_embeddings_ops"]
