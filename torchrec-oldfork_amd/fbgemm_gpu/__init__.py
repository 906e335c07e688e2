"""MI355X-native drop-in for the ``fbgemm_gpu`` surface that samiwilf/torchrec-oldfork calls.

Importing this package
  * loads the in-tree gfx950 C-ABI library (``include/tbe_hip.h``) — ImportError if it has not
    been built; there is no CPU fallback;
  * registers ``torch.ops.fbgemm.*`` for the HIP dispatch key (``_ops.py``).
Module surface: ``split_table_batched_embeddings_ops`` (TBE modules + enums),
``split_embedding_configs`` (SparseType, EmbOptimType).
"""
from . import _lib

_lib.load()

from . import _ops  # noqa: E402,F401  (registers torch.ops.fbgemm.*)
from . import split_embedding_configs  # noqa: E402,F401
from . import split_table_batched_embeddings_ops  # noqa: E402,F401

__all__ = ["split_embedding_configs", "split_table_batched_embeddings_ops"]
