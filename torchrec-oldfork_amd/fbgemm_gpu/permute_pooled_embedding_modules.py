"""Name required at import time by torchrec/distributed/sharding/cw_sharding.py:12.
Column-wise sharding is outside this build's hot path."""
from typing import List, Optional

import torch


class PermutePooledEmbeddings:
    def __init__(self, embs_dims: List[int], permute: List[int],
                 device: Optional[torch.device] = None) -> None:
        raise NotImplementedError(
            "PermutePooledEmbeddings (column-wise sharding) is out of scope of the MI355X hot path"
        )
