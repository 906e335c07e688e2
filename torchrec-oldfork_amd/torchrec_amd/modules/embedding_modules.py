"""EmbeddingBagCollection for the MI355X path.

Same constructor and call contract as the reference module
(torchrec/modules/embedding_modules.py:41-193): `tables: List[EmbeddingBagConfig]`,
`forward(KeyedJaggedTensor) -> KeyedTensor` with keys = feature names in table order and
values `[B, sum(dim)]`.  On device="meta" it only carries the configs (this is how
DistributedModelParallel receives it, examples/dlrm/dlrm_main.py:498-518).  On a HIP device the
unsharded forward runs the TBE kernels through a dense-gradient
DenseTableBatchedEmbeddingBagsCodegen (weights are ordinary parameters, optimizer external).
The reference's CPU design (a python loop of nn.EmbeddingBag) is NOT reproduced here — it lives
in oracle/ebc_torch.py as the CPU baseline.
"""
from typing import Dict, List, Optional

import torch
from torch import nn

from ..sparse.jagged_tensor import KeyedJaggedTensor, KeyedTensor
from .embedding_configs import EmbeddingBagConfig, pooling_type_to_pooling_mode


class EmbeddingBagCollection(nn.Module):
    def __init__(self, tables: List[EmbeddingBagConfig], is_weighted: bool = False,
                 device: Optional[torch.device] = None) -> None:
        super().__init__()
        self._is_weighted = is_weighted
        self._embedding_bag_configs = list(tables)
        names = set()
        self._feature_names: List[str] = []
        self._lengths_per_embedding: List[int] = []
        ftm: List[int] = []
        for t, cfg in enumerate(tables):
            if cfg.name in names:
                raise ValueError(f"Duplicate table name {cfg.name}")
            names.add(cfg.name)
            if not cfg.feature_names:
                cfg.feature_names = [cfg.name]
            for f in cfg.feature_names:
                self._feature_names.append(f)
                self._lengths_per_embedding.append(cfg.embedding_dim)
                ftm.append(t)
        self._device = torch.device(device) if device is not None else torch.device("cpu")
        self._tbe = None
        if self._device.type == "cuda":
            from fbgemm_gpu.split_table_batched_embeddings_ops import DenseTableBatchedEmbeddingBagsCodegen

            poolings = {cfg.pooling for cfg in tables}
            if len(poolings) != 1:
                raise ValueError("all tables of an unsharded collection must share one pooling type")
            with torch.cuda.device(self._device):
                self._tbe = DenseTableBatchedEmbeddingBagsCodegen(
                    [(cfg.num_embeddings, cfg.embedding_dim) for cfg in tables], feature_table_map=ftm,
                    pooling_mode=pooling_type_to_pooling_mode(tables[0].pooling))
            for cfg, w in zip(tables, self._tbe.split_embedding_weights()):
                w.uniform_(cfg.get_weight_init_min(), cfg.get_weight_init_max())
        elif self._device.type != "meta":
            raise RuntimeError(
                "EmbeddingBagCollection: only device='meta' (to be sharded) or a HIP device is supported; "
                "there is no CPU compute path in this package")

    @property
    def embedding_bag_configs(self) -> List[EmbeddingBagConfig]:
        return self._embedding_bag_configs

    @property
    def is_weighted(self) -> bool:
        return self._is_weighted

    def feature_names(self) -> List[str]:
        return self._feature_names

    def table_weights(self) -> Dict[str, torch.Tensor]:
        assert self._tbe is not None
        return {cfg.name: w for cfg, w in zip(self._embedding_bag_configs, self._tbe.split_embedding_weights())}

    def forward(self, features: KeyedJaggedTensor) -> KeyedTensor:
        if self._tbe is None:
            raise RuntimeError("EmbeddingBagCollection on device 'meta' must be sharded before use")
        if features.keys() != self._feature_names:
            order = [features.keys().index(k) for k in self._feature_names]
            features = features.permute(order)
        out = self._tbe(features.values(), features.offsets(),
                        features.weights_or_none() if self._is_weighted else None)
        return KeyedTensor(keys=self._feature_names, length_per_key=self._lengths_per_embedding, values=out)
