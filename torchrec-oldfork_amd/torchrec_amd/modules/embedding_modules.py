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


class EmbeddingCollection(nn.Module):
    """Unpooled ("sequence") embeddings: KeyedJaggedTensor [F x B x L] -> Dict[feature, JaggedTensor]
    whose values are [sum of lengths, D] (torchrec/modules/embedding_modules.py:204-337; consumed by
    examples/bert4rec/models/bert4rec.py:380-408 through fbgemm.jagged_2d_to_dense).

    On a HIP device one TBE launch with PoolingMode.NONE serves every feature
    (tbe_forward_nobag_f32); weights are ordinary parameters of a
    DenseTableBatchedEmbeddingBagsCodegen, or — with `fused_params` — a
    SplitTableBatchedEmbeddingBagsCodegen whose optimizer runs inside backward
    (examples/bert4rec/bert4rec_main.py:488-491 uses fused ADAM)."""

    def __init__(self, tables, device: Optional[torch.device] = None, need_indices: bool = False,
                 fused_params: Optional[dict] = None) -> None:
        super().__init__()
        from fbgemm_gpu.split_table_batched_embeddings_ops import (
            ComputeDevice, DenseTableBatchedEmbeddingBagsCodegen, EmbeddingLocation, PoolingMode,
            SplitTableBatchedEmbeddingBagsCodegen)

        self._embedding_configs = list(tables)
        self._need_indices = need_indices
        dims = {c.embedding_dim for c in tables}
        if len(dims) != 1:
            raise ValueError("All tables in a EmbeddingCollection are required to have same embedding dimension.")
        self.embedding_dim = dims.pop()
        self._feature_names: List[str] = []
        ftm: List[int] = []
        for t, cfg in enumerate(tables):
            if not cfg.feature_names:
                cfg.feature_names = [cfg.name]
            for f in cfg.feature_names:
                self._feature_names.append(f)
                ftm.append(t)
        self._device = torch.device(device) if device is not None else torch.device("cpu")
        if self._device.type != "cuda":
            raise RuntimeError("EmbeddingCollection: only a HIP device is supported (no CPU compute path)")
        with torch.cuda.device(self._device):
            if fused_params is not None:
                self._tbe = SplitTableBatchedEmbeddingBagsCodegen(
                    [(c.num_embeddings, c.embedding_dim, EmbeddingLocation.DEVICE, ComputeDevice.CUDA) for c in tables],
                    feature_table_map=ftm, pooling_mode=PoolingMode.NONE, device=self._device, **fused_params)
            else:
                self._tbe = DenseTableBatchedEmbeddingBagsCodegen(
                    [(c.num_embeddings, c.embedding_dim) for c in tables], feature_table_map=ftm,
                    pooling_mode=PoolingMode.NONE)
        for cfg, w in zip(tables, self._tbe.split_embedding_weights()):
            w.uniform_(cfg.get_weight_init_min(), cfg.get_weight_init_max())

    @property
    def embedding_configs(self):
        return self._embedding_configs

    def table_weights(self) -> Dict[str, torch.Tensor]:
        return {cfg.name: w for cfg, w in zip(self._embedding_configs, self._tbe.split_embedding_weights())}

    def forward(self, features: KeyedJaggedTensor) -> Dict[str, "JaggedTensor"]:
        from ..sparse.jagged_tensor import JaggedTensor

        if features.keys() != self._feature_names:
            features = features.permute([features.keys().index(k) for k in self._feature_names])
        emb = self._tbe(features.values(), features.offsets())  # [N, D], feature-major like the ids
        opk = features.offset_per_key()
        B = features.stride()
        lengths = features.lengths()
        out: Dict[str, JaggedTensor] = {}
        for i, name in enumerate(self._feature_names):
            out[name] = JaggedTensor(
                values=emb[opk[i]:opk[i + 1]], lengths=lengths[i * B:(i + 1) * B],
                weights=features.values()[opk[i]:opk[i + 1]] if self._need_indices else None)
        return out
