"""Table configuration types (torchrec/modules/embedding_configs.py:18-133)."""
import enum
from dataclasses import dataclass, field
from math import sqrt
from typing import List, Optional

from fbgemm_gpu.split_embedding_configs import SparseType
from fbgemm_gpu.split_table_batched_embeddings_ops import PoolingMode


@enum.unique
class PoolingType(enum.Enum):
    SUM = "SUM"
    MEAN = "MEAN"
    NONE = "NONE"


@enum.unique
class DataType(enum.Enum):
    FP32 = "FP32"
    FP16 = "FP16"


def pooling_type_to_pooling_mode(p: PoolingType) -> PoolingMode:
    return {PoolingType.SUM: PoolingMode.SUM, PoolingType.MEAN: PoolingMode.MEAN,
            PoolingType.NONE: PoolingMode.NONE}[p]


def data_type_to_sparse_type(d: DataType) -> SparseType:
    return {DataType.FP32: SparseType.FP32, DataType.FP16: SparseType.FP16}[d]


@dataclass
class BaseEmbeddingConfig:
    num_embeddings: int
    embedding_dim: int
    name: str = ""
    data_type: DataType = DataType.FP32
    feature_names: List[str] = field(default_factory=list)
    weight_init_max: Optional[float] = None
    weight_init_min: Optional[float] = None

    def get_weight_init_max(self) -> float:
        # default U(-sqrt(1/N), sqrt(1/N)) — embedding_configs.py:102-112
        return sqrt(1 / self.num_embeddings) if self.weight_init_max is None else self.weight_init_max

    def get_weight_init_min(self) -> float:
        return -sqrt(1 / self.num_embeddings) if self.weight_init_min is None else self.weight_init_min

    def num_features(self) -> int:
        return len(self.feature_names)


@dataclass
class EmbeddingBagConfig(BaseEmbeddingConfig):
    pooling: PoolingType = PoolingType.SUM


@dataclass
class EmbeddingConfig(BaseEmbeddingConfig):
    pass
