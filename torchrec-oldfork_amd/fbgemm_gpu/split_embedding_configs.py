"""``fbgemm_gpu.split_embedding_configs`` names the reference imports
(torchrec/modules/embedding_configs.py:14, torchrec/distributed/batched_embedding_kernel.py:17,
torchrec/distributed/tests/test_fused_optim.py:13)."""
import enum

import torch


@enum.unique
class EmbOptimType(enum.Enum):
    SGD = "sgd"
    EXACT_SGD = "exact_sgd"
    LAMB = "lamb"
    ADAM = "adam"
    EXACT_ADAGRAD = "exact_adagrad"
    EXACT_ROWWISE_ADAGRAD = "exact_row_wise_adagrad"
    LARS_SGD = "lars_sgd"
    PARTIAL_ROWWISE_ADAM = "partial_row_wise_adam"
    PARTIAL_ROWWISE_LAMB = "partial_row_wise_lamb"
    ROWWISE_ADAGRAD = "row_wise_adagrad"
    MADGRAD = "madgrad"

    def __str__(self) -> str:
        return self.value


@enum.unique
class SparseType(enum.Enum):
    FP32 = "fp32"
    FP16 = "fp16"
    INT8 = "int8"
    INT4 = "int4"
    INT2 = "int2"

    def __str__(self) -> str:
        return self.value

    def as_dtype(self) -> torch.dtype:
        return {
            "fp32": torch.float32,
            "fp16": torch.float16,
            "int8": torch.uint8,
            "int4": torch.uint8,
            "int2": torch.uint8,
        }[self.value]

    def bit_rate(self) -> int:
        return {"fp32": 32, "fp16": 16, "int8": 8, "int4": 4, "int2": 2}[self.value]
