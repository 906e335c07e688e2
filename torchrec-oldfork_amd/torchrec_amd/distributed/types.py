"""Sharding vocabulary (torchrec/distributed/types.py:36-583, embedding_types.py:44-54)."""
import abc
import enum
from dataclasses import dataclass, field
from typing import Dict, Generic, List, Optional, TypeVar

import torch
import torch.distributed as dist

W = TypeVar("W")


class ShardingType(enum.Enum):
    DATA_PARALLEL = "data_parallel"
    TABLE_WISE = "table_wise"
    COLUMN_WISE = "column_wise"
    ROW_WISE = "row_wise"
    TABLE_ROW_WISE = "table_row_wise"
    TABLE_COLUMN_WISE = "table_column_wise"


class EmbeddingComputeKernel(enum.Enum):
    DENSE = "dense"
    SPARSE = "sparse"
    BATCHED_DENSE = "batched_dense"
    BATCHED_FUSED = "batched_fused"
    BATCHED_FUSED_UVM = "batched_fused_uvm"
    BATCHED_FUSED_UVM_CACHING = "batched_fused_uvm_caching"


@dataclass
class ShardMetadata:
    shard_offsets: List[int]
    shard_sizes: List[int]
    placement: str


@dataclass
class ParameterSharding:
    """How one table is sharded (distributed/types.py:300-330)."""

    sharding_type: str
    compute_kernel: str
    ranks: Optional[List[int]] = None
    sharding_spec: Optional[List[ShardMetadata]] = None


@dataclass
class ShardingPlan:
    """module path -> {table name -> ParameterSharding} (distributed/types.py:515-540)."""

    plan: Dict[str, Dict[str, ParameterSharding]] = field(default_factory=dict)

    def get_plan_for_module(self, module_path: str) -> Optional[Dict[str, ParameterSharding]]:
        return self.plan.get(module_path)

    def __str__(self) -> str:
        return "\n".join(f"{m}: " + ", ".join(f"{t}={p.sharding_type}@{p.ranks}" for t, p in tp.items())
                         for m, tp in self.plan.items())


class ShardingEnv:
    """world_size / rank / process_group triple (distributed/types.py:333-360)."""

    def __init__(self, world_size: int, rank: int, pg: Optional[dist.ProcessGroup] = None) -> None:
        self.world_size, self.rank, self.process_group = world_size, rank, pg

    @classmethod
    def from_process_group(cls, pg: Optional[dist.ProcessGroup]) -> "ShardingEnv":
        return cls(dist.get_world_size(pg), dist.get_rank(pg), pg)

    @classmethod
    def from_local(cls, world_size: int, rank: int) -> "ShardingEnv":
        return cls(world_size, rank, None)


class Awaitable(abc.ABC, Generic[W]):
    """Result of an asynchronous step (distributed/types.py:104-147)."""

    @abc.abstractmethod
    def wait(self) -> W:
        ...


class NoWait(Awaitable[W]):
    def __init__(self, obj: W) -> None:
        self._obj = obj

    def wait(self) -> W:
        return self._obj


class LazyAwaitable(Awaitable[W]):
    """Computes on first wait() and caches (distributed/types.py:149-232, without the
    __torch_function__ auto-wait magic: callers of this package wait explicitly)."""

    def __init__(self) -> None:
        self._result: Optional[W] = None
        self._done = False

    @abc.abstractmethod
    def _wait_impl(self) -> W:
        ...

    def wait(self) -> W:
        if not self._done:
            self._result = self._wait_impl()
            self._done = True
        return self._result
