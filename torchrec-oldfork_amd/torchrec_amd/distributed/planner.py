"""Sharding planner with an MI355X hardware model.

Plays the role of EmbeddingShardingPlanner (torchrec/distributed/planner/planners.py:126-309)
for the sharding types on this build's hot path (table-wise, row-wise).  The reference's
perf constants are A100-era (planner/constants.py:14-73: HBM 32 GiB / 897 GB/s, intra-node
600 GB/s); these are MI355X's (288 GB HBM3E, ~6.3 TB/s achievable, 7 xGMI links x ~153 GB/s,
all-to-all uses every link at once).

Cost model.  Per step and per sharded (table-wise) feature the owner sends B_local * D * 4 bytes to
every peer over that peer's own xGMI link; the slowest link decides, so the plan minimises the
MAXIMUM number of (feature, dim) units any rank owns.  Three levers, in this order:
  * data-parallel (replicated) tiny tables: a table of R rows costs an all-reduce of R * D * 4 bytes
    of dense gradient instead of B_local * D * 4 bytes per link and direction; below ~2.5 K rows
    (11 of the 26 Criteo tables, 2.5 MB in total) that is a clear win and removes 42 % of the
    exchanged bytes (the reference offers the same choice as ShardingType.DATA_PARALLEL,
    sharding/dp_sharding.py);
  * table-wise placement of the rest by a greedy longest-processing-time fill
    (planner/partitioners.py:181-197 does this by estimated perf);
  * row-wise only where a table does not fit one GPU's HBM (or on request): with pooled output a
    row-wise feature makes EVERY rank send a partial pool to every peer, W x the bytes of a
    table-wise feature, so it is a capacity tool here, not a balancing tool;
  * host offload where even the row-wise shards do not fit (BASELINE config 4: > 2 TB of rows on a
    node with 8 x 288 GB): the largest tables go to pinned host memory behind the HBM row cache
    (compute kernel `batched_fused_uvm_caching`, embedding_types.py:57-76; HBM cost = caching_ratio x
    shard) until the rest fits.
"""
import os
from dataclasses import dataclass
from typing import Dict, List, Optional

from ..modules.embedding_configs import EmbeddingBagConfig
from .types import EmbeddingComputeKernel, ParameterSharding, ShardingPlan, ShardingType, ShardMetadata

GiB = 1 << 30


@dataclass
class ParameterConstraints:
    """User constraints on one table's plan (planner/types.py:247-258)."""

    sharding_types: Optional[List[str]] = None
    compute_kernels: Optional[List[str]] = None
    caching_ratio: Optional[float] = None  # HBM cache size / table size for batched_fused_uvm_caching


@dataclass
class Topology:
    """planner/types.py:65-108 with MI355X defaults."""

    world_size: int
    compute_device: str = "cuda"
    hbm_cap: int = 288 * 10**9
    hbm_mem_bw: float = 6.3e12
    intra_host_bw: float = 7 * 153e9
    hbm_reserve_fraction: float = 0.15  # activations, workspaces, dense model, allocator slack
    ddr_cap: int = 1024 * 10**9         # host memory per rank for MANAGED* tables (planner/types.py:65-108 ddr_cap)
    caching_ratio: float = 0.2          # planner/constants.py:26


def rw_block_size(rows: int, world_size: int) -> int:
    """ceil(rows / W): row-wise block size (sharding/rw_sharding.py:229-236)."""
    return (rows + world_size - 1) // world_size


def rw_shard_rows(rows: int, world_size: int) -> List[int]:
    """Rows held by each rank, e.g. 10 rows / 3 ranks -> [4, 4, 2]; 5 / 4 -> [2, 2, 1, 0]
    (planner/enumerators.py:277-312)."""
    blk = rw_block_size(rows, world_size)
    return [max(0, min(blk, rows - r * blk)) for r in range(world_size)]


class EmbeddingShardingPlanner:
    def __init__(self, topology: Topology, constraints: Optional[Dict[str, List[str]]] = None,
                 num_row_wise: Optional[int] = None, dp_max_rows: int = 2500) -> None:
        self.topology = topology
        self.constraints = constraints or {}
        self.num_row_wise = num_row_wise
        self.dp_max_rows = dp_max_rows

    def plan_tables(self, tables: List[EmbeddingBagConfig]) -> Dict[str, ParameterSharding]:
        W = self.topology.world_size
        cap = int(self.topology.hbm_cap * (1.0 - self.topology.hbm_reserve_fraction))
        size = {t.name: t.num_embeddings * t.embedding_dim * 4 for t in tables}
        by_size = sorted(tables, key=lambda t: (-size[t.name], t.name))
        st_of = {n: (c.sharding_types if isinstance(c, ParameterConstraints) else c) for n, c in self.constraints.items()}
        ck_of = {n: c.compute_kernels for n, c in self.constraints.items()
                 if isinstance(c, ParameterConstraints) and c.compute_kernels}
        ratio_of = {n: c.caching_ratio for n, c in self.constraints.items()
                    if isinstance(c, ParameterConstraints) and c.caching_ratio}
        forced_rw = {n for n, c in st_of.items() if c == [ShardingType.ROW_WISE.value]}
        forced_tw = {n for n, c in st_of.items() if c == [ShardingType.TABLE_WISE.value]}
        forced_dp = {n for n, c in st_of.items() if c == [ShardingType.DATA_PARALLEL.value]}
        dp = set(forced_dp)
        # TORCHREC_AMD_FORCE_DP=1: rehearsal switch — replicate the tiny tables even on one rank, so that a
        # one-rank run executes the replicated-table path of an N > 1 plan
        if W > 1 or os.environ.get("TORCHREC_AMD_FORCE_DP", "0") == "1":
            for t in tables:
                if t.num_embeddings <= self.dp_max_rows and t.name not in forced_rw and t.name not in forced_tw:
                    dp.add(t.name)
        n_rw = self.num_row_wise if self.num_row_wise is not None else 0
        rw = set(forced_rw)
        for t in by_size:
            if W > 1 and size[t.name] > cap and t.name not in dp:
                rw.add(t.name)  # does not fit one GPU
        for t in by_size:
            if len(rw) >= max(n_rw, len(forced_rw)) or W == 1:
                break
            if t.name not in forced_tw and t.name not in dp:
                rw.add(t.name)
        # compute kernel per table: forced by a constraint, else fused-in-HBM; when the row-wise shards alone
        # exceed a rank's HBM budget the largest row-wise tables move to host memory behind the row cache
        kernel_of: Dict[str, str] = {t.name: EmbeddingComputeKernel.BATCHED_FUSED.value for t in tables}
        for n, ks in ck_of.items():
            if n in kernel_of:
                kernel_of[n] = ks[0]
        uvm = {EmbeddingComputeKernel.BATCHED_FUSED_UVM.value, EmbeddingComputeKernel.BATCHED_FUSED_UVM_CACHING.value}

        def hbm_bytes(t, shard_bytes):
            k = kernel_of[t.name]
            if k == EmbeddingComputeKernel.BATCHED_FUSED_UVM.value:
                return 0
            if k == EmbeddingComputeKernel.BATCHED_FUSED_UVM_CACHING.value:
                return int(shard_bytes * ratio_of.get(t.name, self.topology.caching_ratio))
            return shard_bytes

        def rw_hbm(r):
            return sum(hbm_bytes(t, rw_shard_rows(t.num_embeddings, W)[r] * t.embedding_dim * 4) for t in tables if t.name in rw)

        for t in by_size:  # largest first
            if max(rw_hbm(r) for r in range(W)) <= cap:
                break
            if t.name in rw and kernel_of[t.name] not in uvm and t.name not in ck_of:
                kernel_of[t.name] = EmbeddingComputeKernel.BATCHED_FUSED_UVM_CACHING.value
        # greedy longest-processing-time fill of the table-wise tables
        units = [0.0] * W
        mem = [rw_hbm(r) + sum(size[t.name] for t in tables if t.name in dp) for r in range(W)]
        out: Dict[str, ParameterSharding] = {}
        for t in by_size:
            if t.name in dp:
                out[t.name] = ParameterSharding(
                    ShardingType.DATA_PARALLEL.value, EmbeddingComputeKernel.BATCHED_DENSE.value, list(range(W)),
                    [ShardMetadata([0, 0], [t.num_embeddings, t.embedding_dim], f"rank:{r}/cuda:{r}") for r in range(W)])
                continue
            if t.name in rw:
                rows = rw_shard_rows(t.num_embeddings, W)
                off = 0
                spec = []
                for r in range(W):
                    spec.append(ShardMetadata([off, 0], [rows[r], t.embedding_dim], f"rank:{r}/cuda:{r}"))
                    off += rows[r]
                out[t.name] = ParameterSharding(ShardingType.ROW_WISE.value, kernel_of[t.name], list(range(W)), spec)
                continue
            cost = float(t.num_features() * t.embedding_dim)
            order = sorted(range(W), key=lambda r: (units[r], mem[r], r))
            placed = False
            if W == 1 and kernel_of[t.name] not in uvm and t.name not in ck_of and mem[0] + size[t.name] > cap:
                kernel_of[t.name] = EmbeddingComputeKernel.BATCHED_FUSED_UVM_CACHING.value  # one GPU: offload, not fail
            need = hbm_bytes(t, size[t.name])
            for r in order:
                if mem[r] + need <= cap:
                    units[r] += cost
                    mem[r] += need
                    out[t.name] = ParameterSharding(
                        ShardingType.TABLE_WISE.value, kernel_of[t.name], [r],
                        [ShardMetadata([0, 0], [t.num_embeddings, t.embedding_dim], f"rank:{r}/cuda:{r}")])
                    placed = True
                    break
            if not placed:
                raise RuntimeError(f"planner: table {t.name} ({size[t.name] / GiB:.1f} GiB) does not fit any rank")
        return {t.name: out[t.name] for t in tables}

    def plan(self, module, module_path: str = "") -> ShardingPlan:
        return ShardingPlan({module_path: self.plan_tables(module.embedding_bag_configs)})
