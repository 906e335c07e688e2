"""MI355X hardware model for the REFERENCE planner: feed it through the planner's public
plug-in points instead of replacing it —

    from torchrec.distributed.planner import EmbeddingShardingPlanner, Topology
    from torchrec.distributed.planner.enumerators import EmbeddingEnumerator
    from torchrec.distributed.planner.shard_estimators import EmbeddingStorageEstimator
    from torchrec_amd.distributed.planner_mi355x import MI355XPerfEstimator, mi355x_topology_kwargs

    topo = Topology(**mi355x_topology_kwargs(world_size=8, batch_size=8192))
    planner = EmbeddingShardingPlanner(topology=topo, enumerator=EmbeddingEnumerator(
        topology=topo, estimator=[MI355XPerfEstimator(topo), EmbeddingStorageEstimator(topology=topo)]))

`Topology(hbm_cap, intra_host_bw, ...)`: torchrec/distributed/planner/types.py:65-108; `estimator=`:
planner/enumerators.py:46-67.  The reference's own constants are A100-era (planner/constants.py:14-25: HBM 32 GiB,
897 GB/s, intra-node 600 GB/s) and its perf function prices every exchange at the CROSS-node rate (12 GB/s,
shard_estimators.py:228-262), which makes every single-node plan communication-bound on paper.

This module imports nothing from torchrec: the estimator is duck-typed (`estimate(sharding_options, sharder_map)`
setting `shard.perf`, the ShardEstimator protocol of planner/types.py), so it also loads where the reference is absent
(the GPU box).  tests/golden/planner_criteo_w8.json holds what the reference planner decides with it.
"""
from typing import Any, Dict, List, Optional

HBM_BYTES = 288 * 10**9            # HBM3E per GPU
HBM_BYTES_PER_MS = 6.3e12 / 1e3    # achievable (float4 copy); spec 8 TB/s
XGMI_LINK_BYTES_PER_MS = 153e9 / 1e3   # one link, one direction; 7 links per GPU, point to point
XGMI_LINKS = 7
PCIE_BYTES_PER_MS = 63e9 / 1e3     # host link (MANAGED tables)
ID_BYTES = 8


def mi355x_topology_kwargs(world_size: int, batch_size: int = 8192, compute_device: str = "cuda") -> Dict[str, Any]:
    """Keyword arguments for the reference's `Topology` (one node of `world_size` MI355X).  `intra_host_bw` is what
    ONE rank can push into the fabric at once (all its links; an all-to-all uses every link concurrently); there is
    no second node, `inter_host_bw` gets the same figure so that the reference's default estimator, which prices
    exchanges at the inter-host rate, stays meaningful."""
    bw = XGMI_LINKS * XGMI_LINK_BYTES_PER_MS
    return dict(world_size=world_size, compute_device=compute_device, hbm_cap=HBM_BYTES, local_world_size=world_size,
                intra_host_bw=bw, inter_host_bw=bw, batch_size=batch_size)


def kernel_bytes_per_ms(compute_kernel: str, caching_ratio: Optional[float] = None) -> float:
    """Table-read rate of a compute kernel on MI355X (the role of constants.py:28-73 `kernel_bw_lookup`).  Measured
    with this package's kernels: fused-in-HBM ~ the copy rate; host-mapped tables are bound by
    the host link; the row cache sits between the two by its hit rate (~ caching ratio for uniform ids)."""
    r = 0.2 if caching_ratio is None else caching_ratio
    return {
        "dense": 0.35 * HBM_BYTES_PER_MS, "sparse": 0.35 * HBM_BYTES_PER_MS, "batched_dense": 0.5 * HBM_BYTES_PER_MS,
        "batched_fused": HBM_BYTES_PER_MS, "batched_fused_uvm": PCIE_BYTES_PER_MS,
        "batched_fused_uvm_caching": 1.0 / (r / HBM_BYTES_PER_MS + (1.0 - r) / PCIE_BYTES_PER_MS),
    }.get(compute_kernel, HBM_BYTES_PER_MS)


class MI355XPerfEstimator:
    """Per-shard forward wall time in ms on one MI355X node.  Differences from shard_estimators.py:84-262:
      * exchanges run over xGMI: a rank's bytes to one peer cross ONE link (point to point), all peers in parallel,
        so the cost of an exchange is bytes-per-peer / link rate, not total bytes / node rate;
      * table-wise: the owner receives every rank's ids and sends [B_local, D] per feature to each peer;
      * row-wise: every rank holds a shard, looks up B_global * L / W rows and sends a partial pool [B_local, D] per
        feature to each peer — per-link bytes equal a table-wise owner's, but on EVERY rank;
      * data-parallel: no exchange; a dense gradient all-reduce of rows * D * 4 bytes (ring over the links)."""

    def __init__(self, topology: Any, constraints: Optional[Dict[str, Any]] = None) -> None:
        self._topology = topology
        self._constraints = constraints or {}

    def estimate(self, sharding_options: List[Any], sharder_map: Optional[Dict[str, Any]] = None) -> None:
        W = self._topology.world_size
        for so in sharding_options:
            c = self._constraints.get(so.name)
            bw = kernel_bytes_per_ms(so.compute_kernel, getattr(c, "caching_ratio", None) if c is not None else None)
            B_local = so.batch_size
            B_global = B_local * W
            lens = list(so.input_lengths)
            elem = so.tensor.element_size()
            for shard in so.shards:
                rows, dim = shard.size
                per_peer_out = B_local * dim * len(lens) * elem / XGMI_LINK_BYTES_PER_MS if W > 1 else 0.0
                if so.sharding_type == "data_parallel":
                    compute = B_local * sum(lens) * dim * elem / bw
                    comm = 2.0 * rows * dim * elem * (W - 1) / W / XGMI_LINK_BYTES_PER_MS if W > 1 else 0.0
                    shard.perf = compute + comm
                    continue
                if so.sharding_type == "row_wise":
                    lookups = B_global * sum(lens) / W
                    ids_in = B_local * sum(lens) * ID_BYTES / W / XGMI_LINK_BYTES_PER_MS if W > 1 else 0.0
                else:  # table_wise and the column-wise family (a column shard behaves like a narrower table)
                    lookups = B_global * sum(lens)
                    ids_in = B_local * sum(lens) * ID_BYTES / XGMI_LINK_BYTES_PER_MS if W > 1 else 0.0
                shard.perf = ids_in + lookups * dim * elem / bw + per_peer_out
