"""What the REFERENCE planner decides for the 26 Criteo-1TB tables on 8 x MI355X, recorded by
importing the reference here (build container only).  Three plans, all through public plug-in points:
  a100_defaults : Topology(8, "cuda") as shipped (planner/constants.py)
  mi355x_topology : Topology(**mi355x_topology_kwargs(8, 8192)), the reference's own estimators
  mi355x_estimator : the same + MI355XPerfEstimator via EmbeddingEnumerator(estimator=...)
plus the row-wise / table-wise shard sizes and offsets the reference computes for every table
(planner/enumerators.py:277-312) at world sizes 2, 4, 8.  Output: tests/golden/planner_criteo_w8.json (data only)."""
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
REFERENCE = "/root/reference"
ROWS = [45833188, 36746, 17245, 7413, 20243, 3, 7114, 1441, 62, 29275261, 1572176, 345138, 10, 2209, 11267, 128, 4, 974, 14,
        48937457, 11316796, 40094537, 452104, 12606, 104, 35]
D = 128


def main():
    import _paths  # noqa: F401
    import _cpu_ops

    pe = types.ModuleType("pyre_extensions")
    pe.none_throws = lambda x, msg=None: x

    class _PS:
        def __init__(self, name):
            self.args = object
            self.kwargs = object

    pe.ParameterSpecification = _PS
    sys.modules["pyre_extensions"] = pe
    _cpu_ops.register()
    sys.path.insert(0, REFERENCE)
    import torch
    from torch import nn
    from torchrec.distributed.embeddingbag import EmbeddingBagCollectionSharder
    from torchrec.distributed.planner import EmbeddingShardingPlanner, Topology
    from torchrec.distributed.planner.enumerators import EmbeddingEnumerator
    from torchrec.distributed.planner.shard_estimators import EmbeddingStorageEstimator
    from torchrec.distributed.planner.types import ParameterConstraints
    from torchrec.modules.embedding_configs import EmbeddingBagConfig
    from torchrec.modules.embedding_modules import EmbeddingBagCollection
    from torchrec_amd.distributed.planner_mi355x import MI355XPerfEstimator, mi355x_topology_kwargs

    class Holder(nn.Module):
        def __init__(self, ebc):
            super().__init__()
            self.ebc = ebc

    def model():
        tables = [EmbeddingBagConfig(name=f"t_cat_{i}", embedding_dim=D, num_embeddings=ROWS[i], feature_names=[f"cat_{i}"])
                  for i in range(len(ROWS))]
        return Holder(EmbeddingBagCollection(tables=tables, device=torch.device("meta")))

    sharders = [EmbeddingBagCollectionSharder(fused_params={"learning_rate": 0.1})]
    # the hot path's sharding types and the fused kernel (BASELINE config 3); the planner chooses among them
    cons = {f"t_cat_{i}": ParameterConstraints(sharding_types=["table_wise", "row_wise", "data_parallel"],
                                               compute_kernels=["batched_fused", "batched_dense"]) for i in range(len(ROWS))}

    def dump(plan):
        out = {}
        for name, p in plan.plan["ebc"].items():
            out[name] = {"sharding_type": p.sharding_type, "compute_kernel": p.compute_kernel, "ranks": p.ranks,
                         "shards": ([{"offsets": list(s.shard_offsets), "sizes": list(s.shard_sizes)} for s in p.sharding_spec.shards]
                                    if p.sharding_spec is not None else None)}
        return out

    result = {"rows": ROWS, "dim": D, "world_size": 8, "batch_size_per_rank": 8192}
    topo_a100 = Topology(world_size=8, compute_device="cuda", batch_size=8192)
    result["a100_defaults"] = dump(EmbeddingShardingPlanner(topology=topo_a100, constraints=cons).plan(model(), sharders))
    kw = mi355x_topology_kwargs(8, 8192)
    result["mi355x_topology_kwargs"] = kw
    topo = Topology(**kw)
    result["mi355x_topology"] = dump(EmbeddingShardingPlanner(topology=topo, constraints=cons).plan(model(), sharders))
    topo2 = Topology(**kw)
    enum = EmbeddingEnumerator(topology=topo2, constraints=cons,
                               estimator=[MI355XPerfEstimator(topo2, cons), EmbeddingStorageEstimator(topology=topo2, constraints=cons)])
    result["mi355x_estimator"] = dump(EmbeddingShardingPlanner(topology=topo2, enumerator=enum, constraints=cons).plan(model(), sharders))
    # shard geometry the reference computes per sharding type (enumerators.py:277-312)
    from torchrec.distributed.planner.enumerators import (_calculate_rw_shard_sizes_and_offsets,
                                                           calculate_shard_sizes_and_offsets)
    from torchrec.distributed.types import ShardingType

    geo = {}
    for W in (2, 3, 4, 8):
        geo[str(W)] = {}
        for i, r in enumerate(ROWS):
            sizes, offs = _calculate_rw_shard_sizes_and_offsets(r, W, D)
            geo[str(W)][f"t_cat_{i}"] = {"row_wise": {"sizes": sizes, "offsets": offs}}
            t = torch.empty((r, D), device="meta")
            s2, o2 = calculate_shard_sizes_and_offsets(t, W, W, ShardingType.TABLE_WISE.value)
            geo[str(W)][f"t_cat_{i}"]["table_wise"] = {"sizes": s2, "offsets": o2}
    result["shard_geometry"] = geo
    with open(os.path.join(sys.argv[1] if len(sys.argv) > 1 else HERE, "planner_criteo_w8.json"), "w") as fh:
        json.dump(result, fh, indent=1, sort_keys=True)
    for k in ("a100_defaults", "mi355x_topology", "mi355x_estimator"):
        kinds = [v["sharding_type"] for v in result[k].values()]
        print(k, {s: kinds.count(s) for s in set(kinds)})


if __name__ == "__main__":
    main()
