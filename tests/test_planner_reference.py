"""This package's planner against what the REFERENCE planner decides (tests/golden/planner_criteo_w8.json, made by
tests/golden/make_planner_golden.py from the imported reference: 26 Criteo-1TB tables, 8 ranks, batch 8192 per rank,
sharding types {table_wise, row_wise, data_parallel}) — with its shipped A100 constants, with MI355X numbers through
`Topology(...)`, and with `MI355XPerfEstimator` plugged into `EmbeddingEnumerator(estimator=...)`
(torchrec/distributed/planner/types.py:65-108, enumerators.py:46-67, 277-312, constants.py:14-73)."""
import json
import os

import pytest

import _paths  # noqa: F401

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "planner_criteo_w8.json")


@pytest.fixture(scope="module")
def ref():
    return json.load(open(GOLD))


@pytest.mark.parametrize("W", [2, 3, 4, 8])
def test_row_wise_and_table_wise_shard_geometry_equals_reference(ref, W):
    """Shard sizes / offsets of every Criteo table (enumerators.py:277-312: block = ceil(rows / W), last shards short or
    empty) — what `rw_shard_rows` / `rw_block_size` and the planner's ShardMetadata must reproduce."""
    from torchrec_amd.distributed.planner import EmbeddingShardingPlanner, ParameterConstraints, Topology, rw_block_size, rw_shard_rows
    from torchrec_amd.modules.embedding_configs import EmbeddingBagConfig

    rows, D = ref["rows"], ref["dim"]
    tables = [EmbeddingBagConfig(name=f"t_cat_{i}", embedding_dim=D, num_embeddings=rows[i], feature_names=[f"cat_{i}"])
              for i in range(len(rows))]
    for kind in ("row_wise", "table_wise"):
        cons = {t.name: ParameterConstraints(sharding_types=[kind]) for t in tables}
        plan = EmbeddingShardingPlanner(Topology(W), constraints=cons, dp_max_rows=0).plan_tables(tables)
        for i, t in enumerate(tables):
            g = ref["shard_geometry"][str(W)][t.name][kind]
            assert plan[t.name].sharding_type == kind
            assert [s.shard_sizes for s in plan[t.name].sharding_spec] == g["sizes"], (t.name, kind)
            assert [s.shard_offsets for s in plan[t.name].sharding_spec] == g["offsets"], (t.name, kind)
            if kind == "row_wise":
                assert rw_shard_rows(rows[i], W) == [s[0] for s in g["sizes"]]
                assert [r * rw_block_size(rows[i], W) for r in range(W)][:1] == [g["offsets"][0][0]]


def test_own_plan_against_the_reference_planner_with_the_mi355x_estimator(ref):
    """Same inputs, both planners.  What must agree: no table needs row-wise sharding at 288 GB per GPU (capacity is the
    only reason either planner has for it: a row-wise feature costs every rank a partial-pool exchange), the big tables
    are table-wise, the tiniest are replicated.  Deliberate differences: this package replicates up to
    2500 rows (the reference + MI355X estimator stops at 128: its data-parallel cost adds a ring all-reduce of the dense
    gradient, this build's tiny tables ride the flat dense-gradient all-reduce that exists anyway) and fills ranks by
    longest-processing-time, so no rank owns more than 2 tables where the reference's greedy storage-sorted fill gives
    rank 0 and 1 three."""
    from torchrec_amd.distributed.planner import EmbeddingShardingPlanner, Topology
    from torchrec_amd.modules.embedding_configs import EmbeddingBagConfig

    rows, D = ref["rows"], ref["dim"]
    tables = [EmbeddingBagConfig(name=f"t_cat_{i}", embedding_dim=D, num_embeddings=rows[i], feature_names=[f"cat_{i}"])
              for i in range(len(rows))]
    mine = EmbeddingShardingPlanner(Topology(8)).plan_tables(tables)
    theirs = ref["mi355x_estimator"]
    kinds = lambda plan, get: {k: [n for n in plan if get(plan[n]) == k] for k in ("table_wise", "row_wise", "data_parallel")}  # noqa: E731
    m, t = kinds(mine, lambda p: p.sharding_type), kinds(theirs, lambda p: p["sharding_type"])
    assert m["row_wise"] == [] and t["row_wise"] == []
    assert set(t["data_parallel"]) <= set(m["data_parallel"])            # their 8 tables of <= 128 rows are replicated here too
    assert all(rows[int(n.split("_")[-1])] <= 2500 for n in m["data_parallel"]) and len(m["data_parallel"]) == 11
    assert len(t["data_parallel"]) == 8 and max(rows[int(n.split("_")[-1])] for n in t["data_parallel"]) == 128
    big = {f"t_cat_{i}" for i, r in enumerate(rows) if r > 2500}
    assert big <= set(m["table_wise"]) and big <= set(t["table_wise"])
    per_rank = lambda names, rank_of: [sum(1 for n in names if rank_of(n) == r) for r in range(8)]  # noqa: E731
    mine_load = per_rank(m["table_wise"], lambda n: mine[n].ranks[0])
    their_load = per_rank(t["table_wise"], lambda n: theirs[n]["ranks"][0])
    assert sum(mine_load) == 15 and max(mine_load) == 2
    assert sum(their_load) == 18 and max(their_load) == 3
    # the four largest tables sit on four different ranks in both plans
    top4 = [f"t_cat_{i}" for i in sorted(range(26), key=lambda i: -rows[i])[:4]]
    assert len({mine[n].ranks[0] for n in top4}) == 4 and len({theirs[n]["ranks"][0] for n in top4}) == 4


def test_reference_defaults_would_not_fit_the_machine_model(ref):
    """Why the hardware model matters: with the shipped A100 constants (32 GiB HBM, exchanges priced at 12 GB/s) the
    reference replicates 13 tables incl. two of > 7000 rows; with MI355X capacity / bandwidth through Topology alone
    (its own estimator) it replicates none and puts four tables on one rank."""
    a = [v["sharding_type"] for v in ref["a100_defaults"].values()]
    b = [v["sharding_type"] for v in ref["mi355x_topology"].values()]
    assert a.count("data_parallel") == 13 and a.count("table_wise") == 13
    assert b.count("table_wise") == 26
    assert ref["mi355x_topology_kwargs"]["hbm_cap"] == 288 * 10**9


def test_estimator_module_needs_no_reference():
    """planner_mi355x imports nothing from torchrec: usable where the reference is absent; duck-typed protocol."""
    import types

    from torchrec_amd.distributed.planner_mi355x import MI355XPerfEstimator, kernel_bytes_per_ms, mi355x_topology_kwargs

    kw = mi355x_topology_kwargs(8)
    assert kw["world_size"] == 8 and kw["hbm_cap"] == 288 * 10**9 and kw["intra_host_bw"] == pytest.approx(7 * 153e6)
    assert kernel_bytes_per_ms("batched_fused") > kernel_bytes_per_ms("batched_fused_uvm_caching") > kernel_bytes_per_ms("batched_fused_uvm")
    shard = types.SimpleNamespace(size=[1000, 128], perf=0)
    so = types.SimpleNamespace(name="t", compute_kernel="batched_fused", sharding_type="table_wise", batch_size=8192,
                               input_lengths=[1.0], tensor=types.SimpleNamespace(element_size=lambda: 4), shards=[shard])
    MI355XPerfEstimator(types.SimpleNamespace(world_size=8)).estimate([so])
    tw = shard.perf
    so.sharding_type = "row_wise"
    MI355XPerfEstimator(types.SimpleNamespace(world_size=8)).estimate([so])
    assert 0 < shard.perf < tw  # a row-wise SHARD is cheaper than the table-wise table, but all 8 ranks pay it
