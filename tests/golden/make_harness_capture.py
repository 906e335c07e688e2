"""Harness capture (last bullet): what the REFERENCE's sharded EmbeddingBagCollection hands to the
TBE class, recorded by importing the reference here (build container only; the reference does not travel) and
running planner + DistributedModelParallel + ShardedEmbeddingBagCollection on gloo, world sizes 1 and 2.

Two runs per world size, same tables / plan / weights / batches:
  * compute kernel `batched_fused` with a RECORDING stand-in for
    `fbgemm_gpu.split_table_batched_embeddings_ops.SplitTableBatchedEmbeddingBagsCodegen`
    (torchrec/distributed/batched_embedding_kernel.py:629-640): records the constructor arguments exactly as
    passed, every forward's (indices, offsets, per_sample_weights), the state_dict keys
    (embeddingbag.py:405-416) and the fused optimizer's parameter / state keys (batched_embedding_kernel.py:53-257);
  * compute kernel `dense` (embedding_kernel.py GroupedEmbeddingBag = nn.EmbeddingBag, pure torch): the lookup
    output the reference itself computes for the same ids ([B_global, sum D_local] per rank) and the final
    KeyedTensor (keys, length_per_key, values) after the output dist.

Output: tests/golden/harness_w{1,2}.json (metadata) + harness_w{1,2}.npz (arrays).  Data only.

Note: this container has no GPU, so the reference runs with device "cpu" and therefore passes
EmbeddingLocation.HOST / ComputeDevice.CPU in `embedding_specs` (batched_embedding_kernel.py:612-620 picks
DEVICE / CUDA when device.type == "cuda"); everything else in the capture is device independent.
"""
import json
import os
import sys
import types

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
REFERENCE = "/root/reference"

ROWS = [120, 7, 64, 33]
DIMS = [16, 16, 8, 16]
FEATURES = [["f0"], ["f1"], ["f2", "f2b"], ["f3"]]  # table t2 serves two features
SHARDING = {1: ["table_wise"] * 4, 2: ["table_wise", "row_wise", "table_wise", "row_wise"]}
B_LOCAL = 5
STEPS = 2
LR = 0.1


def _import_reference():
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
    import torchrec  # noqa: F401
    return torchrec


def _enum_name(x):
    return f"{type(x).__name__}.{x.name}" if hasattr(x, "name") and hasattr(type(x), "__members__") else x


def _plain(x):
    """ctor arguments -> JSON: enums by name, tuples as lists, devices as strings."""
    if isinstance(x, (list, tuple)):
        return [_plain(v) for v in x]
    if isinstance(x, dict):
        return {k: _plain(v) for k, v in x.items()}
    if isinstance(x, torch.device):
        return str(x)
    x = _enum_name(x)
    if isinstance(x, (int, float, str, bool)) or x is None:
        return x
    return repr(x)


class RecordingTBE(nn.Module):
    """Stand-in for the TBE class: records what it is given; computes with torch so that the step completes."""

    instances = []

    def __init__(self, embedding_specs, feature_table_map=None, pooling_mode=None, **kwargs):
        super().__init__()
        self.ctor = {"embedding_specs": _plain(embedding_specs), "feature_table_map": _plain(feature_table_map),
                     "pooling_mode": _plain(pooling_mode), **{k: _plain(v) for k, v in kwargs.items()}}
        self.ctor_kwarg_order = ["embedding_specs", "feature_table_map", "pooling_mode"] + list(kwargs.keys())
        self.specs = [(int(s[0]), int(s[1])) for s in embedding_specs]
        self.ftm = list(feature_table_map) if feature_table_map is not None else list(range(len(self.specs)))
        self.weights_list = [torch.zeros(r, d) for r, d in self.specs]
        self.optimizer_args = types.SimpleNamespace(learning_rate=float(kwargs.get("learning_rate", 0.01)))
        self.calls = []
        RecordingTBE.instances.append(self)

    def split_embedding_weights(self):
        return self.weights_list

    def split_optimizer_states(self):
        # EXACT_SGD keeps no state; row-wise Adagrad one float per row (asserted at batched_embedding_kernel.py:146-148)
        if "ROWWISE_ADAGRAD" in str(self.ctor.get("optimizer", "")):
            return [(torch.zeros(r),) for r, _ in self.specs]
        return [() for _ in self.specs]

    def set_learning_rate(self, lr):
        self.optimizer_args.learning_rate = lr

    def flush(self):
        pass

    def forward(self, indices, offsets, per_sample_weights=None):
        self.calls.append((indices.detach().clone(), offsets.detach().clone(),
                           None if per_sample_weights is None else per_sample_weights.detach().clone()))
        F = len(self.ftm)
        B = (offsets.numel() - 1) // F
        outs = []
        for f, t in enumerate(self.ftm):
            o = offsets[f * B:(f + 1) * B + 1]
            s, e = int(o[0]), int(o[-1])
            outs.append(torch.nn.functional.embedding_bag(indices[s:e], self.weights_list[t], o - o[0], mode="sum",
                                                          include_last_offset=True))
        return torch.cat(outs, dim=1)


def _global_weights():
    g = torch.Generator()
    g.manual_seed(1234)
    return [torch.randn(r, d, generator=g) for r, d in zip(ROWS, DIMS)]


def _batches(W):
    rng = np.random.default_rng(99)
    keys = [k for fs in FEATURES for k in fs]
    table_of = [t for t, fs in enumerate(FEATURES) for _ in fs]
    out = []
    for _ in range(STEPS):
        per_rank = []
        for _r in range(W):
            lengths = rng.integers(0, 4, size=len(keys) * B_LOCAL).astype(np.int32)
            vals = np.concatenate([rng.integers(0, ROWS[table_of[f]], size=int(lengths[f * B_LOCAL:(f + 1) * B_LOCAL].sum()))
                                   for f in range(len(keys))]).astype(np.int64)
            per_rank.append((lengths, vals))
        out.append(per_rank)
    return keys, out


def _fill_weights(state_dict, glob):
    """Writes rows of the seeded global tables into the local shards (as the reference's own test does:
    test_utils/test_model_parallel_base.py:92-122)."""
    from torch.distributed._shard.sharded_tensor import ShardedTensor

    for name, tensor in state_dict.items():
        t = int(name.split("embedding_bags.t")[1].split(".")[0])
        if isinstance(tensor, ShardedTensor):
            for sh in tensor.local_shards():
                r0, c0 = sh.metadata.shard_offsets
                sh.tensor.copy_(glob[t][r0:r0 + sh.tensor.shape[0], c0:c0 + sh.tensor.shape[1]])
        else:
            tensor.copy_(glob[t])


def _worker(rank, W, port, outdir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["GLOO_DEVICE_TRANSPORT"] = "TCP"
    _import_reference()
    dist.init_process_group("gloo", rank=rank, world_size=W)
    import torchrec.distributed.batched_embedding_kernel as bek
    import torchrec.distributed.embedding_lookup as el
    from torchrec.distributed.embeddingbag import EmbeddingBagCollectionSharder
    from torchrec.distributed.model_parallel import DistributedModelParallel
    from torchrec.distributed.planner import EmbeddingShardingPlanner, Topology
    from torchrec.distributed.planner.types import ParameterConstraints
    from torchrec.distributed.types import ShardingEnv
    from torchrec.modules.embedding_configs import EmbeddingBagConfig
    from torchrec.modules.embedding_modules import EmbeddingBagCollection
    from torchrec.sparse.jagged_tensor import KeyedJaggedTensor

    class Holder(nn.Module):
        def __init__(self, ebc):
            super().__init__()
            self.ebc = ebc

        def forward(self, kjt):
            return self.ebc(kjt)

    bek.SplitTableBatchedEmbeddingBagsCodegen = RecordingTBE
    lookup_log = []
    orig_lookup_forward = el.GroupedPooledEmbeddingsLookup.forward

    def logging_forward(self, sparse_features):
        out = orig_lookup_forward(self, sparse_features)
        idl = sparse_features.id_list_features
        lookup_log.append((idl.values().detach().clone(), idl.offsets().detach().clone(), out.detach().clone()))
        return out

    glob = _global_weights()
    keys, batches = _batches(W)
    pg = dist.group.WORLD
    meta, arrays = {"world_size": W, "rank": rank, "rows": ROWS, "dims": DIMS, "features": FEATURES, "keys": keys,
                    "b_local": B_LOCAL, "sharding": SHARDING[W]}, {}
    for kernel in ("batched_fused", "dense", "batched_fused:rowwise_adagrad"):
        adagrad = kernel.endswith(":rowwise_adagrad")
        kernel = kernel.split(":")[0]
        tables = [EmbeddingBagConfig(name=f"t{i}", embedding_dim=DIMS[i], num_embeddings=ROWS[i], feature_names=FEATURES[i])
                  for i in range(len(ROWS))]
        model = Holder(EmbeddingBagCollection(tables=tables, device=torch.device("meta")))
        fused_params = {"learning_rate": LR}
        if adagrad:
            from fbgemm_gpu.split_embedding_configs import EmbOptimType

            fused_params = {"learning_rate": LR, "optimizer": EmbOptimType.EXACT_ROWWISE_ADAGRAD, "eps": 1e-3}
        sharders = [EmbeddingBagCollectionSharder(fused_params=fused_params)]
        constraints = {f"t{i}": ParameterConstraints(sharding_types=[SHARDING[W][i]], compute_kernels=[kernel])
                       for i in range(len(ROWS))}
        planner = EmbeddingShardingPlanner(topology=Topology(world_size=W, compute_device="cpu"), constraints=constraints)
        plan = planner.collective_plan(model, sharders, pg)
        if kernel == "dense":
            el.GroupedPooledEmbeddingsLookup.forward = logging_forward
        n_tbe_before = len(RecordingTBE.instances)
        dmp = DistributedModelParallel(model, env=ShardingEnv.from_process_group(pg), device=torch.device("cpu"), plan=plan,
                                       sharders=sharders)
        if adagrad:  # only the optimizer surface is recorded for this variant
            fsd = dmp.fused_optimizer.state_dict()
            meta["fused_optimizer_state_keys_rowwise_adagrad"] = {str(k): sorted(v.keys()) for k, v in fsd["state"].items()}
            meta["fused_optimizer_state_shapes_rowwise_adagrad"] = {
                str(k): {kk: [list(sh.tensor.shape) for sh in vv.local_shards()] for kk, vv in v.items()}
                for k, v in fsd["state"].items()}
            meta["tbe_ctor_rowwise_adagrad"] = [t.ctor for t in RecordingTBE.instances[n_tbe_before:]]
            continue
        sd = dmp.state_dict()
        _fill_weights(sd, glob)
        if kernel == "batched_fused":
            meta["plan"] = {n: {"sharding_type": p.sharding_type, "compute_kernel": p.compute_kernel, "ranks": p.ranks,
                                "shards": [{"offsets": list(s.shard_offsets), "sizes": list(s.shard_sizes),
                                            "placement": str(s.placement)} for s in p.sharding_spec.shards]}
                            for n, p in plan.plan["ebc"].items()}
            meta["state_dict_keys"] = list(sd.keys())
            fo = dmp.fused_optimizer
            meta["fused_optimizer_param_keys"] = list(fo.params.keys())
            fsd = fo.state_dict()
            meta["fused_optimizer_state_dict_keys"] = sorted(fsd.keys())
            meta["fused_optimizer_state_keys_sgd"] = {str(k): sorted(v.keys()) for k, v in fsd["state"].items()}
            meta["fused_optimizer_param_groups"] = [{k: (v if k == "params" else _plain(v)) for k, v in g.items()}
                                                    for g in fsd.get("param_groups", [])]
            meta["named_parameters"] = [n for n, _ in dmp.named_parameters()]
            meta["named_buffers"] = [n for n, _ in dmp.named_buffers()]
        for step, per_rank in enumerate(batches):
            lengths, vals = per_rank[rank]
            kjt = KeyedJaggedTensor.from_lengths_sync(keys=keys, values=torch.from_numpy(vals), lengths=torch.from_numpy(lengths))
            out = dmp(kjt)
            if hasattr(out, "wait"):
                out = out.wait()
            if kernel == "dense":
                arrays[f"step{step}_kt_values"] = out.values().detach().numpy()
                meta.setdefault("kt_keys", out.keys())
                meta.setdefault("kt_length_per_key", out.length_per_key())
            else:
                arrays[f"step{step}_in_lengths"] = lengths
                arrays[f"step{step}_in_values"] = vals
        if kernel == "batched_fused":
            tbes = RecordingTBE.instances[n_tbe_before:]
            meta["tbe"] = []
            for i, tbe in enumerate(tbes):
                meta["tbe"].append({"ctor": tbe.ctor, "ctor_kwarg_order": tbe.ctor_kwarg_order, "calls": len(tbe.calls)})
                for t, w in enumerate(tbe.weights_list):
                    arrays[f"tbe{i}_weight{t}"] = w.detach().numpy().copy()
                for c, (ind, off, psw) in enumerate(tbe.calls):
                    arrays[f"tbe{i}_call{c}_indices"] = ind.numpy()
                    arrays[f"tbe{i}_call{c}_offsets"] = off.numpy()
                    meta["tbe"][i].setdefault("call_dtypes", []).append([str(ind.dtype), str(off.dtype),
                                                                         None if psw is None else str(psw.dtype)])
        else:
            el.GroupedPooledEmbeddingsLookup.forward = orig_lookup_forward
            meta["dense_lookups"] = len(lookup_log)
            for c, (v, o, out) in enumerate(lookup_log):
                arrays[f"dense_lookup{c}_values"] = v.numpy()
                arrays[f"dense_lookup{c}_offsets"] = o.numpy()
                arrays[f"dense_lookup{c}_out"] = out.numpy()
    if rank == 0:
        for t, w in enumerate(glob):
            arrays[f"global_weight{t}"] = w.numpy()
    with open(os.path.join(outdir, f"harness_w{W}_rank{rank}.json"), "w") as fh:
        json.dump(meta, fh, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(outdir, f"harness_w{W}_rank{rank}.npz"), **arrays)
    dist.barrier()
    dist.destroy_process_group()


def main():
    import socket

    outdir = sys.argv[1] if len(sys.argv) > 1 else HERE
    for W in (1, 2):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        mp.spawn(_worker, args=(W, port, outdir), nprocs=W, join=True)
        print("captured world size", W)


if __name__ == "__main__":
    main()
