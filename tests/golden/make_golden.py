"""Generates tests/golden/*.npz by importing the REFERENCE (read-only, /root/reference) in the
build container.  Not run on the GPU box (the reference does not travel); the .npz fixtures are
committed and are pure data: inputs and the reference's outputs.

What is pinned
  ebc_*.npz     reference torchrec.modules.embedding_modules.EmbeddingBagCollection
                (modules/embedding_modules.py:127-193) forward, then sum(out * grad_seed).backward()
                and torch.optim.SGD.step() — the ground truth the reference's own
                sharded-vs-unsharded test uses for BATCHED_FUSED + EXACT_SGD
                (distributed/test_utils/test_model_parallel_base.py:257-283).
  bucketize_*.npz  reference pure-python block_bucketize_ref (distributed/tests/test_utils.py:83-236)
  recat.npz     reference _get_recat (distributed/dist_data.py:40-118)
  dlrm_*.npz    reference torchrec.models.dlrm.DLRM forward on seeded weights (models/dlrm.py:387-406)

The reference imports `fbgemm_gpu` and `pyre_extensions` unconditionally; this script supplies
this repo's own `fbgemm_gpu` package (enums + op schemas) plus oracle-backed CPU op
implementations (tests/_cpu_ops.py) and a 2-function `pyre_extensions` stand-in.  None of the
pinned outputs above depends on those ops except KJT offsets (a cumsum).
"""
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _paths  # noqa: E402,F401
import _cpu_ops  # noqa: E402

REFERENCE = "/root/reference"


def _import_reference():
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


def make_kjt_inputs(rng, F, B, max_len, rows, weighted, fixed_len=None):
    lengths = (np.full(F * B, fixed_len, dtype=np.int32) if fixed_len is not None
               else rng.integers(0, max_len + 1, size=F * B).astype(np.int32))
    vals = []
    for f in range(F):
        n = int(lengths[f * B:(f + 1) * B].sum())
        vals.append(rng.integers(0, rows[f], size=n))
    values = np.concatenate(vals).astype(np.int64) if vals else np.zeros(0, np.int64)
    weights = rng.random(values.size).astype(np.float32) if weighted else None
    return lengths, values, weights


def gen_ebc(torchrec, name, seed, rows, dims, B, max_len, weighted, pooling, fixed_len=None, lr=0.1):
    from torchrec.modules.embedding_configs import EmbeddingBagConfig, PoolingType
    from torchrec.modules.embedding_modules import EmbeddingBagCollection
    from torchrec.sparse.jagged_tensor import KeyedJaggedTensor

    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    F = len(rows)
    keys = [f"f{i}" for i in range(F)]
    tables = [
        EmbeddingBagConfig(name=f"t{i}", embedding_dim=dims[i], num_embeddings=rows[i], feature_names=[keys[i]],
                           pooling=PoolingType.SUM if pooling == "sum" else PoolingType.MEAN)
        for i in range(F)
    ]
    ebc = EmbeddingBagCollection(tables=tables, is_weighted=weighted)
    lengths, values, weights = make_kjt_inputs(rng, F, B, max_len, rows, weighted, fixed_len)
    kjt = KeyedJaggedTensor.from_lengths_sync(
        keys=keys, values=torch.from_numpy(values), lengths=torch.from_numpy(lengths),
        weights=torch.from_numpy(weights) if weights is not None else None)
    w_before = [ebc.embedding_bags[f"t{i}"].weight.detach().clone().numpy() for i in range(F)]
    out = ebc(kjt).values()
    grad_seed = torch.from_numpy(rng.standard_normal(out.shape).astype(np.float32))
    opt = torch.optim.SGD(ebc.parameters(), lr=lr)
    opt.zero_grad()
    (out * grad_seed).sum().backward()
    opt.step()
    w_after = [ebc.embedding_bags[f"t{i}"].weight.detach().clone().numpy() for i in range(F)]
    data = dict(rows=np.array(rows), dims=np.array(dims), B=B, lengths=lengths, values=values,
                offsets=kjt.offsets().numpy().astype(np.int64), out=out.detach().numpy(),
                grad_out=grad_seed.numpy(), lr=np.float32(lr), pooling=pooling, weighted=weighted)
    if weights is not None:
        data["weights"] = weights
    for i in range(F):
        data[f"w_before_{i}"] = w_before[i]
        data[f"w_after_{i}"] = w_after[i]
    np.savez_compressed(os.path.join(HERE, f"ebc_{name}.npz"), **data)
    print("wrote", name, out.shape)


def gen_bucketize(torchrec):
    sys.path.insert(0, os.path.join(REFERENCE, "torchrec", "distributed", "tests"))
    import importlib.util

    # torch.Tensor.cuda is a no-op here so the reference helper (written for a GPU box) runs on CPU
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        spec = importlib.util.spec_from_file_location(
            "ref_test_utils", os.path.join(REFERENCE, "torchrec/distributed/tests/test_utils.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        from torchrec.sparse.jagged_tensor import KeyedJaggedTensor
        import math

        rnd = random.Random(7)
        cases = []
        for case, (W, F, B, idx_t, off_t) in enumerate([
            (1, 1, 1, torch.int64, torch.int32), (2, 3, 4, torch.int64, torch.int32),
            (3, 5, 7, torch.int32, torch.int64), (8, 15, 15, torch.int64, torch.int64),
            (129, 4, 9, torch.int32, torch.int32), (17, 7, 15, torch.int64, torch.int32),
        ]):
            MAXL = 10
            MAXROW = MAXL * 15
            lengths_list = [rnd.randrange(MAXL + 1) for _ in range(F * B)]
            indices_lists = [rnd.sample(range(MAXROW), sum(lengths_list[f * B:(f + 1) * B])) for f in range(F)]
            indices_list = [i for l in indices_lists for i in l]
            weights_list = [rnd.randint(1, 100) for _ in indices_list]
            block_sizes_list = [math.ceil((max(l) + 1) / W) if l else 1 for l in indices_lists]
            kjt = KeyedJaggedTensor(
                keys=[f"feature_{i}" for i in range(F)],
                lengths=torch.tensor(lengths_list, dtype=off_t),
                values=torch.tensor(indices_list, dtype=idx_t),
                weights=torch.tensor(weights_list, dtype=torch.float))
            block_sizes = torch.tensor(block_sizes_list, dtype=idx_t)
            exp = mod.block_bucketize_ref(kjt, W, block_sizes)
            np.savez_compressed(
                os.path.join(HERE, f"bucketize_{case}.npz"), W=W, F=F, B=B,
                lengths=kjt.lengths().numpy(), values=kjt.values().numpy(),
                weights=kjt.weights().numpy(), block_sizes=block_sizes.numpy(),
                exp_lengths=exp.lengths().numpy(), exp_values=exp.values().numpy(),
                exp_weights=exp.weights().numpy())
            cases.append(case)
        print("wrote bucketize cases", cases)
    finally:
        torch.Tensor.cuda = orig_cuda


def gen_recat(torchrec):
    from torchrec.distributed.dist_data import _get_recat

    out = {}
    for (lw, ls, bs) in [(2, 4, 1), (2, 4, 2), (3, 8, 1), (1, 2, 1), (4, 8, 2), (5, 4, 1)]:
        r = _get_recat(local_split=lw, num_splits=ls, stagger=bs, device=torch.device("cpu"))
        out[f"recat_{lw}_{ls}_{bs}"] = r.numpy() if r is not None else np.zeros(0, np.int32)
    np.savez_compressed(os.path.join(HERE, "recat.npz"), **out)
    print("wrote recat", {k: v.tolist() for k, v in out.items()})


def gen_dlrm(torchrec):
    from torchrec.models.dlrm import DLRM
    from torchrec.modules.embedding_configs import EmbeddingBagConfig
    from torchrec.modules.embedding_modules import EmbeddingBagCollection
    from torchrec.sparse.jagged_tensor import KeyedJaggedTensor

    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    B, D, F = 6, 8, 3
    rows = [11, 7, 5]
    keys = [f"f{i}" for i in range(F)]
    ebc = EmbeddingBagCollection(tables=[
        EmbeddingBagConfig(name=f"t{i}", embedding_dim=D, num_embeddings=rows[i], feature_names=[keys[i]])
        for i in range(F)])
    model = DLRM(embedding_bag_collection=ebc, dense_in_features=13, dense_arch_layer_sizes=[16, D],
                 over_arch_layer_sizes=[12, 1])
    lengths, values, _ = make_kjt_inputs(rng, F, B, 3, rows, False)
    kjt = KeyedJaggedTensor.from_lengths_sync(keys=keys, values=torch.from_numpy(values),
                                              lengths=torch.from_numpy(lengths))
    dense = torch.from_numpy(rng.standard_normal((B, 13)).astype(np.float32))
    logits = model(dense_features=dense, sparse_features=kjt)
    data = dict(B=B, D=D, rows=np.array(rows), lengths=lengths, values=values, dense=dense.numpy(),
                logits=logits.detach().numpy())
    for k, v in model.state_dict().items():
        data["sd::" + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "dlrm_small.npz"), **data)
    print("wrote dlrm_small", logits.shape, list(model.state_dict().keys()))


def main():
    torchrec = _import_reference()
    gen_ebc(torchrec, "l1_sum", 1, rows=[50, 3, 1000, 17], dims=[16, 16, 16, 16], B=33, max_len=1, weighted=False,
            pooling="sum", fixed_len=1)
    gen_ebc(torchrec, "ragged_sum", 2, rows=[40, 9, 300], dims=[8, 32, 128], B=21, max_len=6, weighted=False,
            pooling="sum")
    gen_ebc(torchrec, "ragged_weighted", 3, rows=[25, 6, 90, 2], dims=[4, 12, 64, 20], B=18, max_len=5,
            weighted=True, pooling="sum")
    gen_ebc(torchrec, "ragged_mean", 4, rows=[31, 5, 64], dims=[16, 8, 24], B=25, max_len=7, weighted=False,
            pooling="mean")
    gen_ebc(torchrec, "long_bags", 5, rows=[200, 13], dims=[128, 64], B=9, max_len=40, weighted=False,
            pooling="sum")
    # the headline geometry (D = 128, pooling factor 1: a pure gather) on 10 tables incl. tiny ones
    gen_ebc(torchrec, "l1_d128", 6, rows=[300, 3, 100, 17, 62, 4, 97, 128, 220, 10], dims=[128] * 10, B=64, max_len=1,
            weighted=False, pooling="sum", fixed_len=1)
    gen_bucketize(torchrec)
    gen_recat(torchrec)
    gen_dlrm(torchrec)


if __name__ == "__main__":
    main()
