"""CPU, world_size 2 (gloo): the reference-named distribution primitives
(KJTAllToAll, PooledEmbeddingsAllToAll, PooledEmbeddingsReduceScatter, _get_recat) against their
defining properties (torchrec/distributed/tests/test_dist_data.py:58-168, :367-370, :431-434) and the
reference-generated recat golden vectors."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _paths  # noqa: F401

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_get_recat_matches_reference_golden():
    from torchrec_amd.distributed.dist_data import _get_recat

    g = np.load(os.path.join(GOLD, "recat.npz"))
    for k in g.files:
        _, lw, ls, st = k.split("_")
        np.testing.assert_array_equal(_get_recat(int(lw), int(ls), int(st)).numpy(), g[k])


def _make_kjt(rank, keys, B, weighted):
    from torchrec_amd.sparse.jagged_tensor import KeyedJaggedTensor

    rng = np.random.default_rng(50 + rank)
    lengths = rng.integers(0, 4, size=len(keys) * B).astype(np.int32)
    vals = rng.integers(0, 1000, size=int(lengths.sum())).astype(np.int64)
    w = rng.random(vals.size).astype(np.float32) if weighted else None
    return KeyedJaggedTensor.from_lengths_sync(keys, torch.from_numpy(vals), torch.from_numpy(lengths),
                                               torch.from_numpy(w) if weighted else None)


def _worker(rank, W, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    try:
        import _cpu_ops
        _cpu_ops.register()
        from torchrec_amd.distributed.dist_data import (KJTAllToAll, PooledEmbeddingsAllToAll,
                                                          PooledEmbeddingsReduceScatter)

        pg = dist.group.WORLD
        keys = ["a", "b", "c"]
        splits = [2, 1]
        B = 3
        kjt = _make_kjt(rank, keys, B, True)
        out = KJTAllToAll(pg, splits, torch.device("cpu"))(kjt).wait().wait()
        res = {"kjt": (out.keys(), out.lengths().numpy(), out.values().numpy(), out.weights().numpy(), out.stride())}
        # pooled a2a: rank r holds D_r columns for the global batch
        dims = [8, 4]
        Bl = 5
        rng = np.random.default_rng(7 + rank)
        x = torch.from_numpy(rng.standard_normal((W * Bl, dims[rank])).astype(np.float32)).requires_grad_()
        y = PooledEmbeddingsAllToAll(pg, dims, torch.device("cpu"))(x).wait()
        gy = torch.from_numpy(np.random.default_rng(70 + rank).standard_normal(tuple(y.shape)).astype(np.float32))
        y.backward(gy)
        res["a2a"] = (x.detach().numpy(), y.detach().numpy(), gy.numpy(), x.grad.numpy())
        # reduce-scatter
        z = torch.from_numpy(rng.standard_normal((W * Bl, 6)).astype(np.float32)).requires_grad_()
        r = PooledEmbeddingsReduceScatter(pg)(z).wait()
        r.backward(torch.ones_like(r))
        res["rs"] = (z.detach().numpy(), r.detach().numpy(), z.grad.numpy())
        ret[rank] = res
    finally:
        dist.destroy_process_group()


def test_dist_primitives_world2():
    W = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(W, _free_port(), ret), nprocs=W, join=True)
    keys = ["a", "b", "c"]
    owner_keys = [["a", "b"], ["c"]]
    B = 3
    src = [_make_kjt(r, keys, B, True) for r in range(W)]
    for me in range(W):
        okeys, lengths, values, weights, stride = ret[me]["kjt"]
        assert okeys == owner_keys[me] and stride == W * B
        exp_l = np.concatenate([np.concatenate([src[r][k].lengths().numpy() for r in range(W)]) for k in owner_keys[me]])
        exp_v = np.concatenate([np.concatenate([src[r][k].values().numpy() for r in range(W)]) for k in owner_keys[me]])
        exp_w = np.concatenate([np.concatenate([src[r][k].weights().numpy() for r in range(W)]) for k in owner_keys[me]])
        np.testing.assert_array_equal(lengths, exp_l)
        np.testing.assert_array_equal(values, exp_v)
        np.testing.assert_array_equal(weights, exp_w)
    Bl = 5
    for me in range(W):
        _, y, gy, _ = ret[me]["a2a"]
        exp = np.concatenate([ret[r]["a2a"][0][me * Bl:(me + 1) * Bl] for r in range(W)], axis=1)
        np.testing.assert_array_equal(y, exp)
    col = [0, 8, 12]
    for r in range(W):
        # grad of rank r's input = its column block of every rank's grad_out, / W (comm_ops.py:527-528)
        exp = np.concatenate([ret[me]["a2a"][2][:, col[r]:col[r + 1]] for me in range(W)], axis=0) / W
        np.testing.assert_allclose(ret[r]["a2a"][3], exp, rtol=1e-6, atol=1e-7)
    for me in range(W):
        z, red, gz = ret[me]["rs"]
        exp = sum(ret[r]["rs"][0][me * Bl:(me + 1) * Bl] for r in range(W))
        np.testing.assert_allclose(red, exp, rtol=1e-6, atol=1e-6)
        np.testing.assert_array_equal(gz, np.full_like(z, 1.0 / W))  # test_dist_data.py:431-434
