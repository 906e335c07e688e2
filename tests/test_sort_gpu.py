"""The hand-written stable pair sort (csrc/radix_sort.hpp) through the C ABI (tbe_sort_pairs) against numpy's
stable argsort: bit-exact keys AND payload order (equal keys keep their input order — that is what makes the
backward's summation order a function of the input only).  No reference counterpart: fbgemm's backward sorts
inside the absent submodule."""
import ctypes

import numpy as np
import pytest
import torch

import _paths  # noqa: F401

pytestmark = pytest.mark.gpu


def hip_sort(keys: np.ndarray, payload: np.ndarray, key_bits: int):
    from fbgemm_gpu import _lib
    from fbgemm_gpu._lib import check, ptr, stream_ptr

    lib = _lib.load()
    dev = torch.device("cuda", 0)
    n = keys.size
    k = torch.from_numpy(keys.view(np.int32 if keys.itemsize == 4 else np.int64)).to(dev)
    p = torch.from_numpy(payload.view(np.int32 if payload.itemsize == 4 else np.int64)).to(dev)
    kt, pt = torch.empty_like(k), torch.empty_like(p)
    nbytes = lib.tbe_sort_pairs_workspace_bytes(n, key_bits)
    assert nbytes > 0
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
    off = (-ws.data_ptr()) % 256
    check(lib.tbe_sort_pairs(ptr(k), ptr(kt), ptr(p), ptr(pt), n, key_bits, keys.itemsize, payload.itemsize,
                             ws.data_ptr() + off, nbytes, stream_ptr(dev)), "tbe_sort_pairs")
    torch.cuda.synchronize()
    return k.cpu().numpy().view(keys.dtype), p.cpu().numpy().view(payload.dtype)


def timeouts() -> int:
    from fbgemm_gpu import _lib

    c = ctypes.c_int64(-1)
    assert _lib.load().tbe_debug_sort_timeouts(ctypes.byref(c)) == 0
    return c.value


def make_keys(rng, n, key_bits, kdtype, dist):
    hi = (1 << key_bits) - 1
    if dist == "uniform":
        k = rng.integers(0, hi, size=n, endpoint=True, dtype=np.uint64)
    elif dist == "few":  # thousands of equal keys (ids of 3-row tables) + a sentinel
        vals = rng.integers(0, hi, size=7, endpoint=True, dtype=np.uint64)
        k = vals[rng.integers(0, 7, size=n)]
        k[rng.random(n) < 0.05] = hi
    elif dist == "criteo":  # 26 tables' row ranges side by side, ids uniform per table
        rows = np.array([45833188, 36746, 17245, 7413, 20243, 3, 7114, 1441, 62, 29275261, 1572176, 345138, 10, 2209,
                         11267, 128, 4, 974, 14, 48937457, 11316796, 40094537, 452104, 12606, 104, 35], dtype=np.int64)
        base = np.concatenate([[0], np.cumsum(rows)[:-1]])
        t = np.arange(n) % 26
        k = (base[t] + (rng.random(n) * rows[t]).astype(np.int64)).astype(np.uint64) & np.uint64(hi)
    else:
        raise ValueError(dist)
    return k.astype(kdtype)


CASES = [
    # n, key_bits, key dtype, payload dtype, distribution
    (1, 1, np.uint32, np.uint32, "uniform"),
    (63, 7, np.uint32, np.uint32, "uniform"),
    (64, 10, np.uint32, np.uint64, "few"),
    (513, 11, np.uint32, np.uint32, "uniform"),
    (2048, 20, np.uint32, np.uint32, "few"),
    (2049, 28, np.uint32, np.uint32, "criteo"),
    (100_003, 32, np.uint32, np.uint64, "uniform"),
    (106_496, 26, np.uint32, np.uint32, "criteo"),       # 26 x 4096 (BASELINE config 2)
    (212_992, 26, np.uint32, np.uint32, "criteo"),       # per-rank share at 8 GPUs
    (1_703_936, 28, np.uint32, np.uint32, "criteo"),     # 26 x 65 536 (headline)
    (1_703_936, 28, np.uint32, np.uint64, "few"),
    (256 * 8192 + 5, 28, np.uint32, np.uint32, "criteo"),  # two tiles per segment
    (3_000_001, 19, np.uint32, np.uint64, "few"),        # 8-round tiles, three tiles per segment
    (50_000, 40, np.uint64, np.uint64, "uniform"),
    (300_000, 62, np.uint64, np.uint64, "few"),
    (70_000, 33, np.uint64, np.uint32, "uniform"),
]


@pytest.mark.parametrize("n,key_bits,kdt,pdt,dist", CASES)
def test_sort_pairs_matches_numpy_stable(n, key_bits, kdt, pdt, dist):
    rng = np.random.default_rng(n * 131 + key_bits)
    keys = make_keys(rng, n, key_bits, kdt, dist)
    payload = np.arange(n, dtype=pdt)
    before = timeouts()
    sk, sp = hip_sort(keys.copy(), payload.copy(), key_bits)
    order = np.argsort(keys, kind="stable")
    np.testing.assert_array_equal(sk, keys[order])
    np.testing.assert_array_equal(sp, payload[order])
    assert timeouts() == before  # no spin-wait give-up (another test may have injected one on purpose before)


def test_sort_ignores_bits_above_key_bits():
    """Only the low key_bits bits order the pairs (the cache's keys carry no high bits, the contract still holds)."""
    rng = np.random.default_rng(5)
    n, key_bits = 40_000, 13
    keys = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
    payload = np.arange(n, dtype=np.uint32)
    sk, sp = hip_sort(keys.copy(), payload.copy(), key_bits)
    order = np.argsort(keys & np.uint32((1 << key_bits) - 1), kind="stable")
    np.testing.assert_array_equal(sk, keys[order])
    np.testing.assert_array_equal(sp, payload[order])


def test_sort_back_to_back_on_one_workspace():
    """The state block is re-zeroed by every call: the same workspace sorts different sizes back to back."""
    rng = np.random.default_rng(9)
    before = timeouts()
    for n in (5000, 212_992, 77, 1_000_000, 4096):
        keys = make_keys(rng, n, 28, np.uint32, "criteo")
        payload = np.arange(n, dtype=np.uint32)
        sk, sp = hip_sort(keys.copy(), payload.copy(), 28)
        order = np.argsort(keys, kind="stable")
        np.testing.assert_array_equal(sk, keys[order])
        np.testing.assert_array_equal(sp, payload[order])
    assert timeouts() == before
