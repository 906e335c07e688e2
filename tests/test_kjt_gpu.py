"""KeyedJaggedTensor at class level on the GPU: the reference's known answers for to_dict / split / permute
(torchrec/sparse/tests/test_jagged_tensor.py:549-755), as a table of inputs and expected outputs; the permute cases go
through `torch.ops.fbgemm.permute_2D_sparse_data` -> csrc/sparse_ops.hip, offsets through
`asynchronous_complete_cumsum`.  Bit-exact (index / copy work)."""
import pytest
import torch

import _paths  # noqa: F401

pytestmark = pytest.mark.gpu

VALUES = [1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0]
WEIGHTS = [1.0, 0.5, 1.5, 1.0, 0.5, 1.0, 1.0, 1.5]


def _dev():
    return torch.device("cuda", 0)


def _kjt_two_keys():
    """2 keys x 3 samples, built from OFFSETS (test_jagged_tensor.py:550-560)."""
    from torchrec_amd.sparse.jagged_tensor import KeyedJaggedTensor

    d = _dev()
    return KeyedJaggedTensor(keys=["index_0", "index_1"], values=torch.tensor(VALUES, device=d),
                             weights=torch.tensor(WEIGHTS, device=d),
                             offsets=torch.tensor([0, 2, 2, 3, 4, 5, 8], dtype=torch.int32, device=d))


def _kjt_three_keys(weighted):
    """3 keys x 3 samples, built from LENGTHS (test_jagged_tensor.py:633-645)."""
    from torchrec_amd.sparse.jagged_tensor import KeyedJaggedTensor

    d = _dev()
    return KeyedJaggedTensor.from_lengths_sync(
        keys=["index_0", "index_1", "index_2"], values=torch.tensor(VALUES, device=d),
        lengths=torch.tensor([0, 2, 0, 1, 1, 1, 0, 3, 0], dtype=torch.int32, device=d),
        weights=torch.tensor(WEIGHTS, device=d) if weighted else None)


def _eq(t, expected, dtype=None):
    e = torch.tensor(expected, dtype=dtype if dtype is not None else t.dtype)
    assert t.is_cuda and torch.equal(t.cpu(), e), (t.cpu().tolist(), expected)


# key -> (lengths, values, weights) of the two-key tensor: what to_dict and split([1, 1]) must both give
PER_KEY = {"index_0": ([2, 0, 1], [1.0, 2.0, 3.0], [1.0, 0.5, 1.5]),
           "index_1": ([1, 1, 3], [4.0, 5.0, 6.0, 7.0, 8.0], [1.0, 0.5, 1.0, 1.0, 1.5])}


def test_to_dict_known_answer():
    from torchrec_amd.sparse.jagged_tensor import JaggedTensor

    got = _kjt_two_keys().to_dict()
    assert list(got.keys()) == ["index_0", "index_1"]
    for key, (lengths, values, weights) in PER_KEY.items():
        jt = got[key]
        assert isinstance(jt, JaggedTensor)
        _eq(jt.lengths(), lengths, torch.int32)
        _eq(jt.values(), values)
        _eq(jt.weights(), weights)


def test_split_known_answers():
    from torchrec_amd.sparse.jagged_tensor import KeyedJaggedTensor

    parts = _kjt_two_keys().split([1, 1])
    assert [p.keys() for p in parts] == [["index_0"], ["index_1"]] and all(isinstance(p, KeyedJaggedTensor) for p in parts)
    for p in parts:
        lengths, values, weights = PER_KEY[p.keys()[0]]
        _eq(p.lengths(), lengths, torch.int32)
        _eq(p.values(), values)
        _eq(p.weights(), weights)
    # a zero-key segment keeps the stride and is empty everywhere (test_jagged_tensor.py:606-630)
    empty, whole = _kjt_two_keys().split([0, 2])
    assert empty.keys() == [] and empty.stride() == 3 and whole.stride() == 3
    assert empty.lengths().numel() == 0 and empty.values().numel() == 0 and empty.weights().numel() == 0
    assert whole.keys() == ["index_0", "index_1"]
    _eq(whole.lengths(), [2, 0, 1, 1, 1, 3], torch.int32)
    _eq(whole.values(), VALUES)
    _eq(whole.weights(), WEIGHTS)


PERMUTE_CASES = [
    # indices, weighted, keys, offset_per_key, values, lengths, weights
    ([1, 0, 2], True, ["index_1", "index_0", "index_2"], [0, 3, 5, 8], [3.0, 4.0, 5.0, 1.0, 2.0, 6.0, 7.0, 8.0],
     [1, 1, 1, 0, 2, 0, 0, 3, 0], [1.5, 1.0, 0.5, 1.0, 0.5, 1.0, 1.0, 1.5]),
    ([1, 0, 2], False, ["index_1", "index_0", "index_2"], [0, 3, 5, 8], [3.0, 4.0, 5.0, 1.0, 2.0, 6.0, 7.0, 8.0],
     [1, 1, 1, 0, 2, 0, 0, 3, 0], None),
    ([1, 0, 2, 1, 1], False, ["index_1", "index_0", "index_2", "index_1", "index_1"], [0, 3, 5, 8, 11, 14],
     [3.0, 4.0, 5.0, 1.0, 2.0, 6.0, 7.0, 8.0, 3.0, 4.0, 5.0, 3.0, 4.0, 5.0], [1, 1, 1, 0, 2, 0, 0, 3, 0, 1, 1, 1, 1, 1, 1], None),
]


@pytest.mark.parametrize("indices,weighted,keys,opk,values,lengths,weights", PERMUTE_CASES,
                         ids=["weighted", "plain", "duplicates"])
def test_permute_known_answers(indices, weighted, keys, opk, values, lengths, weights):
    out = _kjt_three_keys(weighted).permute(indices)
    assert out.keys() == keys and out.offset_per_key() == opk
    _eq(out.values(), values)
    _eq(out.lengths(), lengths, torch.int32)
    if weights is None:
        assert out.weights_or_none() is None
    else:
        _eq(out.weights(), weights)
    # offsets of the permuted tensor: the complete cumsum of its lengths (jagged_tensor.py:35-36, 796-799)
    _eq(out.offsets(), [0] + torch.tensor(lengths).cumsum(0).tolist(), out.offsets().dtype)
