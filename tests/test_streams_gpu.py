"""Side streams on their own hardware queue (fbgemm_gpu/_streams.py) and the collective stream's priority
(torchrec_amd/distributed/comm.py)."""
import pytest
import torch

import _paths  # noqa: F401

pytestmark = pytest.mark.gpu


def test_probe_sees_a_stream_sharing_with_itself():
    from fbgemm_gpu._streams import shares_hw_queue

    s = torch.cuda.Stream()
    assert shares_hw_queue(s, s)  # in-order by definition: the probe must say so
    assert shares_hw_queue(torch.cuda.default_stream(), torch.cuda.default_stream())


def test_side_stream_runs_beside_the_default_stream():
    from fbgemm_gpu._streams import shares_hw_queue, side_stream

    dev = torch.device("cuda", torch.cuda.current_device())
    a, b = side_stream(dev), side_stream(dev)
    default = torch.cuda.default_stream(dev)
    assert a != default and b != default
    assert not shares_hw_queue(a, default)
    assert not shares_hw_queue(b, default)
    # a spin on the side stream does not hold back the default stream, measured end to end
    x = torch.zeros(1 << 20, device=dev)
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    with torch.cuda.stream(a):
        e0.record()
        torch.cuda._sleep(50_000_000)
        e2.record()
    x.add_(1.0)
    e1.record()
    torch.cuda.synchronize()
    assert e0.elapsed_time(e1) < 0.5 * e0.elapsed_time(e2)


def test_tbe_backward_sort_stream_is_probed():
    from fbgemm_gpu import split_table_batched_embeddings_ops as tbe
    from fbgemm_gpu._streams import shares_hw_queue

    m = tbe.SplitTableBatchedEmbeddingBagsCodegen([(1000, 16, tbe.EmbeddingLocation.DEVICE, tbe.ComputeDevice.CUDA)] * 2,
                                                  optimizer=tbe.OptimType.EXACT_SGD, learning_rate=0.1)
    B = 64
    idx = torch.randint(0, 1000, (2 * B,), device="cuda")
    off = torch.arange(2 * B + 1, device="cuda")
    m(idx, off).sum().backward()
    torch.cuda.synchronize()
    side = m._side_stream
    if side is not None:  # the overlapped sort was used
        assert not shares_hw_queue(side, torch.cuda.default_stream())
