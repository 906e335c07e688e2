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


def _second_group_worker(rank, port, ret):
    import os

    import torch.distributed as dist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from torchrec_amd.distributed.comm import init_rccl_process_group, new_rccl_group

    init_rccl_process_group(dev, rank=0, world_size=1)
    try:
        dense_pg = new_rccl_group(dist.group.WORLD)  # what DLRMTrain.capture_hip_graphs creates at N > 1
        a = torch.arange(1 << 20, dtype=torch.float32, device=dev)
        out = torch.empty_like(a)
        w1 = dist.all_to_all_single(out, a, [a.numel()], [a.numel()], async_op=True)  # first communicator
        g = torch.ones(1 << 16, device=dev)
        w2 = dist.all_reduce(g, group=dense_pg, async_op=True)                      # second one, concurrently
        w1.wait()
        w2.wait()
        torch.cuda.synchronize()
        ret[0] = (bool(torch.equal(out, a)), float(g.sum()), dist.get_world_size(dense_pg))
    finally:
        dist.destroy_process_group()


def test_second_rccl_communicator_for_the_dense_all_reduces():
    import socket

    import torch.multiprocessing as mp

    from _results import ResultStore

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = ResultStore()
    mp.spawn(_second_group_worker, args=(port, ret), nprocs=1, join=True)
    assert ret[0] == (True, float(1 << 16), 1)
