"""CPU: the C-ABI library loads and exports every symbol include/tbe_hip.h declares
(no compute calls without a GPU), and the product refuses CPU tensors loudly."""
import os
import re

import pytest
import torch

import _paths
from fbgemm_gpu import _lib


def declared_symbols():
    hdr = open(os.path.join(_paths.ROOT, "include", "tbe_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(tbe_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/tbe_hip.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), "ctypes SIGNATURES out of sync with the header"
    assert lib.tbe_abi_version() == 3


def test_argument_validation_happens_before_any_launch():
    lib = _lib.load()
    rc = lib.tbe_cumsum(None, None, -1, 4, 0, None, 0, None)
    assert rc == -1 and b"n < 0" in lib.tbe_last_error()
    rc = lib.tbe_forward_pooled_f32(None, None, None, None, 0, 1, 0, None, 0, None, None, 0, None, None, 0, None, None, None)
    assert rc == -1


def test_a_reported_kernel_fault_raises_once_at_the_next_check_point():
    """A sort give-up is written to the library's fault word (GPU-mapped host memory; plain host memory on a box
    without a device) and must surface as an exception at the host's next check point — a delta, so one raise per
    batch of faults.  Here the host-side test hook plays the kernel."""
    lib = _lib.load()
    _lib.raise_on_faults("test: clean start")  # nothing pending
    before = _lib.fault_count()
    assert lib.tbe_debug_inject_fault_host() == 0
    assert _lib.fault_count() == before + 1
    with pytest.raises(_lib.KernelFaultError, match="give-up"):
        _lib.raise_on_faults("test")
    _lib.raise_on_faults("test: the same fault is not reported twice")
    assert lib.tbe_debug_inject_fault_host() == 0 and lib.tbe_debug_inject_fault_host() == 0
    with pytest.raises(_lib.KernelFaultError, match="2 spin-wait"):
        _lib.raise_on_faults("test")


def test_backward_and_prefetch_refuse_more_than_2_pow_29_ids_before_any_launch():
    """the sort's {tag | count} words cap a call at 2^29 - 1 ids; every entry point says so up front."""
    lib = _lib.load()
    big = 1 << 29
    assert lib.tbe_backward_workspace_bytes(big, 1, 1, 128, 20) == 0
    assert lib.tbe_backward_workspace_bytes(big - 1, 1, 1, 128, 20) > 0
    assert lib.tbe_cache_prefetch_workspace_bytes(big, 20) == 0
    rc = lib.tbe_backward_prepare(None, None, 1, 1, 128, 20, None, big, None, 0, 0, None, 0, None, None, None)
    assert rc == -1 and b"2^29" in lib.tbe_last_error()


def test_cpu_tensors_are_refused_not_silently_computed():
    from fbgemm_gpu import _ops

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _ops.asynchronous_complete_cumsum(torch.tensor([1, 2, 3], dtype=torch.int32))
    from fbgemm_gpu.split_table_batched_embeddings_ops import (
        ComputeDevice, DenseTableBatchedEmbeddingBagsCodegen, EmbeddingLocation,
        SplitTableBatchedEmbeddingBagsCodegen)

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        SplitTableBatchedEmbeddingBagsCodegen([(10, 4, EmbeddingLocation.HOST, ComputeDevice.CPU)])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        DenseTableBatchedEmbeddingBagsCodegen([(10, 4)], use_cpu=True)


def test_enum_surface_used_by_reference():
    # names imported at torchrec/distributed/batched_embedding_kernel.py:17-25,
    # embedding_types.py:14, modules/embedding_configs.py:14-15
    from fbgemm_gpu.split_embedding_configs import EmbOptimType, SparseType
    from fbgemm_gpu.split_table_batched_embeddings_ops import (  # noqa: F401
        ComputeDevice, DenseTableBatchedEmbeddingBagsCodegen, EmbeddingLocation,
        IntNBitTableBatchedEmbeddingBagsCodegen, PoolingMode, SplitTableBatchedEmbeddingBagsCodegen,
        rounded_row_size_in_bytes)
    from fbgemm_gpu.permute_pooled_embedding_modules import PermutePooledEmbeddings  # noqa: F401

    assert [e.name for e in EmbeddingLocation] == ["DEVICE", "MANAGED", "MANAGED_CACHING", "HOST"]
    assert {e.name for e in PoolingMode} == {"SUM", "MEAN", "NONE"}
    assert {"FP32", "FP16", "INT8", "INT4", "INT2"} <= {e.name for e in SparseType}
    assert {"EXACT_SGD", "EXACT_ROWWISE_ADAGRAD", "ADAM"} <= {e.name for e in EmbOptimType}


def test_recorded_gemm_choices_file_is_well_formed():
    """torchrec_amd/tuning/gemm_gfx950_dlrm.csv: validators + one line per (op, shape) with a solution name."""
    import os

    import torchrec_amd.tuning as tuning

    lines = [ln.strip() for ln in open(tuning._FILE) if ln.strip()]
    validators = [ln for ln in lines if ln.startswith("Validator,")]
    rows = [ln.split(",") for ln in lines if not ln.startswith("Validator,")]
    assert any("gfx950" in v for v in validators)
    assert rows and all(len(r) == 4 and (r[2].startswith("Gemm_") or r[2] == "Default") and float(r[3]) > 0 for r in rows)
    assert len({(r[0], r[1]) for r in rows}) == len(rows)
    assert any("_65536_" in r[1] for r in rows) and any("_8192_" in r[1] for r in rows)
    assert os.path.basename(tuning._FILE) == "gemm_gfx950_dlrm.csv"


def test_rccl_options_put_the_collective_stream_on_its_own_priority(monkeypatch):
    """torchrec_amd/distributed/comm.py: high-priority collective stream by default (its own hardware queue)."""
    import torch.distributed as dist

    if not hasattr(dist, "ProcessGroupNCCL"):
        pytest.skip("torch built without the nccl (RCCL) backend")
    from torchrec_amd.distributed.comm import rccl_options

    assert rccl_options().is_high_priority_stream is True
    assert rccl_options(False).is_high_priority_stream is False
    monkeypatch.setenv("TORCHREC_AMD_RCCL_HIGH_PRIORITY", "0")
    assert rccl_options().is_high_priority_stream is False


def test_exchange_group_leaves_non_rccl_groups_alone():
    """torchrec_amd/distributed/comm.py exchange_group: only an RCCL group with a normal-priority collective stream gets a
    second communicator; gloo groups (CPU tests, the one-GPU rehearsal) and None are returned as they are."""
    import os
    import socket

    import torch.distributed as dist

    from torchrec_amd.distributed.comm import exchange_group

    assert exchange_group(None, "cpu") is None
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        assert exchange_group(dist.group.WORLD, "cpu") is dist.group.WORLD
    finally:
        dist.destroy_process_group()


def test_explicit_step_policies():
    """models/dlrm.py defaults: late weight-gradient layers only at per-rank batches of 16 384 .. 32 767 (and never in
    half-batch mode), half-batches from 32 768 — the crossovers measured with emulated link / all-reduce times
    (DESIGN.md §4)."""
    import torchrec_amd.models.dlrm as dlrm

    if dlrm._WGRAD_LATE_LAYERS == "auto":
        assert [dlrm._late_layers(b, False) for b in (4096, 8192, 16384, 32767, 32768, 65536)] == [0, 0, 2, 2, 0, 0]
        assert dlrm._late_layers(16384, True) == 0 and dlrm._late_layers(32768, True) == 0
    assert dlrm._HALF_BATCH_MIN == 32768 or "TORCHREC_AMD_HALF_BATCH_MIN" in __import__("os").environ
    assert dlrm._WGRAD_SPLIT_MODE in ("late", "early") and dlrm._GRAPH_EXCHANGE in (False, True)
