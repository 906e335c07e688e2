"""ctypes binding of the gfx950 C-ABI library declared in ``include/tbe_hip.h``.

This is the host side of the drop-in boundary: the library is built in-tree by
``__graft_entry__.build()`` (``torchrec-oldfork_amd/csrc/Makefile``) and loaded here.
There is NO fallback: if the shared object is missing, import fails loudly; if it is
called with non-device tensors, the wrappers raise.
"""
import ctypes
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.normpath(os.path.join(_HERE, "..", "lib", "libtbe_hip.so"))

c_void_p = ctypes.c_void_p
c_i32 = ctypes.c_int32
c_i64 = ctypes.c_int64
c_size = ctypes.c_size_t
c_float = ctypes.c_float


class OptimizerArgs(ctypes.Structure):
    """Mirror of ``tbe_optimizer_args`` (include/tbe_hip.h)."""

    _fields_ = [
        ("optimizer", c_i32),
        ("learning_rate", c_float),
        ("eps", c_float),
        ("weight_decay", c_float),
        ("beta1", c_float),
        ("beta2", c_float),
        ("iteration", c_i64),
    ]


class CacheDesc(ctypes.Structure):
    """Mirror of ``tbe_cache_desc`` (include/tbe_hip.h)."""

    _fields_ = [
        ("tags", c_void_p), ("lru", c_void_p), ("rows", c_void_p), ("state", c_void_p),
        ("staging_keys", c_void_p), ("counters", c_void_p), ("tab_key_base", c_void_p),
        ("tab_weights", c_void_p), ("tab_state", c_void_p), ("tab_D", c_void_p),
        ("num_sets", c_i32), ("row_stride", c_i32), ("staging_cap", c_i32), ("num_tables", c_i32),
    ]


# name -> (restype, argtypes); must list every symbol of include/tbe_hip.h
SIGNATURES = {
    "tbe_last_error": (ctypes.c_char_p, []),
    "tbe_abi_version": (c_i32, []),
    "tbe_fault_status": (ctypes.c_int, [ctypes.POINTER(c_i64)]),
    "tbe_debug_inject_fault_host": (ctypes.c_int, []),
    "tbe_debug_inject_sort_giveup": (ctypes.c_int, [c_void_p]),
    "tbe_profile_enable": (ctypes.c_int, [c_i32]),
    "tbe_profile_read": (ctypes.c_int, [c_i32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_i64)]),
    "tbe_profile_read_rows": (ctypes.c_int, [ctypes.POINTER(c_i64)]),
    "tbe_forward_pooled_f32": (
        ctypes.c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_i32, c_i32, c_i32, c_void_p, c_i64,
         c_void_p, c_void_p, c_i32, c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_void_p],
    ),
    "tbe_forward_nobag_f32": (
        ctypes.c_int,
        [c_void_p, c_void_p, c_i32, c_i32, c_i32, c_void_p, c_i64, c_void_p, c_void_p, c_void_p,
         c_void_p],
    ),
    "tbe_backward_workspace_bytes": (c_size, [c_i64, c_i32, c_i32, c_i32, c_i32]),
    "tbe_backward_fused_f32": (
        ctypes.c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i32, c_i32,
         c_i32, c_i32, c_void_p, c_i64, c_void_p, c_void_p, c_i32, c_void_p, c_void_p, c_i64, OptimizerArgs,
         c_i32, c_void_p, c_size, c_void_p, c_void_p, c_void_p],
    ),
    "tbe_backward_prepare": (
        ctypes.c_int,
        [c_void_p, c_void_p, c_i32, c_i32, c_i32, c_i32, c_void_p, c_i64, c_void_p, c_i32, c_i32, c_void_p,
         c_size, c_void_p, c_void_p, c_void_p],
    ),
    "tbe_sort_pairs_workspace_bytes": (c_size, [c_i64, c_i32]),
    "tbe_sort_pairs": (
        ctypes.c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_i32, c_i32, c_i32, c_void_p, c_size, c_void_p],
    ),
    "tbe_debug_sort_timeouts": (ctypes.c_int, [ctypes.POINTER(c_i64)]),
    "tbe_debug_set_sort_stamps": (ctypes.c_int, [c_void_p]),
    "tbe_debug_set_interaction_stamps": (ctypes.c_int, [c_void_p]),
    "tbe_backward_apply_f32": (
        ctypes.c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i32, c_i32,
         c_i32, c_i32, c_void_p, c_i64, c_void_p, c_void_p, c_i32, c_void_p, c_void_p, c_i64, OptimizerArgs,
         c_i32, c_void_p, c_size, c_void_p],
    ),
    "tbe_cache_prefetch_workspace_bytes": (c_size, [c_i64, c_i32]),
    "tbe_cache_prefetch": (
        ctypes.c_int,
        [ctypes.POINTER(CacheDesc), c_void_p, c_void_p, c_i32, c_i32, c_void_p, c_i64, c_void_p, c_i32, c_i32,
         c_void_p, c_void_p, c_size, c_void_p, c_void_p],
    ),
    "tbe_cache_writeback_staging": (ctypes.c_int, [ctypes.POINTER(CacheDesc), c_void_p]),
    "tbe_cache_flush": (ctypes.c_int, [ctypes.POINTER(CacheDesc), c_i32, c_void_p]),
    "tbe_cumsum_workspace_bytes": (c_size, [c_i64]),
    "tbe_cumsum": (ctypes.c_int, [c_void_p, c_void_p, c_i64, c_i32, c_i32, c_void_p, c_size, c_void_p]),
    "tbe_permute_2d_workspace_bytes": (c_size, [c_i32, c_i32, c_i32]),
    "tbe_permute_2d_lengths": (
        ctypes.c_int,
        [c_void_p, c_i32, c_i32, c_i32, c_void_p, c_i32, c_void_p, c_void_p, c_void_p, c_void_p,
         c_size, c_void_p],
    ),
    "tbe_permute_2d_data": (
        ctypes.c_int,
        [c_void_p, c_i32, c_i32, c_void_p, c_void_p, c_void_p, c_void_p, c_i32, c_void_p, c_void_p,
         c_i32, c_void_p],
    ),
    "tbe_bucketize_workspace_bytes": (c_size, [c_i64, c_i32]),
    "tbe_block_bucketize": (
        ctypes.c_int,
        [c_void_p, c_i32, c_i64, c_void_p, c_i32, c_i64, c_void_p, c_i32, c_i32, c_void_p, c_i32,
         c_i32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size, c_void_p],
    ),
    "tbe_a2a_pooled_unpack": (
        ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_i32, c_i32, c_i32, c_i32, c_float, c_void_p]),
    "tbe_a2a_pooled_pack": (
        ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_i32, c_i32, c_i32, c_i32, c_float, c_void_p]),
    "tbe_pooled_exchange_unpack": (
        ctypes.c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i32, c_i32, c_i32,
         c_i32, c_i32, c_float, c_void_p]),
    "tbe_pooled_exchange_pack": (
        ctypes.c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i32, c_i32, c_i32,
         c_i32, c_i32, c_float, c_void_p]),
    "tbe_dlrm_interaction_forward_f32": (
        ctypes.c_int, [c_void_p, c_void_p, c_i32, c_i32, c_i32, c_void_p, c_i64, c_void_p]),
    "tbe_dlrm_interaction_backward_f32": (
        ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i32, c_i32, c_i32, c_void_p, c_void_p, c_void_p]),
    "tbe_relu_backward_bias_grad_workspace_bytes": (c_size, [c_i64, c_i32]),
    "tbe_relu_backward_bias_grad_f32": (
        ctypes.c_int, [c_void_p, c_void_p, c_i64, c_i32, c_void_p, c_void_p, c_void_p, c_size, c_void_p]),
    "tbe_weighted_colsum_workspace_bytes": (c_size, [c_i64, c_i32]),
    "tbe_weighted_colsum_f32": (
        ctypes.c_int, [c_void_p, c_void_p, c_i64, c_i32, c_void_p, c_void_p, c_size, c_void_p]),
    "tbe_colsum_row_blocks": (c_i64, [c_i64, c_i32]),
    "tbe_relu_backward_bias_partials_f32": (
        ctypes.c_int, [c_void_p, c_void_p, c_i64, c_i32, c_void_p, c_void_p, c_size, c_void_p]),
    "tbe_weighted_colsum_partials_f32": (
        ctypes.c_int, [c_void_p, c_void_p, c_i64, c_i32, c_void_p, c_size, c_void_p]),
    "tbe_multi_chunk_sum_f32": (ctypes.c_int, [c_void_p, c_i32, c_i64, c_void_p, c_float, c_void_p]),
    "tbe_multi_chunk_sum_host_table_f32": (ctypes.c_int, [c_void_p, c_i32, c_i64, c_void_p, c_float, c_void_p]),
    "tbe_bce_with_logits_workspace_bytes": (c_size, []),
    "tbe_bce_with_logits_f32": (
        ctypes.c_int, [c_void_p, c_void_p, c_i32, c_i64, c_void_p, c_void_p, c_void_p, c_size, c_void_p]),
    "tbe_jagged_2d_to_dense_f32": (
        ctypes.c_int, [c_void_p, c_void_p, c_i32, c_i32, c_i32, c_void_p, c_void_p]),
    "tbe_dense_to_jagged_2d_f32": (
        ctypes.c_int, [c_void_p, c_void_p, c_i32, c_i32, c_i32, c_i64, c_void_p, c_void_p]),
    "tbe_offsets_range": (ctypes.c_int, [c_void_p, c_i64, c_i64, c_void_p, c_void_p]),
    "tbe_copy_rows": (ctypes.c_int, [c_void_p, c_i64, c_void_p, c_i32, c_i64, c_void_p, c_void_p]),
}

_lib: Optional[ctypes.CDLL] = None


def load() -> ctypes.CDLL:
    """Loads the in-tree HIP library; raises ImportError (never falls back) if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the MI355X HIP extension has not been built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` at the repo root "
            "(or `make -C torchrec-oldfork_amd/csrc`). There is no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().tbe_last_error()
        raise RuntimeError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


class KernelFaultError(RuntimeError):
    """A kernel of this library gave up on something its result depends on (include/tbe_hip.h `tbe_fault_status`)."""


_faults_seen = 0


def fault_count() -> int:
    """Sort give-ups reported so far in this process (no sync: what has arrived in the fault word)."""
    n = c_i64(0)
    check(load().tbe_fault_status(ctypes.byref(n)), "tbe_fault_status")
    return int(n.value)


def raise_on_faults(where: str) -> None:
    """Raises KernelFaultError if a kernel reported a fault since the last check (a delta: one raise per batch of
    faults).  Costs one ctypes call, no sync — cheap enough for every optimizer step.  Called from the points where the
    reference talks to its TBE module outside forward / backward (set_learning_rate in step / zero_grad:
    torchrec/distributed/batched_embedding_kernel.py:250-257; flush before state_dict: :563) and from
    bounds_check_errors() / split_embedding_weights()."""
    global _faults_seen
    n = fault_count()
    if n > _faults_seen:
        new, _faults_seen = n - _faults_seen, n
        raise KernelFaultError(
            f"{where}: {new} spin-wait give-up(s) inside the embedding pair sort since the last check — a predecessor "
            "workgroup did not publish its histogram in time (GPU shared with another process, a debugger or profiler "
            "pause?).  The sorted order, and with it the row updates (or the row cache's state), of at least one "
            "backward / prefetch since then are WRONG; restore the tables from a checkpoint.")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device address of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def require_gpu(*tensors: Optional[torch.Tensor]) -> torch.device:
    """All given tensors must live on one HIP device; returns it."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "fbgemm_gpu (MI355X build): tensor on device "
                f"'{t.device}' — this path only runs on a HIP device; there is no CPU fallback"
            )
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"tensors on different devices: {dev} vs {t.device}")
    if dev is None:
        raise RuntimeError("no tensor given")
    return dev


def stream_ptr(dev: torch.device) -> int:
    """hipStream_t of torch's current stream on `dev` (the raw-handle call: torch.cuda.current_stream() builds a Stream
    object per call, ~8 us of host time, and a train step asks a dozen times)."""
    idx = dev.index
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device() if idx is None else idx)


def workspace(nbytes: int, dev: torch.device) -> torch.Tensor:
    """256-B aligned scratch from the caching allocator (stream-ordered on the current stream)."""
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)
