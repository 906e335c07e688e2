"""Process-group set-up for one process per GPU over RCCL.

Why a helper: HIP maps every stream of a process onto a few hardware queues (GPU_MAX_HW_QUEUES of them per priority level),
and two streams on one hardware queue execute in order.  ProcessGroupNCCL takes its internal stream from torch's pool, so
by default the collective's stream can land on the hardware queue of the compute stream: a pooled all-to-all that waits for
its link then holds back every compute kernel enqueued behind it (kernel trace, `Queue_Id` column — the RCCL
kernels and the default stream's kernels on one queue; both all-to-alls of a step fully exposed).  Hardware queues are kept
per priority level, so a HIGH-PRIORITY collective stream never shares one with the normal-priority compute streams — and a
collective should start as soon as its input is there anyway."""
import os
from typing import Optional

import torch
import torch.distributed as dist


def rccl_options(high_priority: Optional[bool] = None):
    """ProcessGroupNCCL.Options for `init_process_group(..., pg_options=)` / `new_group(..., pg_options=)`.
    high_priority None: TORCHREC_AMD_RCCL_HIGH_PRIORITY (default 1)."""
    if high_priority is None:
        high_priority = os.environ.get("TORCHREC_AMD_RCCL_HIGH_PRIORITY", "1") == "1"
    opts = dist.ProcessGroupNCCL.Options()
    opts.is_high_priority_stream = bool(high_priority)
    return opts


def init_rccl_process_group(device: torch.device, high_priority: Optional[bool] = None, **kwargs) -> None:
    """`dist.init_process_group("nccl", device_id=device, ...)` with the collective stream on its own hardware queue."""
    dist.init_process_group("nccl", device_id=device, pg_options=rccl_options(high_priority), **kwargs)


def new_rccl_group(process_group=None, high_priority: Optional[bool] = None):
    """A second communicator over the ranks of `process_group` (default: the world), with its own (high-priority) collective
    stream: collectives issued on it never queue behind those of the first one — DLRMTrain puts its dense gradient
    all-reduces there, clear of the pooled all-to-alls.  Collective: every rank of the default group must call it."""
    ranks = dist.get_process_group_ranks(process_group if process_group is not None else dist.group.WORLD)
    return dist.new_group(ranks=ranks, pg_options=rccl_options(high_priority))


_exchange_groups = {}


def exchange_group(process_group, device) -> Optional["dist.ProcessGroup"]:
    """The group a sharded module's id / pooled exchanges should run on: `process_group` itself when it is not RCCL or its
    collective stream is already high-priority (init_rccl_process_group), else a second communicator over the same ranks
    whose stream is (new_rccl_group; one per process group, shared by every sharded module).  This is what lets a launcher
    written for CUDA — `dist.init_process_group(backend="nccl")`, examples/dlrm/dlrm_main.py:469-478 — run unmodified without
    the exchange landing on the compute stream's hardware queue.  Collective when it creates the group: sharded modules are
    built by every rank.  TORCHREC_AMD_OWN_EXCHANGE_GROUP=0: always `process_group`."""
    if process_group is None or os.environ.get("TORCHREC_AMD_OWN_EXCHANGE_GROUP", "1") != "1":
        return process_group
    try:
        if dist.get_backend(process_group) != "nccl":
            return process_group
        if bool(process_group._get_backend(torch.device(device)).options.is_high_priority_stream):
            return process_group
    except Exception:  # cannot tell: leave the caller's group alone
        return process_group
    key = id(process_group)
    hit = _exchange_groups.get(key)
    if hit is not None:
        parent, own = hit
        alive = getattr(dist.distributed_c10d, "_world", None)
        if parent is process_group and (alive is None or own in alive.pg_map):  # (not a new group at a recycled address,
            return own                                                         #  not one destroy_process_group() took down)
    own = new_rccl_group(process_group, high_priority=True)
    _exchange_groups[key] = (process_group, own)
    return own
