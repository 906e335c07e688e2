"""Side streams that really run beside the compute stream.

HIP maps the streams of a process onto a few hardware queues (GPU_MAX_HW_QUEUES per priority level; torch's stream pool
hands out 32 streams per priority) and two streams on one hardware queue execute IN ORDER: a backward sort "overlapped" on a
side stream that shares the default stream's queue is simply inline, and a side stream that waits (for a copy, a collective)
holds the compute kernels behind it back.  Which pool stream lands on which queue follows creation order and is not visible
through the HIP API, but it is measurable: `shares_hw_queue` runs a ~0.3-ms spin kernel on one stream and a tiny kernel on
the other and looks at when the tiny one finished (rocprofv3's kernel trace shows the same in its Queue_Id
column).  `side_stream` draws pool streams until it finds one on another queue than the device's default stream (and,
if it can, than the side streams handed out before).  TBE_STREAM_PROBE=0: take the first pool stream, as before."""
import os
from typing import Dict, List, Optional

import torch

_PROBE = os.environ.get("TBE_STREAM_PROBE", "1") == "1"
_cycles: Dict[int, int] = {}
_handed_out: Dict[int, List[torch.cuda.Stream]] = {}
_scratch: Dict[int, torch.Tensor] = {}


def _spin_cycles(dev: torch.device, us: float = 300.0) -> int:
    c = _cycles.get(dev.index)
    if c is None:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s = torch.cuda.current_stream(dev)
        torch.cuda._sleep(1000)
        s.synchronize()
        a.record(s)
        torch.cuda._sleep(2_000_000)
        b.record(s)
        b.synchronize()
        c = _cycles[dev.index] = max(int(2_000_000 / (a.elapsed_time(b) * 1e3) * us), 1000)
    return c


def shares_hw_queue(a: torch.cuda.Stream, b: torch.cuda.Stream) -> bool:
    """True if work on `b` waits for earlier work on `a` although nothing orders them (measured; both streams are drained
    first, ~0.4 ms)."""
    dev = a.device
    with torch.cuda.device(dev):
        cycles = _spin_cycles(dev)
        x = _scratch.get(dev.index)
        if x is None:
            x = _scratch[dev.index] = torch.zeros(64, device=dev)
        a.synchronize()
        b.synchronize()
        s0, s1, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        with torch.cuda.stream(a):
            s0.record(a)
            torch.cuda._sleep(cycles)
            s1.record(a)
        with torch.cuda.stream(b):
            x.add_(1.0)
            e1.record(b)
        s1.synchronize()
        e1.synchronize()
        return s0.elapsed_time(e1) > 0.5 * s0.elapsed_time(s1)


def side_stream(device: torch.device, priority: int = 0, tries: int = 12) -> torch.cuda.Stream:
    """A stream of the pool that does not share a hardware queue with the device's default stream (where the compute
    runs) nor, if possible, with the side streams handed out before."""
    device = torch.device(device)
    if not _PROBE or torch.cuda.is_current_stream_capturing():
        return torch.cuda.Stream(device, priority=priority)
    with torch.cuda.device(device):
        default = torch.cuda.default_stream(device)
        others = _handed_out.setdefault(device.index, [])
        fallback: Optional[torch.cuda.Stream] = None
        for _ in range(tries):
            s = torch.cuda.Stream(device, priority=priority)
            if shares_hw_queue(s, default):
                continue
            if fallback is None:
                fallback = s
            if not any(shares_hw_queue(s, o) for o in others[-2:]):
                fallback = s
                break
        s = fallback if fallback is not None else torch.cuda.Stream(device, priority=priority)
        others.append(s)
        return s
