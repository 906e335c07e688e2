"""Development probe (2): which ingredient makes the ids all-to-all's HOST call wait ~0.4 ms inside the train step?"""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29535")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
pg = dist.group.WORLD
big_a = torch.randn(8192 * 3328, device=dev)
big_b = torch.empty_like(big_a)
ids = torch.randint(0, 1000, (26 * 8192,), device=dev)
idb = torch.empty_like(ids)
x0 = torch.randn(4096, 4096, device=dev)
side = torch.cuda.Stream()


def busy(n):
    x = x0
    for _ in range(n):
        x = x @ x0
    return x


def t(fn):
    t0 = time.perf_counter()
    r = fn()
    return round((time.perf_counter() - t0) * 1e6), r


def scenario(name, n_gemm, first_on_main, side_kernel_first, from_side=True):
    out = []
    for _ in range(5):
        torch.cuda.synchronize()
        a_us = 0
        if first_on_main:
            a_us, w0 = t(lambda: dist.all_to_all_single(big_b, big_a, group=pg, async_op=True))
        busy(n_gemm)
        ev = torch.cuda.Event()
        ev.record()
        if from_side:
            with torch.cuda.stream(side):
                side.wait_event(ev)
                if side_kernel_first:
                    y = ids + 1
                us, w = t(lambda: dist.all_to_all_single(idb, ids, [ids.numel()], [ids.numel()], group=pg, async_op=True))
        else:
            us, w = t(lambda: dist.all_to_all_single(idb, ids, [ids.numel()], [ids.numel()], group=pg, async_op=True))
        torch.cuda.synchronize()
        out.append((a_us, us))
    print(f"{name:70s} (first a2a host us, ids a2a host us): {out}", flush=True)


scenario("ids a2a from side stream, idle GPU", 0, False, False)
scenario("ids a2a from side stream behind 10 GEMMs on main", 10, False, False)
scenario("... with a pooled a2a on main first", 10, True, False)
scenario("... with a pooled a2a on main first + a kernel on the side stream first", 10, True, True)
scenario("ids a2a from MAIN behind pooled a2a + 10 GEMMs", 10, True, False, from_side=False)
# ---- behind a replayed HIP graph on the main stream, from a side stream that depends on an EARLIER event ----------------
g = torch.cuda.CUDAGraph()
st = torch.cuda.Stream()
xg = torch.randn(4096, 4096, device=dev)
with torch.cuda.stream(st):
    for _ in range(3):
        y = xg @ x0
torch.cuda.synchronize()
with torch.cuda.graph(g, stream=st):
    y = xg
    for _ in range(10):
        y = y @ x0


def graph_case(name, from_side, with_kernel, small_copy_first=False):
    out = []
    for _ in range(5):
        torch.cuda.synchronize()
        ev = torch.cuda.Event()
        ev.record()
        g.replay()
        if from_side:
            with torch.cuda.stream(side):
                side.wait_event(ev)
                if with_kernel:
                    y2 = ids + 1
                us, w = t(lambda: dist.all_to_all_single(idb, ids, [ids.numel()], [ids.numel()], group=pg, async_op=True))
        else:
            us, w = t(lambda: dist.all_to_all_single(idb, ids, [ids.numel()], [ids.numel()], group=pg, async_op=True))
        t0 = time.perf_counter()
        torch.cuda.synchronize()
        out.append((us, round((time.perf_counter() - t0) * 1e6)))
    print(f"{name:70s} (ids a2a host us, drain us): {out}", flush=True)


graph_case("graph replay on main, ids a2a from main", False, False)
graph_case("graph replay on main, ids a2a from side (event before the graph)", True, False)
graph_case("graph replay on main, kernel + ids a2a from side", True, True)
dist.destroy_process_group()
