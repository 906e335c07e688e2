"""Development probe: does the HOST call of a collective (one-rank RCCL group) wait for work already queued on the compute
stream?  Times the host side of all_to_all_single / all_reduce / plain cross-stream event hand-over behind N queued GEMMs."""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29534")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
pg = dist.group.WORLD
a = torch.randn(1 << 20, device=dev)
b = torch.empty_like(a)
x0 = torch.randn(4096, 4096, device=dev)
side = torch.cuda.Stream()


def busy(n):
    x = x0
    for _ in range(n):
        x = x @ x0
    return x


def probe(name, fn, n_gemm):
    out = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        busy(n_gemm)
        t1 = time.perf_counter()
        w = fn()
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        out.append((round((t1 - t0) * 1e6), round((t2 - t1) * 1e6), round((t3 - t2) * 1e6)))
        del w
    print(f"{name:34s} gemms={n_gemm:3d}  (enqueue gemms us, host call us, drain us): {out}")


def xstream():
    ev = torch.cuda.Event()
    ev.record()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        b.copy_(a)
        ev2 = torch.cuda.Event()
        ev2.record()
    torch.cuda.current_stream().wait_event(ev2)
    return ev2


# the same behind a replayed HIP graph (the train step's dense segments are graphs)
g = torch.cuda.CUDAGraph()
xg = torch.randn(4096, 4096, device=dev)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for _ in range(3):
        y = xg @ x0
torch.cuda.synchronize()
with torch.cuda.graph(g, stream=st):
    y = xg
    for _ in range(10):
        y = y @ x0


def probe_graph(name, fn, replays):
    out = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(replays):
            g.replay()
        t1 = time.perf_counter()
        w = fn()
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        out.append((round((t1 - t0) * 1e6), round((t2 - t1) * 1e6), round((t3 - t2) * 1e6)))
        del w
    print(f"{name:34s} graph replays={replays}  (enqueue us, host call us, drain us): {out}")


for r in (1, 4):
    probe_graph("all_to_all_single async", lambda: dist.all_to_all_single(b, a, [1 << 20], [1 << 20], group=pg, async_op=True), r)
    probe_graph("all_reduce async", lambda: dist.all_reduce(a, group=pg, async_op=True), r)
    probe_graph("event hand-over + copy on side", xstream, r)
    probe_graph("copy_ same stream", lambda: b.copy_(a), r)
    probe_graph("graph replay again", lambda: g.replay(), r)
for n in (0, 10):
    probe("all_to_all_single async", lambda: dist.all_to_all_single(b, a, [1 << 20], [1 << 20], group=pg, async_op=True), n)
    probe("all_reduce async", lambda: dist.all_reduce(a, group=pg, async_op=True), n)
    probe("event hand-over + copy on side", xstream, n)
    probe("copy_ same stream", lambda: b.copy_(a), n)
dist.destroy_process_group()
