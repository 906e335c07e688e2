"""Development probe: host-side cost (us per call, no GPU wait) of torch.distributed collectives over a one-rank RCCL group."""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
pg = dist.group.WORLD


def cost(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return t


for numel in (1 << 10, 213_000, 1 << 20, 15_728_640):
    a = torch.randn(numel, device=dev)
    b = torch.empty_like(a)
    ids = torch.randint(0, 1000, (numel,), device=dev)
    idb = torch.empty_like(ids)
    print(f"numel {numel}:")
    print("   all_to_all_single async   ", round(cost(lambda: dist.all_to_all_single(b, a, [numel], [numel], group=pg, async_op=True)), 1))
    print("   all_to_all_single sync    ", round(cost(lambda: dist.all_to_all_single(b, a, [numel], [numel], group=pg)), 1))
    print("   all_to_all_single no split", round(cost(lambda: dist.all_to_all_single(b, a, group=pg, async_op=True)), 1))
    print("   all_to_all_single int64   ", round(cost(lambda: dist.all_to_all_single(idb, ids, [numel], [numel], group=pg, async_op=True)), 1))
    print("   all_reduce async          ", round(cost(lambda: dist.all_reduce(a, group=pg, async_op=True)), 1))
    print("   copy_ (D2D)               ", round(cost(lambda: b.copy_(a)), 1))
side = torch.cuda.Stream()
a = torch.randn(1 << 20, device=dev)
b = torch.empty_like(a)


def with_busy_queue():
    # the compute stream has a long kernel queued: does the collective's host call wait for it?
    x = torch.randn(4096, 4096, device=dev)
    for _ in range(10):
        x = x @ x
    t0 = time.perf_counter()
    dist.all_to_all_single(b, a, [1 << 20], [1 << 20], group=pg, async_op=True)
    return (time.perf_counter() - t0) * 1e6


torch.cuda.synchronize()
print("all_to_all_single behind 10 queued GEMMs:", [round(with_busy_queue(), 1) for _ in range(5)])
dist.destroy_process_group()
