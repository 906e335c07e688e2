"""Development probe: do kernels on two HIP streams overlap on this box, or do some stream pairs share a hardware queue
(GPU_MAX_HW_QUEUES) and serialise?  Main stream: 24 fp32 GEMMs (4096^3); side stream k: a spin kernel of ~2 ms."""
import os
import time

import torch

dev = torch.device("cuda", 0)
x = torch.randn(4096, 4096, device=dev)
torch.cuda._sleep(1000)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); torch.cuda._sleep(10_000_000); b.record(); b.synchronize()
cyc_per_us = 10_000_000 / (a.elapsed_time(b) * 1e3)


def gemms():
    y = x
    for _ in range(24):
        y = y @ x
    return y


def wall(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


gemms()
t_g = min(wall(gemms) for _ in range(3))
t_s = min(wall(lambda: torch.cuda._sleep(int(2000 * cyc_per_us))) for _ in range(3))
print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', 'default')}: gemms alone {t_g:.2f} ms, spin alone {t_s:.2f} ms")
streams = [torch.cuda.Stream(dev) for _ in range(10)]
for k, s in enumerate(streams):
    def both():
        with torch.cuda.stream(s):
            torch.cuda._sleep(int(2000 * cyc_per_us))
        gemms()
    t = min(wall(both) for _ in range(3))
    print(f"  side stream {k}: together {t:.2f} ms -> {'overlaps' if t < t_g + 0.5 * t_s else 'SERIALISED with the compute stream'}")

# ---- the same question for the collective's stream: a one-rank RCCL all-to-all (a 1-GiB local copy) beside the GEMMs ------
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29536")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
src = torch.randn(1 << 28, device=dev)
dst = torch.empty_like(src)
dist.all_to_all_single(dst, src)
torch.cuda.synchronize()
t_a = min(wall(lambda: dist.all_to_all_single(dst, src, async_op=True)) for _ in range(3))


def both_a2a():
    w = dist.all_to_all_single(dst, src, async_op=True)
    gemms()
    w.wait()


t = min(wall(both_a2a) for _ in range(3))
print(f"  RCCL all-to-all alone {t_a:.2f} ms; beside the gemms {t:.2f} ms (gemms alone {t_g:.2f}) -> "
      f"{'overlaps' if t < t_g + 0.5 * t_a else 'SERIALISED with the compute stream'}")
dist.destroy_process_group()
