"""Development probe: runs bench.main() with a few torch calls wrapped by wall-clock timers and prints the
caching-allocator counters, to find host-side stalls in the train step."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

acc = {}


def wrap(owner, name, label=None):
    fn = getattr(owner, name)
    label = label or name

    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        d = time.perf_counter() - t0
        c = acc.setdefault(label, [0, 0.0, 0.0])
        c[0] += 1
        c[1] += d
        c[2] = max(c[2], d)
        return r
    setattr(owner, name, w)


wrap(torch.Tensor, "index_select")
wrap(torch, "empty")
wrap(torch.Tensor, "record_stream")
import torch.distributed as dist  # noqa: E402
wrap(dist, "all_to_all_single")
wrap(dist, "all_reduce")
import bench  # noqa: E402

s0 = None
orig_enable = None
bench.main()
st = torch.cuda.memory_stats()
print("device allocs (hipMalloc):", st.get("num_device_alloc"), "frees:", st.get("num_device_free"), "retries:", st.get("num_alloc_retries"))
for k, (n, tot, mx) in acc.items():
    print(f"{k:20s} calls {n:6d}  avg {tot / n * 1e6:8.1f} us  max {mx * 1e6:9.1f} us  total {tot * 1e3:8.1f} ms")
