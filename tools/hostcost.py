"""Development probe: host-side cost (us per call, no GPU wait) of a few ops used per step."""
import time
import torch

dev = torch.device("cuda", 0)
v = torch.randint(0, 1000, (26, 8192), device=dev)
idx = torch.tensor([3, 5, 7, 9, 11, 13, 15, 17, 19, 21, 23, 25, 0, 1, 2], device=dev)
side = torch.cuda.Stream()


def cost(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return t


print("index_select          ", cost(lambda: v.index_select(0, idx)))
print("advanced index        ", cost(lambda: v[idx]))
def on_side():
    with torch.cuda.stream(side):
        v.index_select(0, idx)
print("index_select on side  ", cost(on_side))
def ctx_only():
    with torch.cuda.stream(side):
        pass
print("stream ctx only       ", cost(ctx_only))
x = torch.randn(8192, 512, device=dev)
print("empty alloc           ", cost(lambda: torch.empty(8192 * 3328, device=dev)))
print("event create+record   ", cost(lambda: torch.cuda.Event().record()))
print("tensor.float()        ", cost(lambda: idx.float()))
