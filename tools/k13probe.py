"""Development probe: cost of the DLRM bottom MLP's first layer (13 -> 512, K = 13) through the BLAS libraries."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import _paths  # noqa: E402,F401
from torchrec_amd import tuning  # noqa: E402

tuning.enable_tuned_gemms()


def t(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for B in (65536, 8192):
    x = torch.randn(B, 13, device="cuda")
    w = torch.randn(512, 13, device="cuda")
    b = torch.randn(512, device="cuda")
    gy = torch.randn(B, 512, device="cuda")
    print(f"B={B}: fwd addmm+relu {t(lambda: torch._addmm_activation(b, x, w.t(), use_gelu=False)):.1f} us "
          f"(write {B * 512 * 4 / 1e6:.0f} MB)")
    for c in (1, 4, 8, 16, 32):
        if B % c:
            continue
        f = (lambda c=c: torch.bmm(gy.view(c, B // c, -1).transpose(1, 2), x.view(c, B // c, -1)).sum(dim=0)) if c > 1 else (lambda: gy.t() @ x)
        print(f"   wgrad chunks={c}: {t(f):.1f} us (read {B * 512 * 4 / 1e6:.0f} MB)")
