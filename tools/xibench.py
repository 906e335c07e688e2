"""Timing of the fused dot-interaction kernels at the Criteo shape (development tool).
TBE_INTERACTION_ABLATION=1 (no output stores) / 2 (no MFMA) switch the forward kernel's tuning variants."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import _paths  # noqa: E402,F401
from torchrec_amd.models.dlrm import _FusedDotInteraction  # noqa: E402


def main():
    B, F, D = 65536, 26, 128
    dense = torch.randn(B, D, device="cuda", requires_grad=True)
    sparse = torch.randn(B, F, D, device="cuda", requires_grad=True)
    g = torch.randn(B, D + (F + 1) * F // 2, device="cuda")

    def t(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / n * 1e3

    pad = os.environ.get("XIBENCH_PAD", "0") == "1"
    if pad:
        g = torch.randn(B, D + ((F + 1) * F // 2 + 3) // 4 * 4, device="cuda")[:, :D + (F + 1) * F // 2]
    with torch.no_grad():
        f = t(lambda: _FusedDotInteraction.apply(dense, sparse, pad))
    out = _FusedDotInteraction.apply(dense, sparse, pad)
    fb = t(lambda: torch.autograd.grad(_FusedDotInteraction.apply(dense, sparse, pad), (dense, sparse), g))
    rd, wr = B * (F + 1) * D * 4, B * (D + (F + 1) * F // 2) * 4
    print(f"pad_rows={pad} ablation={os.environ.get('TBE_INTERACTION_ABLATION', '0')}: forward {f:.1f} us ({(rd + wr) / f / 1e6:.2f} TB/s of "
          f"{(rd + wr) / 1e9:.2f} GB), forward+backward {fb:.1f} us")


if __name__ == "__main__":
    main()
