"""Numerical audit of the recorded GEMM choices (torchrec_amd/tuning): every DLRM dense layer shape,
forward + dgrad + wgrad, library default vs replayed choice, both against a float64 reference.
Usage: python tools/check_tuned_gemms.py [csv]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import _paths  # noqa: E402,F401
import torch.cuda.tunable as tunable  # noqa: E402
from torchrec_amd.modules.mlp import Perceptron  # noqa: E402
from torchrec_amd.tuning import _FILE, enable_tuned_gemms  # noqa: E402

LAYERS = [(13, 512), (512, 256), (256, 128), (479, 1024), (1024, 1024), (1024, 512), (512, 256)]


def run(p, x, g):
    xi = x.clone().requires_grad_()
    p.zero_grad()
    y = p(xi)
    y.backward(g)
    return y.detach(), xi.grad, p._linear.weight.grad.clone(), p._linear.bias.grad.clone()


def ref64(p, x, g, y32):
    """float64 reference; the ReLU mask is the one the fp32 path itself produced (a pre-activation within
    rounding of 0 may legitimately land on either side)."""
    w, b = p._linear.weight.detach().double(), p._linear.bias.detach().double()
    xd, gd = x.double(), g.double()
    y = torch.relu(xd @ w.t() + b)
    gm = gd * (y32 > 0)
    return y, gm @ w, gm.t() @ xd, gm.sum(0)


def err(a, r):
    # error in units of the result's typical magnitude
    return float((a.double() - r).abs().max() / (r.abs().mean() + 1e-30))


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else _FILE
    torch.manual_seed(0)
    worst = 0.0
    for B in (65536, 32768, 16384, 8192, 4096):
        for i, o in LAYERS:
            p = Perceptron(i, o, device=torch.device("cuda"))
            x = torch.randn(B, i, device="cuda")
            g = torch.randn(B, o, device="cuda")
            tunable.enable(False)
            d = run(p, x, g)
            assert enable_tuned_gemms(path)
            t = run(p, x, g)
            tunable.enable(False)
            ed = [err(a, b) for a, b in zip(d, ref64(p, x, g, d[0]))]
            et = [err(a, b) for a, b in zip(t, ref64(p, x, g, t[0]))]
            flag = "  <-- BAD" if max(et) > 10 * max(max(ed), 1e-5) else ""
            worst = max(worst, max(et))
            print(f"B={B:6d} {i:4d}->{o:4d}  default y/dx/dw/db {ed[0]:.1e} {ed[1]:.1e} {ed[2]:.1e} {ed[3]:.1e}   "
                  f"tuned {et[0]:.1e} {et[1]:.1e} {et[2]:.1e} {et[3]:.1e}{flag}", flush=True)
    print("worst tuned error (relative to mean magnitude):", worst)


if __name__ == "__main__":
    main()
