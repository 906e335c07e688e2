"""fp32 GEMM timing probe for the DLRM dense layers (development tool): is the 479-wide interaction
output (row stride 1916 B, not 16-B aligned) slower than a 480-wide one?"""
import torch


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


def main():
    dev = "cuda"
    for B in (65536, 8192):
        for K in (479, 480, 512):
            N = 1024
            x = torch.randn(B, K, device=dev)
            w = torch.randn(N, K, device=dev)
            b = torch.randn(N, device=dev)
            gy = torch.randn(B, N, device=dev)
            fl = 2 * B * K * N
            f = t(lambda: torch._addmm_activation(b, x, w.t(), use_gelu=False))
            d = t(lambda: gy @ w)
            g = t(lambda: gy.t() @ x)
            print(f"B={B} K={K}: fwd {f:7.1f} us ({fl / f / 1e6:6.1f} TF)  dgrad {d:7.1f} us ({fl / d / 1e6:6.1f} TF)  "
                  f"wgrad {g:7.1f} us ({fl / g / 1e6:6.1f} TF)")
        # strided view: logical K = 479 inside a 480-wide buffer
        xp = torch.randn(B, 480, device=dev)
        wp = torch.randn(1024, 480, device=dev)
        xv, wv = xp[:, :479], wp[:, :479]
        f = t(lambda: torch.addmm(b, xv, wv.t()))
        print(f"B={B} K=479 views of 480-stride buffers: fwd {f:7.1f} us")


if __name__ == "__main__":
    main()
