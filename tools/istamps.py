"""Where a wave of the LDS-DMA interaction forward spends a sample: cycle-counter stamps written by
interaction_fwd_glds_kernel<128, 6> (TBE_INTERACTION_ABLATION=6 + tbe_debug_set_interaction_stamps).
Prints the median duration of each phase over the stamped waves / samples, in shader-clock cycles."""
import os
import sys

os.environ["TBE_INTERACTION_ABLATION"] = "6"
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import _paths  # noqa: E402,F401
from fbgemm_gpu import _lib  # noqa: E402
from fbgemm_gpu._lib import check, ptr, stream_ptr  # noqa: E402

PHASES = ["wait for the copy", "fragment reads (ds_read_b128)", "issue the next copy", "MFMAs + pair staging", "stores issued",
          "(loop latch)"]


def main():
    B, F, D = 65536, 26, 128
    dev = torch.device("cuda", 0)
    lib = _lib.load()
    dense = torch.randn(B, D, device=dev)
    sparse = torch.randn(B, F, D, device=dev)
    out = torch.empty(B, D + (F + 1) * F // 2, device=dev)
    stamps = torch.zeros(8 * 4 * 32 * 6, dtype=torch.int64, device=dev)
    for it in range(3):
        if it == 2:
            lib.tbe_debug_set_interaction_stamps(stamps.data_ptr())
        check(lib.tbe_dlrm_interaction_forward_f32(ptr(dense), ptr(sparse), B, F, D, ptr(out), out.shape[1], stream_ptr(dev)),
              "tbe_dlrm_interaction_forward_f32")
        torch.cuda.synchronize()
    lib.tbe_debug_set_interaction_stamps(None)
    st = stamps.cpu().numpy().reshape(32, 32, 6).astype(np.float64)  # [wave, sample, stamp]
    st = st[:, 4:30]  # steady state
    per_sample = st[:, 1:, 0] - st[:, :-1, 0]
    print(f"cycles per sample per wave: median {np.median(per_sample):.0f} (p10 {np.percentile(per_sample, 10):.0f}, "
          f"p90 {np.percentile(per_sample, 90):.0f})")
    for k in range(5):
        d = st[:, :, k + 1] - st[:, :, k]
        print(f"  {PHASES[k]:32s} median {np.median(d):7.0f}  p10 {np.percentile(d, 10):7.0f}  p90 {np.percentile(d, 90):7.0f}")
    # wave 0's absolute timeline for a few samples
    w = st[0] - st[0, 0, 0]
    for i in range(4):
        print("  wave 0 sample", i, " ".join(f"{x:8.0f}" for x in w[i]))


if __name__ == "__main__":
    main()
