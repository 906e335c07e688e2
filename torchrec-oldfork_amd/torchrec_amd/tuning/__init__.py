"""Library-GEMM solution selection for the DLRM dense layers on MI355X.

The dense MLPs run on hipBLASLt / rocBLAS fp32 GEMMs (torch is the plumbing).  Their default
heuristic picks kernels that average 133 TFLOP/s over the step's 23 GEMM shapes; PyTorch's TunableOp
can instead replay a per-shape choice measured once on the target.  `gemm_gfx950_dlrm.csv` holds that
choice for the Criteo-1TB DLRM layer shapes at per-rank batches 65 536 / 32 768 / 16 384 / 8 192 / 4 096
(tools: `PYTORCH_TUNABLEOP_TUNING=1 python bench.py --global-batch B`, merged by hand): 150 TFLOP/s
(0.96 of the 157 TFLOP/s fp32 MFMA peak) at batch 65 536.  Same arithmetic type (fp32 MFMA), only the
kernel / tile choice changes.  Nothing is tuned at run time: tuning stays disabled, shapes that are
not in the file use the library default, and a file recorded for another library build is ignored.
"""
import os
import warnings

_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_gfx950_dlrm.csv")


def enable_tuned_gemms(path: str = _FILE) -> bool:
    """Turns on replay of the recorded GEMM choices for this process.  Returns False (and leaves the
    library defaults in place) when the file does not match this ROCm / hipBLASLt / GPU."""
    import torch
    import torch.cuda.tunable as tunable

    if not torch.cuda.is_available() or not os.path.exists(path):
        return False
    tunable.enable(True)
    tunable.tuning_enable(False)
    tunable.set_filename(path, insert_device_ordinal=False)
    try:
        ok = bool(tunable.read_file(path))
    except Exception as e:  # validators of another build
        warnings.warn(f"tuned GEMM file not usable here: {e}")
        ok = False
    if not ok:
        tunable.enable(False)
    return ok
