"""Library GEMM selection for the fixed shapes of the path (plumbing: which hipBLASLt / rocBLAS kernel a torch Linear
runs; the arithmetic stays an fp32 GEMM).

hipBLASLt's default heuristic picks small macro-tiles for the transformer's M = 6304 rows (64x64 for the 512 -> 4096
feed-forward, 32x64 for the 512 -> 512 output projection: 99 and 54 TFLOP/s); benchmarking its candidates once per
shape (PyTorch's TunableOp, tools/tune_gemms.py on an MI355X) finds 120 and 93 TFLOP/s.  The winners are shipped in
gemm_tuning_gfx950.csv and only LOOKED UP here: tuning itself is off at run time (the point refiner's GEMMs have
data-dependent row counts; tuning every new size would stall the stream), shapes that are not in the file use the
library default, and the file is ignored as a whole when its validators (torch / ROCm / hipBLASLt versions, gfx
architecture) do not match the running stack.  AMAV_TUNED_GEMMS=0 turns the lookup off, AMAV_TUNED_GEMMS=tune is what
the tuning tool sets.
"""
import os

import torch

TUNING_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_tuning_gfx950.csv")
_loaded = {"done": False, "active": False}


def use_tuned_gemms(path=None) -> bool:
    """Idempotent.  Returns True when tuned solutions are being looked up."""
    mode = os.environ.get("AMAV_TUNED_GEMMS", "1")
    if mode == "tune":
        return True  # tools/tune_gemms.py drives TunableOp itself
    if _loaded["done"]:
        return _loaded["active"]
    _loaded["done"] = True
    path = path or TUNING_FILE
    if mode == "0" or not torch.cuda.is_available() or not os.path.exists(path):
        return False
    import torch.cuda.tunable as tunable

    if tunable.is_enabled():  # the user configured TunableOp through PYTORCH_TUNABLEOP_*: leave it alone
        return False
    tunable.enable(True)
    tunable.tuning_enable(False)
    if not tunable.read_file(path):
        tunable.enable(False)
        return False
    _loaded["active"] = True
    return True


# ---- fp16 x 2 split projections: hipBLASLt kernel per shape (csrc/gemm.hip) ------------------------------------------
SPLIT_TUNING_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_split_tuning_gfx950.csv")
_split = {"table": None}


def split_gemm_index(rows: int, n: int, k3: int) -> int:
    """hipBLASLt algorithm index for out[rows, n] = a[rows, k3] x w[n, k3]^T (fp16 in, fp32 out) from the shipped table
    (tools/tune_split_gemms.py), or -1 = the library's own heuristic: unknown shape, AMAV_TUNED_GEMMS=0, or a table that
    was made with another hipBLASLt build / architecture (its first line names both; indices are only meaningful there)."""
    if _split["table"] is None:
        table = {}
        if os.environ.get("AMAV_TUNED_GEMMS", "1") != "0" and os.path.exists(SPLIT_TUNING_FILE) and torch.cuda.is_available():
            from . import ops

            with open(SPLIT_TUNING_FILE) as fh:
                lines = [ln.strip() for ln in fh if ln.strip()]
            arch = torch.cuda.get_device_properties(0).gcnArchName.split(":")[0]
            if lines and lines[0].lstrip("# ").split() == [ops.gemm_library_version(), arch]:
                for ln in lines[1:]:
                    r, nn, kk, idx = (int(x) for x in ln.split(",")[:4])
                    table[(r, nn, kk)] = idx
        _split["table"] = table
    return _split["table"].get((int(rows), int(n), int(k3)), -1)
