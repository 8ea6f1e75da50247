"""Pre-tuned library-GEMM selections (PyTorch TunableOp results for gfx950, made by
tools/tune_gemms.py).  Loading them only changes WHICH hipBLASLt / rocBLAS solution a GEMM shape
runs; tuning itself is off at run time, and shapes not in the file use the library default."""
from __future__ import annotations

import os

_DIR = os.path.dirname(os.path.abspath(__file__))
RESULTS = os.path.join(_DIR, "tunableop_gfx950.csv")
_done = False


def enable_tuned_gemms() -> bool:
    """Idempotent.  Returns True when the committed results were handed to TunableOp."""
    global _done
    if _done:
        return True
    if os.environ.get("VLLM_OMNI_AMD_TUNED_GEMMS", "1") != "1" or not os.path.exists(RESULTS):
        return False
    try:
        import torch.cuda.tunable as tn

        if os.environ.get("PYTORCH_TUNABLEOP_TUNING") == "1":
            return False  # somebody is tuning on purpose: do not interfere
        tn.enable(True)
        tn.tuning_enable(False)
        tn.record_untuned_enable(False) if hasattr(tn, "record_untuned_enable") else None
        ok = tn.read_file(RESULTS)
        _done = bool(ok) or ok is None
        return _done
    except Exception:  # noqa: BLE001 - an older torch or a validator mismatch: library defaults
        return False
