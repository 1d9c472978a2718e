"""Which GEMM runs a linear of a step: md_linear (hand-written weight-streaming skinny GEMM, csrc/gemm.hip) or the
library GEMM (hipBLASLt through F.linear, with the TunableOp table of magicdec_amd/tuned/).

The table below is the outcome of the same-box A/B in profiles/r02_gemm_ab.txt (tools/gemm_bench.py: every decode /
verify shape of the BASELINE models timed on both, weights cycled to defeat the Infinity Cache).  MAGICDEC_GEMM=hip
forces md_linear wherever it supports the shape, MAGICDEC_GEMM=lib forces the library (the A/B switch of bench.py)."""
import os

_MODE = os.environ.get("MAGICDEC_GEMM", "auto")
MAX_M_AUTO = 256          # md_linear covers M <= 256; above that the product is compute-bound library territory


def set_mode(mode: str):
    global _MODE
    assert mode in ("auto", "hip", "lib")
    _MODE = mode


def use_skinny(M: int, N: int, K: int, swiglu: bool, int8: bool) -> bool:
    if _MODE == "lib":
        return False
    if M > 256:
        return False
    if _MODE == "hip" or int8:          # int8 rows are only streamed by md_linear (the library path dequantises)
        return True
    return M <= MAX_M_AUTO
