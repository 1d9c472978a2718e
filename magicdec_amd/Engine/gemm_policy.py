"""Which GEMM runs a linear of a step: md_linear (hand-written weight-streaming skinny GEMM, csrc/gemm.hip) or the
library GEMM (hipBLASLt through F.linear, with the TunableOp table of magicdec_amd/tuned/).

The rules below are the outcome of the same-box, graph-captured A/B of profiles/r02_gemm_ab.txt (tools/gemm_bench.py:
every decode / verify shape of the BASELINE models timed on both, weights cycled to defeat the Infinity Cache):

  * md_linear over the STREAMING weight layout (ops.PackedWeight) wins where the product is a long weight stream --
    M <= 64: w1|w3 (+ fused SiLU*mul), w2 and the lm head of the 8B model, w1|w3 and the lm head of the 1B model
    (4.5-6.0 TB/s against 3.0-5.1); M <= 128: w1|w3 and w2; M = 256: w2 only (K = 14336);
  * the library wins the short streams (qkv / wo of both models, every TP8 shard: the whole call is 10-20 us and the
    split-K combine of md_linear costs a second launch) and everything at M = 256 with K <= 4096 (there the product is
    MFMA / LDS bound, not HBM bound: a 256-row activation slab has to be re-read per 128 output columns).

A weight that md_linear will serve is re-packed once at setup_caches (Transformer._pack_weights); the row-major copy
stays for the prefill-sized products.  MAGICDEC_GEMM=hip forces md_linear wherever it supports the shape (packing
everything), MAGICDEC_GEMM=lib forces the library (the A/B switch)."""
import os

_MODE = os.environ.get("MAGICDEC_GEMM", "auto")
MIN_STREAM_BYTES = 60e6        # bf16 weight bytes of one call above which the streaming kernel wins at M <= 64
MIN_STREAM_BYTES_M128 = 100e6


def set_mode(mode: str):
    global _MODE
    assert mode in ("auto", "hip", "lib")
    _MODE = mode


def mode() -> str:
    return _MODE


def want_packed(N: int, K: int) -> bool:
    """Re-pack this weight into the streaming layout at load time?"""
    if _MODE == "lib" or K % 128:
        return False
    return _MODE == "hip" or N * K * 2 >= MIN_STREAM_BYTES


def use_skinny(M: int, N: int, K: int, swiglu: bool, int8: bool, packed: bool) -> bool:
    if _MODE == "lib" or M > 256:
        return False
    if _MODE == "hip" or int8:          # int8 rows are only streamed by md_linear (the library path dequantises)
        return True
    if not packed:
        return False
    nbytes = N * K * 2
    if M <= 64:
        return nbytes >= MIN_STREAM_BYTES
    if M <= 128:
        return nbytes >= MIN_STREAM_BYTES_M128
    return K >= 8192 and nbytes >= MIN_STREAM_BYTES_M128
