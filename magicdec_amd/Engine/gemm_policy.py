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

  * round 3: the LAUNCH-BOUND small products (every linear of the 1B draft model, every tensor-parallel shard, the
    narrow projections of an 8B step) go to md_linear_fused (csrc/tilegemm.hip): one launch for the product AND the op
    behind it (rope+append, residual add, SiLU*mul), no split-K combine, no partial sums in HBM.

  * round 4: the 129..256-ROW products of a verify step whose column count fills the chip without splitting K, and the
    K = 14336 down projection, go to md_linear_block (csrc/blockgemm.hip: 256 x 128 block tiles, LDS-DMA rings, loader +
    MFMA waves) -- profiles/r04_block_ab_final.txt, same process, graph-captured, weights cycled (us; "lib" includes the
    elementwise kernel behind it, "skinny" its combine launch):
        8B w1|w3 + SiLU*mul  M = 256: lib 85-89 (68.8 + 6.2 inside the cfg3 trace), skinny 108-119, block 67-72
        8B w2 + add + norm   M = 256: lib 66-68, skinny 53.5-55, block 48.6-49.9
        8B lm head           M = 256: lib 293-297, skinny 345-350, block 274-278
        8B w1|w3             M = 128: lib 58-65, skinny 61.6-63.8, block 52.3-57
    and NOT the narrow projections, where splitting K over ~256 workgroups costs a partial-sum round trip the product is
    too small to pay for (wqkv 30-32 vs lib 27 + 6.3 for rope/append: a wash; wo 34-35 vs 25; every TP8 shard slower).

A weight a hand-written kernel will serve is re-packed once at setup_caches (Transformer._pack_weights); the row-major
copy stays for the prefill-sized products.  MAGICDEC_GEMM=hip forces md_linear wherever it supports the shape,
MAGICDEC_GEMM=lib forces the library, MAGICDEC_FUSED=0/1 switches md_linear_fused off / on everywhere (the A/B
switches)."""
import os

_MODE = os.environ.get("MAGICDEC_GEMM", "auto")
_FUSED = os.environ.get("MAGICDEC_FUSED", "auto")     # "0": never md_linear_fused, "1": wherever it supports the shape
MIN_STREAM_BYTES = 60e6        # bf16 weight bytes of one call above which the streaming kernel wins at M <= 64
MIN_STREAM_BYTES_M128 = 100e6
# md_linear_fused reads a weight tile once per 32-row M tile (through L2) and a 32-row slab of the activations once per
# weight tile: it is the launch-bound regime's kernel.  Rules from the A/B of tools/fused_bench.py
# (profiles/r03_fused_ab.txt; "unfused" there = hipBLASLt + the small kernel behind it, graph-captured, weights cycled):
#   * M <= 128 (draft steps, autoregressive target steps): fused while m_tiles x weight bytes <= 70 MB and K <= 4096 --
#     the 1B model's wqkv 8.0 vs 15.6 us, wo 10.3 vs 12.2, every TP4 shard of it (6.1-9.8 vs 10.0-15.0), every TP8
#     shard of the 8B model at M = 64 (8.9-12.9 vs 10.8-15.7).  Above that the x slab (M x K, one hot MB) is re-read by
#     every column tile through the same L2 channels and the kernel falls behind (1B w2, K = 8192: 23.6 vs 19.0;
#     1B w1|w3, 1024 tiles: 22.2 vs 20.4) -- those stay on md_linear / the library;
#   * M = 256 (verify): only the qkv projection with its rope+append epilogue (8B/8: 9.2 vs 14.0 us, 8B/4: 18.5 vs
#     19.8); wo / w1|w3 / w2 shards are served faster by the library (14.4 vs 14.1, 30.2 vs 22.9, 23.0 vs 16.6).
FUSED_MAX_L2_BYTES = 70e6
FUSED_MAX_L2_BYTES_T22 = 120e6
FUSED_DEEP_SWIGLU_T22_MAX_L2_BYTES = float(os.environ.get("MAGICDEC_DEEP_SWIGLU_L2", "150e6"))   # (env: A/B switch)
_SHARD70B = os.environ.get("MAGICDEC_SHARD70B", "1")     # "0": without the two round-6 rules for the 70B shards (A/B switch)
FUSED_MAX_K = 4096
FUSED_QKV_M256_MAX_L2_BYTES = 110e6
# round 5 (profiles/r05_fused_tp8_shard_tiles.txt, M = 256): two more TP8 shards of the 8B verify pass go to the fused kernel --
# w1|w3 (29.4 MB; 2 x 2 tiles: 21.1-22.3 us against 24.1 for the library + the SiLU*mul launch) and wo (4.2 MB, K = 512:
# 12.0 against 14.1 incl. the add + norm behind it); w2 (14.7 MB: 17.1 vs 16.9) and everything wider stay on the library
FUSED_M256_SWIGLU_MAX_BYTES = 32e6
FUSED_M256_NARROW_MAX_BYTES = 5e6
FUSED_M256_RESID_T22_MAX_BYTES = 34e6
_WO256 = os.environ.get("MAGICDEC_WO256", "1")        # "0": the 8B wo at 256 rows stays on the library (A/B switch)
# a linear that can ALSO absorb the RMSNorm in front of it (deferred norm: its input is the un-normalised h of a fused
# residual epilogue) saves that launch (~5 us + a boundary): the 1B w1|w3 at M = 64 (134 MB of tile traffic, 22.2 us
# against md_linear's 21.9) then wins on the fused kernel
FUSED_MAX_L2_BYTES_WITH_NORM = 140e6


# round 4: md_linear (the streaming kernel) can absorb the same deferred norm (md_linear_normed): a slab is normalised once
# per workgroup on its way to LDS.  Free while the kernel waits for weights anyway (<= 64 rows); at 128 / 256 rows the
# staging of a slab is as much vector work as its MFMAs and the launch it saves is cheaper (tools/fused_bench.py --pro 1:
# 1B w1|w3 at 128 rows 35.4 vs 30.1 us with the norm as its own launch, 8B at 256 rows 179 vs 105).
SKINNY_NORM_MAX_M = 64
_SKINNY_NORM = os.environ.get("MAGICDEC_SKINNY_NORM", "auto")     # "0": always materialise the norm in front of md_linear


def skinny_absorbs_norm(M: int) -> bool:
    return _SKINNY_NORM != "0" and M <= SKINNY_NORM_MAX_M


def set_mode(mode: str):
    global _MODE
    assert mode in ("auto", "hip", "lib")
    _MODE = mode


def set_fused(mode: str):
    global _FUSED
    assert mode in ("auto", "0", "1")
    _FUSED = mode


def mode() -> str:
    return _MODE


def fused_mode() -> str:
    return _FUSED


_PACKED = os.environ.get("MAGICDEC_PACKED_COPIES", "auto")   # "0": no streaming-layout copies at all (every linear on the
                                                             # library / the row-major md_linear): saves their HBM


def set_packed_copies(mode: str):
    global _PACKED
    assert mode in ("auto", "0")
    _PACKED = mode


def want_packed(N: int, K: int, swiglu: bool = False, int8: bool = False, rows=None) -> bool:
    """Re-pack this weight into the streaming layout at load time?  Only when a hand-written kernel may actually be
    chosen for it at some row count of a decode / verify step: `rows` = the counts the engine will run (the back-end knows
    them: batch x {1, 2, dec_len}), or None = any of 1..256 (round 3 packed every bf16 weight with
    K % 128 == 0; round 6: the 8B wqkv, served by the library at 64 and 256 rows, is no longer packed for the 32-row
    steps configs[2] never runs)."""
    if _MODE == "lib" or _PACKED == "0" or K % 64:
        return False
    if int8:
        return K % 128 == 0            # int8 rows are only streamed by md_linear, at every row count
    if _MODE == "hip" or _BLOCK == "1" or _FUSED == "1" or _SPLIT == "1":
        return True
    kinds = ("swiglu",) if swiglu else ("plain", "resid", "qkv")
    for M in ((1,) + tuple(range(32, 257, 32)) if rows is None else tuple(m for m in rows if m <= 256)):
        # (default: every 32-row tile count a rule can depend on -- ADVICE r4)
        if use_skinny(M, N, K, swiglu, False, True):
            return True
        for kind in kinds:
            if use_block(M, N, K, kind) or use_fused(M, N, K, kind, absorbs_norm=True) or use_split(M, N, K, kind):
                return True
    return False


def use_fused(M: int, N: int, K: int, kind: str = "plain", absorbs_norm: bool = False) -> bool:
    """kind: "qkv" (rope+append epilogue), "resid", "swiglu" or "plain"; absorbs_norm: the call would also apply a
    deferred RMSNorm to its input."""
    if _MODE == "lib" or _FUSED == "0" or M > 256 or K % 128 or N % 32:
        return False
    if _FUSED == "1":
        return True
    m_tiles = (M + 31) // 32
    # a product the kernel runs with 2 x 2 tiles (csrc/tilegemm.hip launch_tile_pro: two M tiles, an even number of column
    # tiles, >= 192 tile groups; with the deferred-norm prologue since round 6, K <= 4096) re-reads the weights once per
    # PAIR of M tiles: the 1B w1|w3 -- 18.4-18.9 us at 64 rows against md_linear's 22.8-23.2, and with the norm absorbed
    # 20.3 (profiles/r06_fused_pro22_ab.txt); at 128 rows (the two-token draft step) 32.0 with the norm against the library
    # + rmsnorm + SiLU*mul launches (20.5 + 5.1 + 5.4 in the trace), which also frees that weight's row-major copy
    t22 = (kind in ("swiglu", "plain", "resid") and m_tiles >= 2 and (N // 32) % 2 == 0
           and (N // 64) * ((m_tiles + 1) // 2) >= 192 and (not absorbs_norm or (kind == "swiglu" and K <= 4096)))
    l2_bytes = ((m_tiles + 1) // 2 if t22 else m_tiles) * N * K * 2
    if M <= 128:
        # round 6, the 70B model's TP-8 shards at the 128 rows of configs[3]'s verify (tools/shard_bench.py,
        # profiles/r06_shard70b_ab.txt): w2 (8192 x 3584, 2 x 2 tiles) 22.0 us against 24.4 for the library -- wide products
        # on 2 x 2 tiles may re-read up to 120 MB; wqkv (1280 x 8192) 16.7 against 21.1 + the rope/append launch -- a deep
        # NARROW qkv shard is fused up to K = 8192 once >= 128 tile workgroups exist (at 32 rows, 40 workgroups, the
        # library wins: 15.1 vs 18.2)
        wide = t22 and _SHARD70B != "0"
        lim = FUSED_MAX_L2_BYTES_WITH_NORM if absorbs_norm else (FUSED_MAX_L2_BYTES_T22 if wide else FUSED_MAX_L2_BYTES)
        if l2_bytes <= lim and K <= FUSED_MAX_K:
            return True
        if _SHARD70B == "0" or not (FUSED_MAX_K < K <= 2 * FUSED_MAX_K):
            return False
        # deep (4096 < K <= 8192) shards -- the 70B model's and Qwen2.5-32B's (dim 5120) at TP-8, tools/shard_bench.py:
        #   qkv: 70B 1280 x 8192 at 128 rows 16.7 us against 21.1 + rope/append; Qwen 896 x 5120 11.5-12.1 against 13.7 + 5
        #        (112 tile workgroups); with 40 workgroups (70B at 32 rows) the library wins, 15.1 vs 18.2;
        #   w1|w3 on 2 x 2 tiles: Qwen 6912 x 5120 at 128 rows 27.0-27.2 against 36.5 for library + SiLU*mul (34.5 md_linear)
        if kind == "qkv":
            return l2_bytes <= FUSED_QKV_M256_MAX_L2_BYTES and (N // 32) * m_tiles >= 100
        return kind == "swiglu" and t22 and not absorbs_norm and l2_bytes <= FUSED_DEEP_SWIGLU_T22_MAX_L2_BYTES
    if kind == "swiglu":
        return N * K * 2 <= FUSED_M256_SWIGLU_MAX_BYTES and K <= FUSED_MAX_K
    if kind in ("plain", "resid"):
        if N * K * 2 <= FUSED_M256_NARROW_MAX_BYTES and K <= FUSED_MAX_K:         # measured at K = 512 only
            return True
        # round 6: the 8B wo at 256 rows on 2 x 2 tiles (4 x 64 tile groups = 256 workgroups) WITH its residual add is a
        # measured tie with the library + add/norm launch (25.2-25.9 vs 24.9-25.0 us, profiles/r06_fused_pro22_ab.txt;
        # in-trace A/B profiles/r06_ab_wo256_*): taking it leaves NO weight of configs[2] held in two layouts during
        # decode (the library read the row-major tensor) and one library kernel fewer in the verify pass
        return (_WO256 != "0" and kind == "resid" and t22 and K <= FUSED_MAX_K
                and N * K * 2 <= FUSED_M256_RESID_T22_MAX_BYTES)
    return kind == "qkv" and l2_bytes <= FUSED_QKV_M256_MAX_L2_BYTES and K <= FUSED_MAX_K


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


# round 6: md_linear_fused_split (csrc/tilegemm.hip, FL_PARTIAL) -- the tile kernel with K ALSO split over workgroups, for the
# deep narrow output projections of a <= 128-row step whose combine launch is the residual add + RMSNorm launch anyway: the 1B
# model's w2 (N = 2048, K = 8192, 33.5 MB), which every single-launch decomposition leaves per-CU-ingest-bound (fused tile
# kernel 22-24 us, library 15.7 + 5 for the add + norm, md_linear 17 + 5).  Rule from tools/split_bench.py
# (profiles/r06_split_ab.txt).
SPLIT_MIN_K = 8192
SPLIT_MAX_M = 128
_SPLIT = os.environ.get("MAGICDEC_SPLIT", "auto")      # "0": never, "1": every output projection it supports


def set_split(mode: str):
    global _SPLIT
    assert mode in ("auto", "0", "1")
    _SPLIT = mode


def use_split(M: int, N: int, K: int, kind: str) -> bool:
    """md_linear_fused_split_add_rmsnorm for this output projection (kind "resid")?"""
    if _MODE == "lib" or _SPLIT == "0" or kind != "resid" or M > 256 or K % 128 or N % 32 or N > 8192:
        return False
    if _SPLIT == "1":
        return True
    return M <= SPLIT_MAX_M and K >= SPLIT_MIN_K and N * K * 2 < MIN_STREAM_BYTES


BLOCK_MIN_TILES = 160          # 128-column tiles that fill the chip without a K split (w1|w3: 224, lm head: 1002)
_BLOCK = os.environ.get("MAGICDEC_BLOCK", "auto")      # "0": never md_linear_block, "1": wherever it supports the shape


def set_block(mode: str):
    global _BLOCK
    assert mode in ("auto", "0", "1")
    _BLOCK = mode


def block_mode() -> str:
    return _BLOCK


def use_block(M: int, N: int, K: int, kind: str) -> bool:
    """md_linear_block for this linear?  kind: "swiglu", "resid" (output projection + residual add + RMSNorm: the split-K
    combine launch does both) or "plain"."""
    if _MODE == "lib" or _BLOCK == "0" or M > 256 or N % 128 or K % 64:
        return False
    if _BLOCK == "1":
        return True
    tiles = N // 128
    if M > 128:
        if kind == "resid":
            return K >= 8192 and N <= 8192 and N * K * 2 >= MIN_STREAM_BYTES_M128   # the 8B w2 (K = 14336); wo and
                                                                                    # the TP shards stay on the library
        return tiles >= BLOCK_MIN_TILES and kind in ("swiglu", "plain")
    return M > 64 and kind == "swiglu" and tiles >= BLOCK_MIN_TILES      # cfg2's 128-row verify: w1|w3 only


def choose(M: int, N: int, K: int, swiglu: bool, int8: bool, packed: bool, kind: str = None,
           absorbs_norm: bool = False) -> str:
    """"fused" (md_linear_fused), "block" (md_linear_block), "skinny" (md_linear) or "lib" (hipBLASLt) for one linear
    of a step."""
    kind = kind or ("swiglu" if swiglu else "plain")
    if packed and not int8 and _BLOCK == "1" and use_block(M, N, K, kind):
        return "block"
    if packed and not int8 and use_fused(M, N, K, kind, absorbs_norm):
        return "fused"
    if packed and not int8 and use_block(M, N, K, kind):
        return "block"
    return "skinny" if use_skinny(M, N, K, swiglu, int8, packed) else "lib"
