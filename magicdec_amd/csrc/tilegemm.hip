// Fused small-problem GEMM for the launch-bound linears of a decode / verify step (gfx950):
//     out = epilogue( x[M][K] . W[N][K]^T + bias )            M <= 256, weights in the streaming layout of md_linear
//
// replaces: nn.Linear of Attention / FeedForward in a decode step TOGETHER with the op that consumes its output --
//   wqkv   + RoPE + paged KV append        (Engine/SnapKV/model.py:322-336: wqkv, apply_rope, update_kv)
//   wo     + residual add                  (model.py:260-278: h = x + attention(...))
//   w1|w3  + SiLU * mul                    (model.py:451-455)
//   w2     + residual add                  (model.py:260-278: out = h + feed_forward(...))
//
// Why a second GEMM kernel.  md_linear (gemm.hip) is a weight-STREAMING kernel: split-K over ~256 workgroups + a combine
// launch; right for 60..500 MB of weights.  The 1B draft model's linears, every tensor-parallel shard, and the qkv / wo
// of an 8B step are 2..30 MB: there the whole call is latency (hipBLASLt: 7-12 us for 3 MB of weights,
// profiles/r02_gemm_ab_short_stream_rejected.txt) and each is followed by a 5 us elementwise launch (rope+append,
// add, SiLU*mul) -- a 1B draft step is ~150 launches for 0.4 ms of HBM time (profiles/r03_emulated_tp8_iter_breakdown_before.csv:
// 40 % of a TP8 rank's iteration is such GEMMs, another 20 % the small kernels behind them).  This kernel removes the
// second launch of every pair and runs the product itself with nothing between "all loads issued" and "all data there":
//
//   * a workgroup (8 wavefronts) owns one 32-row x 32-column output tile and the WHOLE K range: no split-K across
//     workgroups, no partial sums in HBM, no combine launch; grid = (M/32) x (N/32) tiles, block id -> tile mapping
//     keeps the M-tiles of one weight tile on one XCD (they share its L2 lines);
//   * the K range is split over the workgroup's 8 wavefronts (16 when the grid is too small to give every CU two
//     workgroups); every wavefront is an independent pipeline with NO barrier in its loop: W fragments stream global ->
//     registers (one contiguous KiB per instruction, 8-deep rolling ring; non-temporal when there is one M tile, kept in
//     L2 for the sibling M tiles otherwise), its private slice of the activations goes global -> registers -> wave-private LDS image ->
//     MFMA A fragments (full 256-B row segments per 16 lanes from L2 instead of fragment-shaped 32-B pieces -- the
//     latter is what made round 2's "short-stream" attempt 1.4-3x slower) with the next 128-deep chunk's loads in flight
//     under the current chunk's MFMAs;
//   * v_mfma_f32_32x32x16_bf16, fp32 accumulators; the 8 partial tiles are summed through LDS in wave order
//     (deterministic: eager, graph replay and every TP rank see the same bits);
//   * epilogue on the finished tile, one thread per adjacent column pair, with the reference's rounding points:
//       NONE         out = bf16(acc + bias)
//       RESID        out = bf16(resid + bf16(acc + bias))                                   (bf16 add)
//       SWIGLU       out = bf16(bf16(silu(bf16(h1))) * bf16(h3))   (tile = 16 rows of w1 + the same 16 of w3)
//       ROPE_APPEND  qkv = bf16(acc + bias); q columns: interleaved RoPE (fp32 table, un-fused mul/add: bit-identical
//                    to md_rope_append) -> q_out; k columns: RoPE -> paged cache(s); v columns -> paged cache(s)
//                    (bf16 or fp8 e4m3 pages, NHD or HND, optional second bf16 cache; page table read on the device).
//
// Deferred RMSNorm (the norm between two fused linears without a launch of its own).  The reference's block is
// h = x + wo(...); y = rmsnorm(h) * w; ... = w13(y).  A tile kernel cannot normalise what it produces (a row's sum of
// squares spans all column tiles), so the work is split: the RESID epilogue also writes, per row, the sum of squares of
// ITS 32 columns of h (`ssq_out[M][N/32]`, a fixed-order 16-lane reduction), and the consuming linear (PRO = true) takes
// the un-normalised h as x: it adds a row's N/32 partials in a fixed order, forms rstd = rsqrt(sum / K + eps) exactly as
// md_rmsnorm does, and applies y = bf16(bf16(h * rstd) * w) -- the reference's rounding points
// (Engine/SnapKV/model.py:464-469) -- to its activation slice on the way from registers to the LDS image.  Only the
// order of the fp32 sum of squares differs from the stand-alone kernel's.
//
// Where it wins and where it does not (profiles/r03_fused_ab.txt, Engine/gemm_policy.py): per-workgroup time is
// ~2.5 us + 128 B x K / (45-50 GB/s per CU): the 32-row activation slab (M x K, one hot MB) is re-read by every column
// tile through the same L2 channels.  K <= 4096 with few M tiles: 6-13 us against 10-16 for the library + the small
// kernel behind it; K = 8192 or M = 256 with many column tiles: slower than the library -- those shapes stay there.
#include "md_common.h"

unsigned int* md_page_overflow_counter_device();   // kvops.hip: rows dropped beyond a request's mapped pages

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kKC = 128;                 // k depth of one activation chunk
constexpr int kPitch = kKC * 2 + 16;     // LDS row pitch (272 B: consecutive rows start 4 banks apart)
constexpr int kWaveLds = 32 * kPitch;    // wave-private activation image (8704 B >= the 4 KiB partial tile)

enum { FL_NONE = 0, FL_SWIGLU = 1, FL_RESID = 2, FL_ROPE_APPEND = 3,
       FL_PARTIAL = 4 };      // internal: the K range is ALSO split over workgroups; fp32 partial tiles -> workspace

struct KvTable {
    void* cache;
    const int32_t* indices;
    const int32_t* indptr;
    const int32_t* last;
};

struct TileParams {
    const bf16_t* x;
    const bf16_t* w;          // streaming layout [N/32][K/16][64][8]
    const bf16_t* bias;
    bf16_t* out;
    const bf16_t* resid;
    int64_t ldx, ldo, ldr;
    int M, N, K, n_tiles, m_tiles;
    // ROPE_APPEND
    int H, KH, D, rows_per_req, max_pos, page_size, hnd;
    const int32_t* offsets;
    const float* cos_sin;
    KvTable t1, t2;
    const float* k_scale;
    const float* v_scale;
    unsigned int* overflow;   // dropped-row counter (md_page_overflow_count)
    // deferred RMSNorm (see the kernel header): producer side (RESID) / consumer side (PRO)
    float* ssq_out;           // RESID: [M][n_tiles] sum of squares of the tile's 32 output columns per row, or null
    const float* pro_ssq;     // PRO: [M][pro_tiles] partial sums of squares of the rows of x (x = the un-normalised h)
    const bf16_t* pro_w;      // PRO: RMSNorm weight [K]
    float pro_eps;
    int pro_tiles;
    // FL_PARTIAL (md_linear_fused_split): S K-slices over workgroups, fp32 partial planes [S][M][N]
    float* partial;
    int S;
    unsigned long long* dbg;  // dev builds: per-wave phase timestamps (md_debug_set_tile_timing), or null
};

#ifdef MD_TILE_TIMING
// `make TIMING=1` only (libmagicdec_hip_timing.so, tools/tile_timing.py): phase timestamps of a wavefront (100 MHz wall
// clock: comparable across CUs), kept in registers and written once at the end.  The product build has none of it.
#define MD_TS(i) do { if (p.dbg) ts[i] = wall_clock64(); } while (0)
#else
#define MD_TS(i) do { } while (0)
#endif

__device__ __forceinline__ float silu_bf16(float h1) {
    return bf16_to_f32(f32_to_bf16(h1 / (1.0f + expf(-h1))));     // same expression as md_silu_mul / md_linear
}

__device__ __forceinline__ unsigned int pack2(float a, float b) {
    const bf16x2 pk = {f32_to_bf16(a), f32_to_bf16(b)};
    return *reinterpret_cast<const unsigned int*>(&pk);
}

__device__ __forceinline__ f32x2 unpack2(unsigned int v) {          // the two bf16 of a dword as fp32 (low half first)
    return f32x2{__uint_as_float(v << 16), __uint_as_float(v & 0xffff0000u)};
}

// lane l of every 16-lane group receives the value lane `i` of ITS group holds (v_mov_b32_dpp row_newbcast:i; `i` is a
// constant after unrolling -- the builtin wants a literal)
__device__ __forceinline__ float row_bcast(float v, int i) {
    const int x = (int)__float_as_uint(v);
#define MD_BC(I) case I: return __uint_as_float((unsigned int)__builtin_amdgcn_update_dpp(0, x, 0x150 + I, 0xf, 0xf, true));
    switch (i & 15) {
        MD_BC(0) MD_BC(1) MD_BC(2) MD_BC(3) MD_BC(4) MD_BC(5) MD_BC(6) MD_BC(7)
        MD_BC(8) MD_BC(9) MD_BC(10) MD_BC(11) MD_BC(12) MD_BC(13) MD_BC(14) MD_BC(15)
    }
#undef MD_BC
    return v;
}

template <bool NT>
__device__ __forceinline__ u32x4 ld_w(const bf16_t* p) {
    if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return *reinterpret_cast<const u32x4*>(p);
}

// store an adjacent pair (two consecutive cache elements) of a bf16 / fp8 page
template <bool FP8>
__device__ __forceinline__ void store_pair(void* cache, int64_t off, float a, float b, float inv_scale) {
    if constexpr (FP8) {
        const float qa = fminf(fmaxf(__fmul_rn(a, inv_scale), -448.f), 448.f);
        const float qb = fminf(fmaxf(__fmul_rn(b, inv_scale), -448.f), 448.f);
        const unsigned int r = __builtin_amdgcn_cvt_pk_fp8_f32(qa, qb, 0u, false);
        *reinterpret_cast<unsigned short*>(reinterpret_cast<unsigned char*>(cache) + off) = (unsigned short)(r & 0xffffu);
    } else {
        *reinterpret_cast<unsigned int*>(reinterpret_cast<bf16_t*>(cache) + off) = pack2(a, b);
    }
}

// element offset of (row `pos` of the request, kv head h, element d) in the K half of a paged cache; -1: not stored
__device__ __forceinline__ int64_t kv_elem_offset(const KvTable& t, int b, int rows_per_req, int jrow, int page_size,
                                                  int KH, int D, int h, int d, bool hnd, bool* overflow) {
    const int p0 = t.indptr[b];
    const int np = t.indptr[b + 1] - p0;
    const int len = np > 0 ? (np - 1) * page_size + t.last[b] : 0;
    const int pos = len - rows_per_req + jrow;
    *overflow = pos >= np * page_size;
    if (pos < 0 || *overflow) return -1;
    const int page = pos / page_size;
    const int slot = pos - page * page_size;
    const int64_t base = (int64_t)t.indices[p0 + page] * 2 * page_size * KH * D;
    return hnd ? base + ((int64_t)h * page_size + slot) * D + d : base + ((int64_t)slot * KH + h) * D + d;
}

// NW = wavefronts per workgroup = K slices (8, or 16 when the grid cannot give every CU two workgroups: the bytes a CU
// has in flight -- NW x (8 KiB of W + 8 KiB of x) -- are what its ingest rate is made of, ~50 GB/s per CU at NW = 8);
// WNT = stream W with non-temporal loads (one M tile: every weight byte is read once) or keep it in L2 for the sibling
// M tiles of the same weight tile.
// MT x NT = 32-row x 32-column MFMA tiles per workgroup (round 4).  What a CU has to ingest is W x (M tiles that
// re-read it) + the x slab x (column tiles that re-read it), and a CU ingests ~50 GB/s whatever the kernel
// (DESIGN.md 3.3): with 1 x 1 tiles the 1B w1|w3 at M = 64 (67 MB of W) moves 134 MB of W (two M tiles) + 134 MB of x
// (512 column tiles x 256 KB) = 1.05 MB per CU = the measured 24.5 us; a 2 x 2 tile (64 rows x 64 columns, one
// workgroup per CU) halves both.  Each K-slice wave then owns 2 x 2 accumulators, two W fragment streams and a 64-row
// activation image; the summation order over the K slices (wave order) and every epilogue are unchanged, so the bits
// are those of the 1 x 1 form.
template <int EPI, bool FP8, int NW, bool WNT, bool PRO, int MT, int NT>
__global__ __launch_bounds__(64 * NW, MT * NT > 1 ? 2 : 4) void tile_gemm_kernel(const TileParams p) {
    constexpr int kNW = NW;
    // Round 6, the deferred-norm prologue on 2 x 2 tiles (the 1B w1|w3 at 64 rows).  Phase timestamps of the 1 x 1 form
    // (tools/tile_timing.py, profiles/r06_tile_phase_timing.txt) show where its 6 us go: ~1 us of prologue in front of the
    // first load, +2.4 us per activation chunk between "landed" and "staged" (four wavefronts per SIMD normalising at
    // once, 11 vector instructions per dword), and every column tile normalising the same rows (512 x).  The 2 x 2 form
    // halves the last; it used to sit at 256 registers and spill.  A CU keeps ~64 KB of loads in flight whoever issues
    // them, so 8 wavefronts need 8 KB each, not 32: the W ring of this form is FOUR deep (kRD), which
    // frees 32 registers; the row scales' partial sums are requested FIRST, the W ring and the first activation chunk
    // behind them, and the workgroup meets at a bare s_barrier (no vmcnt(0) drain); the norm weights of a chunk are
    // unpacked once, both roundings are one v_cvt_pk_bf16_f32.
    constexpr bool kNewPro = PRO && MT * NT > 1;
    constexpr int kRD = kNewPro ? 4 : 8;                  // W fragments in flight per column tile and wavefront
    constexpr int kWaveLdsT = MT * kWaveLds;               // wave-private activation image of 32 x MT rows
    static_assert(MT * NT * 4096 <= kWaveLdsT, "the partial tiles of a wave must fit its activation image");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // NW x kWaveLdsT (+ 32 MT floats rstd when PRO)
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform: scalar base addresses
#ifdef MD_TILE_TIMING
    unsigned long long ts[6] = {0, 0, 0, 0, 0, 0};
#endif
    MD_TS(0);
    // block id -> (weight tile group, M tile group): block b runs on XCD b % 8; the M groups of one weight tile group
    // stay on one XCD
    const int m_groups = (p.m_tiles + MT - 1) / MT, n_groups = p.n_tiles / NT;
    int tng, tmg, slice = 0;
    if constexpr (EPI == FL_PARTIAL) {
        // K slice in the FAST index: with S = 8 a slice -- i.e. the Kt columns of x it reads -- lives on one XCD's L2
        // (block b runs on XCD b % 8), and the partial planes of a slice are written by one XCD
        slice = blockIdx.x % p.S;
        const int rest = blockIdx.x / p.S;
        tmg = rest % m_groups;
        tng = rest / m_groups;
    } else {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        tng = (slot / m_groups) * 8 + xcd;
        tmg = slot % m_groups;
    }
    if (tng >= n_groups) return;
    const int m0 = tmg * 32 * MT, tn0 = tng * NT;

    // 16-deep MFMA k-steps of this wavefront's slice (FL_PARTIAL: of this workgroup's K slice, then of the wavefront)
    const int ksteps_w = (p.K >> 4) / (kNW * (EPI == FL_PARTIAL ? p.S : 1));
    const int ks0 = (slice * kNW + wave) * ksteps_w;
    const bf16_t* wbase[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
        wbase[nt] = p.w + ((int64_t)(tn0 + nt) * (p.K >> 4) + ks0) * 512;    // wave-uniform; + lane * 8 per lane
    unsigned char* my_lds = lds + wave * kWaveLdsT;

    // activation staging: instruction i covers rows 4i + (lane >> 4), 16 lanes read one 256-B row segment
    const int ar = lane >> 4, c16 = lane & 15;
    const unsigned char* xbase = reinterpret_cast<const unsigned char*>(p.x + ks0 * 16);   // wave-uniform
    unsigned int xrow[8];                            // byte offsets of the lane's rows (M * ldx * 2 < 2^32); MT == 1
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = m0 + 4 * i + ar;
        xrow[i] = (unsigned int)(r < p.M ? r : p.M - 1) * (unsigned int)p.ldx * 2u;   // rows >= M re-read row M-1
    }
    const int nchunk = (ksteps_w + 7) >> 3;
    u32x4 xa[8 * MT];
    u32x4 nwv = {0u, 0u, 0u, 0u};                   // PRO: the norm weights of this lane's 8 columns of the chunk
    float rs8[8];                                   // PRO, MT == 1: rstd of this lane's staging rows (MT > 1: re-read
                                                    // from LDS per chunk -- 16 more registers would spill)
    // (the row scales are formed BEFORE the first loads are issued: with the loads in flight first, the barrier below
    // drains them -- 1B w1|w3 at 64 rows 29.2 us against 25.8, profiles/r04_fused_pro_prologue_ab.txt.  What the
    // deferred norm costs is the VALU work of normalising the same rows in every column tile: 18.9 -> 25.8 us there.)
    float my_rs = 0.f;                              // kNewPro: the row scale this lane keeps for its 16-lane group
    float pv[MT][8];                                // kNewPro: this thread's partial sums of squares (in flight)
    if constexpr (kNewPro) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            const int r = (tid >> 4) + 32 * mi, part = tid & 15;
            const int gr = m0 + r < p.M ? m0 + r : p.M - 1;
#pragma unroll
            for (int u = 0; u < 8; ++u) {              // unconditional (clamped) loads: no branch per value
                const int i = part + 16 * u < p.pro_tiles ? part + 16 * u : p.pro_tiles - 1;
                pv[mi][u] = p.pro_ssq[(int64_t)gr * p.pro_tiles + i];
            }
        }
        // every one of these loads in FRONT of the W / x loads below: the scheduler otherwise sinks the last one behind
        // them, and its wait (vmcnt(0)) drains the whole prefetch
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (PRO && !kNewPro) {
        float* rstd_lds = reinterpret_cast<float*>(lds + NW * kWaveLdsT);
        if (tid < 512) {
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
                const int r = (tid >> 4) + 32 * mi, part = tid & 15;
                const int gr = m0 + r < p.M ? m0 + r : p.M - 1;
                float t = 0.f;
                for (int i = part; i < p.pro_tiles; i += 16) t += p.pro_ssq[(int64_t)gr * p.pro_tiles + i];
                t += __shfl_xor(t, 1);
                t += __shfl_xor(t, 2);
                t += __shfl_xor(t, 4);
                t += __shfl_xor(t, 8);
                if (part == 0) rstd_lds[r] = rsqrtf(t / (float)p.K + p.pro_eps);      // md_rmsnorm's expression
            }
        }
        __syncthreads();
        if constexpr (MT == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) rs8[i] = rstd_lds[4 * i + ar];
        }
    }
    auto a_load = [&](int c) {
        const int klen = min(ksteps_w - c * 8, 8) * 16;               // k elements of this chunk (wave-uniform)
        const unsigned int cc = (unsigned int)(c * kKC + (c16 * 8 < klen ? c16 : 0) * 8) * 2u;   // lanes past a short
                                                                      // tail chunk re-read its column 0
        if constexpr (kNewPro)        // the chunk's norm weights FIRST: a_store unpacks them before it touches a row
            nwv = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(p.pro_w + ks0 * 16) + cc);
#pragma unroll
        for (int i = 0; i < 8 * MT; ++i) {
            unsigned int ro;
            if constexpr (MT == 1) {
                ro = xrow[i];
            } else {                                                  // recomputed: 16 more live registers would spill
                const int r = m0 + 4 * i + ar;
                ro = (unsigned int)(r < p.M ? r : p.M - 1) * (unsigned int)p.ldx * 2u;
            }
            xa[i] = *reinterpret_cast<const u32x4*>(xbase + (ro + cc));
        }
        if constexpr (PRO && !kNewPro)
            nwv = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(p.pro_w + ks0 * 16) + cc);
    };
    auto a_store = [&]() {
        f32x2 nwf[4];                                 // kNewPro: the chunk's norm weights, unpacked once
        if constexpr (kNewPro) {
#pragma unroll
            for (int w = 0; w < 4; ++w) nwf[w] = unpack2(nwv[w]);
        }
#pragma unroll
        for (int i = 0; i < 8 * MT; ++i) {
            u32x4 v = xa[i];
            if constexpr (kNewPro) {
                // rows 0..31 of the tile from the 16 instructions i = 0..15?  no: instruction i covers rows 4 i + ar of
                // the 64-row image (i < 16), and its scale sits in lane 16 ar + i of the lane's own 16-lane group
                const float rs = row_bcast(my_rs, i);                                    // row_newbcast:i
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const f32x2 t = unpack2(v[w]) * f32x2{rs, rs};
                    unsigned int pk, po;
                    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(t[0]), "v"(t[1]));
                    const f32x2 o = unpack2(pk) * nwf[w];
                    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(po) : "v"(o[0]), "v"(o[1]));
                    v[w] = po;
                }
            } else if constexpr (PRO) {
                float rs;
                if constexpr (MT == 1) rs = rs8[i];
                else rs = reinterpret_cast<const float*>(lds + NW * kWaveLdsT)[4 * i + ar];
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    // y = bf16(bf16(h * rstd) * weight), element by element (two bf16 per dword).  Every column tile of
                    // the consumer repeats this for the same rows (1B w1|w3: 512 times, 18.9 -> 25.8 us), so it is
                    // written for the instruction count: packed fp32 multiplies, and the inner rounding as ONE
                    // v_cvt_pk_bf16_f32 per pair (through the C cast hipcc converts the halves separately)
                    const f32x2 t = unpack2(v[w]) * f32x2{rs, rs};
                    unsigned int pk;
                    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(t[0]), "v"(t[1]));
                    const f32x2 o = unpack2(pk) * unpack2(nwv[w]);
                    v[w] = pack2(o[0], o[1]);
                }
            }
            *reinterpret_cast<u32x4*>(my_lds + (4 * i + ar) * kPitch + c16 * 16) = v;
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    bf16x8 wr[NT][kRD];
    a_load(0);
#pragma unroll
    for (int s = 0; s < kRD; ++s)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const u32x4 v = ld_w<WNT>(wbase[nt] + (s < ksteps_w ? s : ksteps_w - 1) * 512 + lane * 8);
            wr[nt][s] = *reinterpret_cast<const bf16x8*>(&v);
        }
    if constexpr (kNewPro) {
        // the row scales, from the partial sums requested in front of everything above (they return first): the order
        // of the additions is the old prologue's (i = part, part + 16, ...), then the 16-lane butterfly
        float* rstd_lds = reinterpret_cast<float*>(lds + NW * kWaveLdsT);
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            const int r = (tid >> 4) + 32 * mi, part = tid & 15;
            float t = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u)                // pro_tiles <= 128 (K <= 4096): the launcher's condition for this form
                if (part + 16 * u < p.pro_tiles) t += pv[mi][u];
            t += __shfl_xor(t, 1);
            t += __shfl_xor(t, 2);
            t += __shfl_xor(t, 4);
            t += __shfl_xor(t, 8);
            if (part == 0) rstd_lds[r] = rsqrtf(t / (float)p.K + p.pro_eps);      // md_rmsnorm's expression
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the LDS writes only: the global loads stay in flight
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // staging instruction i of a lane covers row 4 i + (lane >> 4): lane 16 a + i keeps the scale of row 4 i + a, so
        // that ONE row_newbcast:i DPP move hands every lane of the 16-lane group a its scale of instruction i -- no LDS
        // read (and no lgkmcnt wait) per staged row
        my_rs = rstd_lds[4 * (lane & 15) + (lane >> 4)];
    }
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned char* a_frag = my_lds + j * kPitch + kh * 16;
    MD_TS(1);                                     // every first load is issued
#pragma unroll 1
    for (int c = 0; c < nchunk; ++c) {
        const int nst = min(ksteps_w - c * 8, 8);
        a_store();
#ifdef MD_TILE_TIMING
        if (c == 0) MD_TS(2);                     // the first activation chunk has landed and is staged
#endif
        if (c + 1 < nchunk) a_load(c + 1);
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            const bool live = st < nst;                               // wave-uniform; only the last chunk can be short
            const int nxt = c * 8 + st + kRD;
            bf16x8 b[NT], a[MT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                b[nt] = wr[nt][st % kRD];
                const u32x4 v = ld_w<WNT>(wbase[nt] + (nxt < ksteps_w ? nxt : ksteps_w - 1) * 512 + lane * 8);
                wr[nt][st % kRD] = *reinterpret_cast<const bf16x8*>(&v);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const bf16x8*>(a_frag + mt * 32 * kPitch + st * 32);
            if (!live) {                                              // stale LDS may hold NaN: 0 x 0, not 0 x garbage
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a[mt] = zero8;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) b[nt] = zero8;
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
        }
    }

    // ---- the NW partial tiles (x MT x NT sub-tiles) -> LDS (each wavefront overwrites its own, fully consumed,
    // activation image).  acc[mt][nt][r] = D[row mt*32 + (r&3) + 8*(r>>2) + 4*kh][column nt*32 + j]
    float* red = reinterpret_cast<float*>(my_lds);
    MD_TS(3);                                     // the wavefront's K slice is consumed (last W fragment has landed)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                red[(mt * NT + nt) * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * kh) * 32 + j] = acc[mt][nt][r];
    __syncthreads();
    MD_TS(4);                                     // every wavefront's partial tile is in LDS

#ifdef MD_TILE_TIMING
    auto ts_flush = [&]() {
        if (p.dbg && lane == 0) {
            ts[5] = wall_clock64();
            unsigned long long* d = p.dbg + ((int64_t)blockIdx.x * kNW + wave) * 6;
#pragma unroll
            for (int i = 0; i < 6; ++i) d[i] = ts[i];
        }
    };
    if (NW > 8 && tid >= 512) ts_flush();
#endif
    if (NW > 8 && tid >= 512) return;                     // 512 threads finish each 32 x 32 sub-tile (one column pair each)
    const int row = tid >> 4, cp = tid & 15;
#pragma unroll 1
    for (int sub = 0; sub < MT * NT; ++sub) {
        const int tn = tn0 + sub % NT;
        const int gm = m0 + 32 * (sub / NT) + row;
        const unsigned char* part = lds + sub * 4096;         // this sub-tile inside every wave's image
        auto tile_sum2 = [&](int col) -> f32x2 {              // columns col, col+1 of row `row`, summed in wave order
            f32x2 s = {0.f, 0.f};
#pragma unroll
            for (int w = 0; w < kNW; ++w)
                s += *reinterpret_cast<const f32x2*>(part + w * kWaveLdsT + (row * 32 + col) * 4);
            return s;
        };
        auto tile_sum1 = [&](int col) -> float {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < kNW; ++w) s += *reinterpret_cast<const float*>(part + w * kWaveLdsT + (row * 32 + col) * 4);
            return s;
        };

        if constexpr (EPI == FL_PARTIAL) {
            // this workgroup's K slice of the tile, fp32, into plane `slice`: 16 lanes write one 128-B row segment
            const f32x2 s = tile_sum2(2 * cp);
            if (gm < p.M)
                *reinterpret_cast<f32x2*>(p.partial + ((int64_t)slice * p.M + gm) * p.N + tn * 32 + 2 * cp) = s;
            continue;
        } else if constexpr (EPI == FL_SWIGLU) {
            const int I = p.N >> 1;
            const int i = tn * 16 + cp;
            const float h1 = bf16_to_f32(f32_to_bf16(tile_sum1(cp)));
            const float h3 = bf16_to_f32(f32_to_bf16(tile_sum1(16 + cp)));
            if (gm < p.M && i < I) p.out[(int64_t)gm * p.ldo + i] = f32_to_bf16(silu_bf16(h1) * h3);
            continue;
        } else {
            const int n = tn * 32 + 2 * cp;
            f32x2 s = tile_sum2(2 * cp);
            if (p.bias) {
                s[0] += bf16_to_f32(p.bias[n]);
                s[1] += bf16_to_f32(p.bias[n + 1]);
            }
            if (gm >= p.M) continue;
            // the linear's own output, rounded to bf16 as nn.Linear returns it
            const float o0 = bf16_to_f32(f32_to_bf16(s[0])), o1 = bf16_to_f32(f32_to_bf16(s[1]));
            if constexpr (EPI == FL_NONE) {
                *reinterpret_cast<unsigned int*>(p.out + (int64_t)gm * p.ldo + n) = pack2(o0, o1);
            } else if constexpr (EPI == FL_RESID) {
                const unsigned int rv = *reinterpret_cast<const unsigned int*>(p.resid + (int64_t)gm * p.ldr + n);
                const float r0 = __uint_as_float(rv << 16), r1 = __uint_as_float(rv & 0xffff0000u);
                const float h0 = bf16_to_f32(f32_to_bf16(r0 + o0)), h1 = bf16_to_f32(f32_to_bf16(r1 + o1));
                *reinterpret_cast<unsigned int*>(p.out + (int64_t)gm * p.ldo + n) = pack2(h0, h1);
                if (p.ssq_out) {
                    // sum of squares of this tile's 32 columns of row gm (the 16 lanes of the row, fixed butterfly order)
                    float q = h0 * h0 + h1 * h1;
                    q += __shfl_xor(q, 1);
                    q += __shfl_xor(q, 2);
                    q += __shfl_xor(q, 4);
                    q += __shfl_xor(q, 8);
                    if (cp == 0) p.ssq_out[(int64_t)gm * p.n_tiles + tn] = q;
                }
            } else {                                            // FL_ROPE_APPEND
                const int HD = p.H * p.D, KD = p.KH * p.D;
                const int b = gm / p.rows_per_req, jrow = gm - b * p.rows_per_req;
                const bool is_v = n >= HD + KD;
                float y0 = o0, y1 = o1;
                if (!is_v) {
                    int pos = p.offsets[b] + jrow;
                    pos = pos < 0 ? 0 : (pos >= p.max_pos ? p.max_pos - 1 : pos);
                    const int d = (n < HD ? n : n - HD) % p.D;          // even: (d, d+1) is one interleaved pair
                    const f32x2 cs = *reinterpret_cast<const f32x2*>(p.cos_sin + (int64_t)pos * p.D + d);
                    y0 = __fsub_rn(__fmul_rn(o0, cs[0]), __fmul_rn(o1, cs[1]));
                    y1 = __fadd_rn(__fmul_rn(o1, cs[0]), __fmul_rn(o0, cs[1]));
                }
                if (n < HD) {
                    *reinterpret_cast<unsigned int*>(p.out + (int64_t)gm * p.ldo + n) = pack2(y0, y1);
                    continue;
                }
                const int nn = is_v ? n - HD - KD : n - HD;
                const int h = nn / p.D, d = nn - h * p.D;
                const int64_t half = (int64_t)p.page_size * KD;
                bool over;
                const int64_t d1 = kv_elem_offset(p.t1, b, p.rows_per_req, jrow, p.page_size, p.KH, p.D, h, d, p.hnd != 0,
                                                  &over);
                // one count per dropped row (as md_rope_append): the thread holding the row's first K pair reports it
                if (over && !is_v && nn == 0) atomicAdd(p.overflow, 1u);
                // what the cache receives is the bf16 tensor k / v of the reference (RoPE output rounded to bf16) -- an fp8
                // page quantises THAT value, not the fp32 rotation result (md_rope_append does the same)
                y0 = bf16_to_f32(f32_to_bf16(y0));
                y1 = bf16_to_f32(f32_to_bf16(y1));
                if (d1 >= 0) {
                    float inv = 1.f;
                    if constexpr (FP8) inv = 1.0f / (is_v ? p.v_scale[h] : p.k_scale[h]);
                    store_pair<FP8>(p.t1.cache, d1 + (is_v ? half : 0), y0, y1, inv);
                }
                if (p.t2.cache) {                                 // second cache (self-speculation draft cache): bf16, NHD
                    const int64_t d2 = kv_elem_offset(p.t2, b, p.rows_per_req, jrow, p.page_size, p.KH, p.D, h, d, false,
                                                      &over);
                    if (over && !is_v && nn == 0) atomicAdd(p.overflow, 1u);
                    if (d2 >= 0) store_pair<false>(p.t2.cache, d2 + (is_v ? half : 0), y0, y1, 1.f);
                }
            }
        }
    }
#ifdef MD_TILE_TIMING
    ts_flush();
#endif
}

template <int EPI, bool FP8, int NW, bool WNT, bool PRO, int MT = 1, int NT = 1>
int launch_tile_cfg(const TileParams& p, hipStream_t st) {
    constexpr int lds = NW * MT * kWaveLds + (PRO ? 128 * MT : 0);
    auto k = tile_gemm_kernel<EPI, FP8, NW, WNT, PRO, MT, NT>;
    static MdPerDeviceOnce once;
    if (once.first()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
            hipSuccess) {
            once.undo();
            md_set_error("md_linear_fused: hipFuncSetAttribute(%d B LDS) failed", lds);
            return MD_ERR_LAUNCH;
        }
    }
    const int m_groups = (p.m_tiles + MT - 1) / MT, n_groups = p.n_tiles / NT;
    const int grid = ((n_groups + 7) / 8) * 8 * m_groups;
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * NW), lds, st, p);
    return MD_OK;
}

unsigned long long* g_tile_dbg = nullptr;    // dev knob (md_debug_set_tile_timing): per-wave phase timestamps

int g_force_nw = 0;   // dev knob (md_debug_set_fused_nw): 0 = the rule below, 8 / 16 = forced where the shape allows

int g_force_tile = 0; // dev knob (md_debug_set_fused_nw 11 / 22): force 1 x 1 / 2 x 2 tiles where the shape allows

template <int EPI, bool FP8, bool PRO>
int launch_tile_pro(const TileParams& p, hipStream_t st) {
    // 2 x 2 tiles (64 rows x 64 columns per workgroup, one per CU): when the M range has two 32-row tiles and the 64-column
    // groups alone still give (nearly) every CU a workgroup -- the ingest-bound wide products of a 64-row step (the 1B
    // w1|w3: 256 groups).  Narrow products keep 1 x 1 tiles: more, smaller workgroups matter more there.
    // (not with the deferred-RMSNorm prologue: that instantiation sits at the 256-register cap, re-reads rstd from LDS per
    // chunk, and measured 25.9 us in the cfg3 iteration against 24.5 for the 1 x 1 form -- profiles/r04_bench_cfg3_iter_breakdown.csv)
    // round 6: also WITH the deferred-RMSNorm prologue (SwiGLU consumer), now that that instantiation has a four-deep W
    // ring, no spills, and its own prologue (kNewPro in the kernel)
    if constexpr ((EPI == FL_SWIGLU || EPI == FL_NONE || EPI == FL_RESID) && (!PRO || EPI == FL_SWIGLU)) {
        const int groups = (p.n_tiles / 2) * ((p.m_tiles + 1) / 2);
        bool t22 = p.m_tiles >= 2 && p.n_tiles % 2 == 0 && p.K % 128 == 0 && groups >= 192;
        if (g_force_tile == 11) t22 = false;
        if (g_force_tile == 22) t22 = p.n_tiles % 2 == 0 && p.K % 128 == 0;
        if (PRO && p.pro_tiles > 128) t22 = false;       // the 2 x 2 prologue keeps <= 8 partial sums per thread in flight
        if (t22) return launch_tile_cfg<EPI, FP8, 8, false, PRO, 2, 2>(p, st);
    }
    // 16 wavefronts (K/16 slices) when the grid is too small to put two 8-wave workgroups on every CU
    // (round 6: not below K = 1024 -- two k-steps per wavefront and a 16-way reduce cost more than they hide: the K = 512 wo
    // shards run 4.1-4.2 us with 8 slices against 5.3-5.4 with 16, tools/shard_bench.py / profiles/r06_shard_ab.txt)
    const int wgs = p.n_tiles * p.m_tiles;
    bool nw16 = p.K % 256 == 0 && p.K >= 1024 && wgs <= 384;
    if (g_force_nw == 8) nw16 = false;
    if (g_force_nw == 16) nw16 = p.K % 256 == 0;
    const bool wnt = p.m_tiles == 1;
    if (nw16)
        return wnt ? launch_tile_cfg<EPI, FP8, 16, true, PRO>(p, st) : launch_tile_cfg<EPI, FP8, 16, false, PRO>(p, st);
    return wnt ? launch_tile_cfg<EPI, FP8, 8, true, PRO>(p, st) : launch_tile_cfg<EPI, FP8, 8, false, PRO>(p, st);
}

template <int EPI, bool FP8>
int launch_tile(const TileParams& p, hipStream_t st) {
    if constexpr (EPI == FL_SWIGLU || EPI == FL_ROPE_APPEND) {       // the linears that consume a normalised input
        if (p.pro_ssq) return launch_tile_pro<EPI, FP8, true>(p, st);
    }
    return launch_tile_pro<EPI, FP8, false>(p, st);
}

bool aligned16(const void* q) { return ((uintptr_t)q & 15) == 0; }

// ---- md_linear_fused_split: the tile kernel with the K range ALSO split over S workgroups (FL_PARTIAL) ----------------
// Why (round 6): the deep narrow products of a 64-row draft step -- the 1B w2, N = 2048, K = 8192 -- are per-CU-ingest
// bound on every single-launch decomposition (a workgroup that owns the whole K range re-reads a 32-row x 8192 slab of x
// per 32 columns: 1 MB per CU, 22-24 us; hipBLASLt: 768 KB per CU, 15.7 us + 5 for the add + norm behind it) and
// latency-bound on md_linear (4 waves x 8 KiB of W in flight per CU: 17 + 5).  A 64 x 64 tile x K / 8 per workgroup
// ingests 128 KB of W + 128 KB of x, ALL of it requested in the first instructions of its 8 wavefronts (8 k-steps each:
// the W ring and one activation chunk cover the whole slice), on every CU; the 4 MB of fp32 partial planes stay in L2 /
// MALL for the combine launch, which is the launch that adds the residual and normalises anyway (reduce_add_rmsnorm,
// elementwise.hip: the same launch md_linear_add_rmsnorm ends with, slices added in order -> deterministic).
struct SplitPlan { int mt, nt, S; };
int g_force_split = 0;       // dev knob (md_debug_set_fused_split): force S where the shape allows

SplitPlan split_plan(int M, int N, int K) {
    SplitPlan pl;
    const int m_tiles = (M + 31) / 32, n_tiles = N / 32;
    pl.mt = m_tiles >= 2 ? 2 : 1;
    pl.nt = (n_tiles % 2 == 0 && n_tiles >= 16) ? 2 : 1;
    const int groups = (n_tiles / pl.nt) * ((m_tiles + pl.mt - 1) / pl.mt);
    const int ksteps = K >> 4;
    int best = 1;
    for (int S = 1; S <= 32; ++S) {                  // the smallest S that gives every CU a workgroup, 8 K-slices per
        if (ksteps % (S * 8)) continue;              // workgroup (waves) x S workgroups must divide the k-steps
        best = S;
        if (groups * S >= 256) break;
    }
    pl.S = best;
    if (g_force_split > 0 && ksteps % (g_force_split * 8) == 0) pl.S = g_force_split;
    return pl;
}

template <int MT, int NT>
int launch_split_cfg(const TileParams& p, hipStream_t st) {
    constexpr int lds = 8 * MT * kWaveLds;
    auto k = tile_gemm_kernel<FL_PARTIAL, false, 8, true, false, MT, NT>;
    static MdPerDeviceOnce once;
    if (once.first()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
            hipSuccess) {
            once.undo();
            md_set_error("md_linear_fused_split: hipFuncSetAttribute(%d B LDS) failed", lds);
            return MD_ERR_LAUNCH;
        }
    }
    const int m_groups = (p.m_tiles + MT - 1) / MT, n_groups = p.n_tiles / NT;
    hipLaunchKernelGGL(k, dim3(n_groups * m_groups * p.S), dim3(512), lds, st, p);
    return MD_OK;
}

int launch_split(const TileParams& p, const SplitPlan& pl, hipStream_t st) {
    if (pl.mt == 2) return pl.nt == 2 ? launch_split_cfg<2, 2>(p, st) : launch_split_cfg<2, 1>(p, st);
    return pl.nt == 2 ? launch_split_cfg<1, 2>(p, st) : launch_split_cfg<1, 1>(p, st);
}

int split_check(const char* who, const void* x, int64_t ldx, const void* w, int M, int N, int K, const void* ws,
                size_t ws_bytes) {
    MD_CHECK_ARG(x && w, "%s: null pointer argument", who);
    MD_CHECK_ARG(M >= 1 && M <= 256 && K >= 128 && K % 128 == 0 && N >= 32 && N % 32 == 0,
                 "%s: unsupported shape M=%d N=%d K=%d (need 1 <= M <= 256, K %% 128 == 0, N %% 32 == 0)", who, M, N, K);
    MD_CHECK_ARG(aligned16(x) && aligned16(w) && ldx % 8 == 0, "%s: x / w must be 16-byte aligned, ldx %% 8 == 0", who);
    MD_CHECK_ARG(ldx > 0 && ((int64_t)(M - 1) * ldx + K) * 2 < ((int64_t)1 << 32),
                 "%s: x spans more than 4 GiB (M=%d, ldx=%lld)", who, M, (long long)ldx);
    const SplitPlan pl = split_plan(M, N, K);
    MD_CHECK_ARG(ws && aligned16(ws) && ws_bytes >= (size_t)pl.S * M * N * 4,
                 "%s: workspace of %zu bytes needed (md_linear_fused_split_workspace_bytes), got %zu", who,
                 (size_t)pl.S * M * N * 4, ws_bytes);
    return MD_OK;
}

int split_main(const void* x, int64_t ldx, const void* w, int M, int N, int K, void* ws, hipStream_t st, int* S_out) {
    const SplitPlan pl = split_plan(M, N, K);
    TileParams p = {};
    p.x = (const bf16_t*)x;
    p.w = (const bf16_t*)w;
    p.ldx = ldx;
    p.M = M;
    p.N = N;
    p.K = K;
    p.n_tiles = N / 32;
    p.m_tiles = (M + 31) / 32;
    p.partial = (float*)ws;
    p.S = pl.S;
    p.dbg = g_tile_dbg;
    *S_out = pl.S;
    return launch_split(p, pl, st);
}

}  // namespace

// gemm.hip / elementwise.hip: the split-K combine launches shared with md_linear / md_linear_block
int md_internal_launch_skinny_reduce(const float* partial, int S, int M, int N, int epilogue, const void* bias, void* out,
                                     int64_t ldo, hipStream_t st);
int md_internal_launch_reduce_add_rmsnorm(const float* partial, int S, int M, int N, const void* bias, const void* scales,
                                          const void* x, int64_t ldx, const void* w, void* h_out, void* y, float eps,
                                          hipStream_t st);

#ifdef MD_DEV_KNOBS
extern "C" void md_debug_set_fused_split(int S) { g_force_split = S > 0 ? S : 0; }
extern "C" void md_debug_set_tile_timing(void* buf) { g_tile_dbg = (unsigned long long*)buf; }
#endif

extern "C" size_t md_linear_fused_split_workspace_bytes(int M, int N, int K) {
    if (M < 1 || M > 256 || K < 128 || K % 128 || N < 32 || N % 32) return 0;
    return (size_t)split_plan(M, N, K).S * M * N * 4;
}

extern "C" int md_linear_fused_split(const void* x, int64_t ldx, const void* w_packed, const void* bias, void* out,
                                     int64_t ldo, int M, int N, int K, void* workspace, size_t workspace_bytes,
                                     md_stream_t stream) {
    const int rc0 = split_check("md_linear_fused_split", x, ldx, w_packed, M, N, K, workspace, workspace_bytes);
    if (rc0 != MD_OK) return rc0;
    MD_CHECK_ARG(out && ((uintptr_t)out & 7) == 0 && ldo % 4 == 0, "md_linear_fused_split: out must be 8-byte aligned, ldo %% 4 == 0");
    int S = 1;
    int rc = split_main(x, ldx, w_packed, M, N, K, workspace, (hipStream_t)stream, &S);
    if (rc != MD_OK) return rc;
    rc = md_internal_launch_skinny_reduce((const float*)workspace, S, M, N, 0 /* EPI_NONE */, bias, out, ldo,
                                          (hipStream_t)stream);
    if (rc != MD_OK) return rc;
    MD_CHECK_LAUNCH("md_linear_fused_split");
    return MD_OK;
}

extern "C" int md_linear_fused_split_add_rmsnorm(const void* x, int64_t ldx, const void* w_packed, const void* bias,
                                                 const void* resid, int64_t ldr, const void* norm_weight, float eps,
                                                 void* h_out, void* y_out, int M, int N, int K, void* workspace,
                                                 size_t workspace_bytes, md_stream_t stream) {
    const int rc0 = split_check("md_linear_fused_split_add_rmsnorm", x, ldx, w_packed, M, N, K, workspace, workspace_bytes);
    if (rc0 != MD_OK) return rc0;
    MD_CHECK_ARG(resid && norm_weight && h_out && y_out, "md_linear_fused_split_add_rmsnorm: null pointer argument");
    MD_CHECK_ARG(N % 8 == 0 && N <= 8192 && ldr % 8 == 0,
                 "md_linear_fused_split_add_rmsnorm: N %% 8 == 0, N <= 8192 (a row is normalised by one workgroup), ldr %% 8 == 0");
    MD_CHECK_ARG(aligned16(resid) && aligned16(norm_weight) && aligned16(h_out) && aligned16(y_out),
                 "md_linear_fused_split_add_rmsnorm: resid / norm_weight / h_out / y_out must be 16-byte aligned");
    int S = 1;
    int rc = split_main(x, ldx, w_packed, M, N, K, workspace, (hipStream_t)stream, &S);
    if (rc != MD_OK) return rc;
    rc = md_internal_launch_reduce_add_rmsnorm((const float*)workspace, S, M, N, bias, nullptr, resid, ldr, norm_weight,
                                               h_out, y_out, eps, (hipStream_t)stream);
    if (rc != MD_OK) return rc;
    MD_CHECK_LAUNCH("md_linear_fused_split_add_rmsnorm");
    return MD_OK;
}

#ifdef MD_DEV_KNOBS
extern "C" void md_debug_set_fused_nw(int nw) {
    // 11: 1 x 1 tiles AND 8 K slices (what 2 x 2 uses); 22 pins 8 slices too, for the forms that always stay 1 x 1
    g_force_nw = (nw == 8 || nw == 16) ? nw : ((nw == 11 || nw == 22) ? 8 : 0);
    g_force_tile = (nw == 11 || nw == 22) ? nw : 0;
}
#endif

extern "C" int md_linear_fused_supported(int M, int N, int K, int epilogue) {
    if (M < 1 || M > 256 || K < 128 || K % 128 || N < 32 || N % 32) return 0;
    return (epilogue >= FL_NONE && epilogue <= FL_ROPE_APPEND) ? 1 : 0;
}

extern "C" int md_linear_fused(const md_fused_linear_args* a, md_stream_t stream) {
    MD_CHECK_ARG(a && a->x && a->w_packed && a->out, "md_linear_fused: null pointer argument");
    MD_CHECK_ARG(md_linear_fused_supported(a->M, a->N, a->K, a->epilogue),
                 "md_linear_fused: unsupported shape M=%d N=%d K=%d epilogue=%d (need 1 <= M <= 256, K %% 128 == 0, "
                 "N %% 32 == 0)", a->M, a->N, a->K, a->epilogue);
    MD_CHECK_ARG(aligned16(a->x) && aligned16(a->w_packed) && aligned16(a->out) && a->ldx % 8 == 0 && a->ldo % 2 == 0,
                 "md_linear_fused: x / w / out must be 16-byte aligned, ldx %% 8 == 0, ldo %% 2 == 0");
    MD_CHECK_ARG(!(a->epilogue == FL_SWIGLU && a->bias), "md_linear_fused: the SwiGLU epilogue takes no bias");
    // the kernel addresses x with 32-bit byte offsets from a scalar base
    MD_CHECK_ARG(a->ldx > 0 && ((int64_t)(a->M - 1) * a->ldx + a->K) * 2 < ((int64_t)1 << 32),
                 "md_linear_fused: x spans more than 4 GiB (M=%d, ldx=%lld): pass a compact activation tensor", a->M,
                 (long long)a->ldx);
    TileParams p = {};
    p.x = (const bf16_t*)a->x;
    p.w = (const bf16_t*)a->w_packed;
    p.bias = (const bf16_t*)a->bias;
    p.out = (bf16_t*)a->out;
    p.ldx = a->ldx;
    p.ldo = a->ldo;
    p.M = a->M;
    p.N = a->N;
    p.K = a->K;
    p.n_tiles = a->N / 32;
    p.m_tiles = (a->M + 31) / 32;
    if (a->pro_ssq) {
        MD_CHECK_ARG(a->epilogue == FL_SWIGLU || a->epilogue == FL_ROPE_APPEND,
                     "md_linear_fused: the deferred-RMSNorm prologue exists for the qkv and w1|w3 linears");
        MD_CHECK_ARG(a->pro_norm_w && a->pro_tiles > 0 && aligned16(a->pro_norm_w),
                     "md_linear_fused: the deferred-RMSNorm prologue needs the norm weight (16-byte aligned) and pro_tiles");
        MD_CHECK_ARG(a->pro_tiles * 32 == a->K, "md_linear_fused: pro_ssq must hold K / 32 = %d partial sums per row, got %d",
                     a->K / 32, a->pro_tiles);
        p.pro_ssq = a->pro_ssq;
        p.pro_w = (const bf16_t*)a->pro_norm_w;
        p.pro_eps = a->pro_eps;
        p.pro_tiles = a->pro_tiles;
    }
    MD_CHECK_ARG(!a->ssq_out || a->epilogue == FL_RESID, "md_linear_fused: ssq_out belongs to the residual epilogue");
    p.ssq_out = a->ssq_out;
    p.dbg = g_tile_dbg;
    hipStream_t st = (hipStream_t)stream;
    int rc = MD_OK;
    switch (a->epilogue) {
    case FL_NONE:
        rc = launch_tile<FL_NONE, false>(p, st);
        break;
    case FL_SWIGLU:
        rc = launch_tile<FL_SWIGLU, false>(p, st);
        break;
    case FL_RESID:
        MD_CHECK_ARG(a->resid && ((uintptr_t)a->resid & 3) == 0 && a->ldr % 2 == 0,
                     "md_linear_fused: the residual epilogue needs `resid` (4-byte aligned, ldr %% 2 == 0)");
        p.resid = (const bf16_t*)a->resid;
        p.ldr = a->ldr;
        rc = launch_tile<FL_RESID, false>(p, st);
        break;
    default: {
        MD_CHECK_ARG(a->H > 0 && a->KH > 0 && (a->D == 64 || a->D == 128) && a->N == (a->H + 2 * a->KH) * a->D,
                     "md_linear_fused: rope+append needs N == (H + 2 KH) * D with D in {64, 128} (H=%d KH=%d D=%d N=%d)",
                     a->H, a->KH, a->D, a->N);
        MD_CHECK_ARG(a->rows_per_req > 0 && a->M % a->rows_per_req == 0 && a->offsets && a->cos_sin && a->max_pos > 0 &&
                         a->page_size > 0,
                     "md_linear_fused: rope+append needs rows_per_req | M, offsets, the RoPE table and page_size");
        MD_CHECK_ARG(a->cache && a->page_indices && a->page_indptr && a->last_page_len,
                     "md_linear_fused: rope+append needs the paged cache and its page table");
        MD_CHECK_ARG(!a->cache2 || (a->page_indices2 && a->page_indptr2 && a->last_page_len2),
                     "md_linear_fused: the second cache needs its page table");
        MD_CHECK_ARG(a->ldo == (int64_t)a->H * a->D, "md_linear_fused: q_out must be contiguous [M][H*D]");
        int kvd = a->kv_dtype;
        p.hnd = (kvd & MD_KV_LAYOUT_HND) ? 1 : 0;
        MD_CHECK_ARG((kvd & ~(MD_KV_DTYPE_MASK | MD_KV_LAYOUT_HND)) == 0, "md_linear_fused: unknown kv_dtype flags");
        kvd &= MD_KV_DTYPE_MASK;
        MD_CHECK_ARG(kvd == MD_KV_BF16 || (kvd == MD_KV_FP8_E4M3 && a->k_scale && a->v_scale),
                     "md_linear_fused: kv_dtype must be MD_KV_BF16 or MD_KV_FP8_E4M3 (with per-head scales)");
        p.H = a->H;
        p.KH = a->KH;
        p.D = a->D;
        p.rows_per_req = a->rows_per_req;
        p.max_pos = a->max_pos;
        p.page_size = a->page_size;
        p.offsets = a->offsets;
        p.cos_sin = a->cos_sin;
        p.t1 = KvTable{a->cache, a->page_indices, a->page_indptr, a->last_page_len};
        p.t2 = KvTable{a->cache2, a->page_indices2, a->page_indptr2, a->last_page_len2};
        p.k_scale = a->k_scale;
        p.v_scale = a->v_scale;
        p.overflow = md_page_overflow_counter_device();
        MD_CHECK_ARG(p.overflow, "md_linear_fused: cannot resolve the page-overflow counter");
        rc = kvd == MD_KV_FP8_E4M3 ? launch_tile<FL_ROPE_APPEND, true>(p, st) : launch_tile<FL_ROPE_APPEND, false>(p, st);
    }
    }
    if (rc != MD_OK) return rc;
    MD_CHECK_LAUNCH("md_linear_fused");
    return MD_OK;
}
