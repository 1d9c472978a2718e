// Skinny GEMM for the decode / verify linears of the draft/verify loop (gfx950):  out[M][N] = x[M][K] . W[N][K]^T
//
// replaces: the nn.Linear calls of Attention / FeedForward / the lm head in a decode step
//           (Engine/SnapKV/model.py:288-289,446-455: wqkv, wo, w1|w3, w2, output), M = batch x rows-per-request <= 256.
//
// Why not a library GEMM: at M <= 256 these products are WEIGHT-STREAMING problems (the activations are a few hundred
// KB, the weights 10..500 MB per call and read exactly once), and for the narrow projections (N = 2048..6144) a
// tile-per-workgroup GEMM has too few workgroups to pull HBM bandwidth (hipBLASLt: 1.0-1.4 TB/s on the 8B qkv / wo at
// M = 256, profiles/r01_bench_cfg3_iter_breakdown.csv).  This kernel is organised like the verify-attention kernel
// instead -- W plays the role of K/V:
//   * a wavefront owns 32 output columns (32 rows of W) and a K range; it streams those rows STRAIGHT from global
//     memory into registers in the MFMA B-operand layout (lane (j, kh) loads the 16 B  W[n0+j][k + kh*8 .. +8]; the
//     next instruction takes the next 32 B of the same rows, so every 128-B line is consumed by four consecutive
//     instructions) with a rolling prefetch ring of RD k-steps -- no LDS round trip and no barrier for W;
//   * the activations go through LDS once per 128-deep slab (double-buffered, one barrier per slab) and are shared
//     by the workgroup's 4 wavefronts (4 x 32 = 128 output columns), read as MFMA A fragments with conflict-free
//     ds_read_b128 (row pitch 272 B);
//   * v_mfma_f32_32x32x16_bf16, fp32 accumulators: M <= 32*MT rows, MT in {1,2,4,8};
//   * split-K across workgroups (grid.y) so that even N = 2048 yields >= 256 workgroups; partial sums are written as
//     fp32 and combined IN A FIXED ORDER by a small second kernel (deterministic: graph replays, eager runs and TP
//     ranks see the same bits), which also applies the epilogue; with one K slice the epilogue runs in the main kernel.
//     (Two alternatives were built, measured and removed: the last-arriving slice combining in-kernel -- one workgroup
//     reading S 32-KB slabs serially is 2x slower than the parallel combine kernel -- and a "short-stream" variant with
//     K split over the waves of one workgroup and the activations loaded straight from global memory in MFMA layout --
//     32-byte pieces of 32 rows per instruction: 1.4-3x slower.  profiles/r02_gemm_ab_*_rejected.txt)
// Epilogues: bias add (Qwen wqkv), SwiGLU (w1|w3: a wavefront takes 16 rows of w1 and the matching 16 rows of w3 as
// its 32 columns, so silu(h1)*h3 needs one cross-lane move; rounding points of the reference: h1, h3 -> bf16,
// silu -> bf16, product -> bf16), weight-only int8 (Engine/quantize.py:72-86: bf16(acc) * bf16 scale -> bf16).
#include "md_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kSlabK = 128;                 // k depth of one LDS activation slab
constexpr int kStepsPerSlab = kSlabK / 16;  // 8 MFMA k-steps (32x32x16) per slab
constexpr int kPitch = kSlabK * 2 + 16;     // LDS row pitch in bytes (272: rows start 4 banks apart)

enum { EPI_NONE = 0, EPI_SWIGLU = 1 };

struct GemmParams {
    const bf16_t* x;      // [M][K], row stride ldx (elements)
    const void* w;        // [N][K] row-major bf16 (or int8 when W8), N = GEMM columns (2*I for SwiGLU: [w1; w3])
    const bf16_t* bias;   // [N] or null
    const bf16_t* scales; // [N] bf16 per-output-channel scales (int8 weights) or null
    bf16_t* out;          // [M][Nout], row stride ldo
    float* partial;       // [S][M][N] fp32 when S > 1
    int64_t ldx, ldo;
    int M, N, K, kblk, S, Nout;
    int packed;           // W is in the fragment-major streaming layout (md_pack_weight layout, see md_linear)
    int skip_reduce;      // md_linear_add_rmsnorm: the caller launches its own combine kernel
    // deferred RMSNorm (PRO = true, md_linear_normed): x is the un-normalised h, see md_linear_fused
    const float* pro_ssq; // [M][pro_tiles] partial sums of squares of the rows of x
    const bf16_t* pro_w;  // RMSNorm weight [K]
    float pro_eps;
    int pro_tiles;
};

__device__ __forceinline__ f32x2 unpack2(unsigned int v) {          // the two bf16 of a dword as fp32 (low half first)
    return f32x2{__uint_as_float(v << 16), __uint_as_float(v & 0xffff0000u)};
}

__device__ __forceinline__ bf16x8 ld_frag(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }

// 8 int8 -> 8 bf16 (exact)
__device__ __forceinline__ bf16x8 cvt_i8x8(const u32x2 v) {
    bf16x8 r;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int b = (int)(signed char)((v[h] >> (8 * e)) & 0xffu);
            r[h * 4 + e] = f32_to_bf16((float)b);
        }
    }
    return r;
}

template <bool W8>
__device__ __forceinline__ bf16x8 ld_w(const void* base, int64_t elem_off) {
    if constexpr (W8) {
        const u32x2 v = __builtin_nontemporal_load(
            reinterpret_cast<const u32x2*>(reinterpret_cast<const signed char*>(base) + elem_off));
        return cvt_i8x8(v);
    } else {
        const u32x4 v = __builtin_nontemporal_load(
            reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(base) + elem_off));
        return *reinterpret_cast<const bf16x8*>(&v);
    }
}

__device__ __forceinline__ float silu_bf16(float h1) {
    // F.silu on a bf16 tensor: computed in fp32, rounded to bf16 (x * sigmoid(x)); same expression as md_silu_mul
    return bf16_to_f32(f32_to_bf16(h1 / (1.0f + expf(-h1))));
}

// One workgroup: NW wavefronts x 32 GEMM columns, rows [0, M), k in [blockIdx.y*kblk, +kblk).
// NW (round 5): 4 by default; 6 or 7 where that makes the grid a whole number of workgroups per CU.  A CU ingests ~24 GB/s
// of HBM whatever is resident on it (profiles/r04_ingest_probe.txt), so a launch ends when the most loaded CU does: the
// 8B w1|w3 (896 wave tiles, two K slices) ran as 448 four-wave workgroups = 192 CUs with two and 64 with one (1 048 KB
// against a mean of 918 KB); as 256 seven-wave workgroups every CU streams 917 KB.  Bits do not change: a wave's tile,
// K range and slab order are the same, only its neighbours in the workgroup differ.
// PRO (round 4): the deferred RMSNorm of md_linear_fused on this kernel's activation path -- x is the un-normalised h
// the residual epilogue of the producing linear wrote together with per-row partial sums of squares; the workgroup forms
// rstd per row (the tile kernel's sum, term for term) and normalises every slab on its way from registers to LDS:
// y = bf16(bf16(h * rstd) * w).  A slab is normalised once per workgroup and shared by its 128 columns, so the cost is a
// few dozen vector instructions per slab -- against a 5 us md_rmsnorm launch in front of every w1|w3 of a draft pass.
template <int MT, int EPI, bool W8, int RD, bool PRO, int NW>
__global__ __launch_bounds__(64 * NW) void skinny_gemm_kernel(const GemmParams p) {
    constexpr int MP = MT * 32;
    constexpr int NT = 64 * NW;                      // threads
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // 2 x [MP][kPitch] (+ MP floats rstd when PRO)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, kh = lane >> 5;
    const int k_beg = blockIdx.y * p.kblk;
    const int nslab = p.kblk / kSlabK;

    // ---- this lane's W row (GEMM column).  SwiGLU: the wave's 32 columns are 16 rows of w1 and the same 16 of w3.
    int col;            // GEMM column index in [0, N)
    int out_col = -1;   // output column this lane is responsible for (epilogue), -1: none
    if constexpr (EPI == EPI_SWIGLU) {
        const int I = p.N >> 1;
        const int i = (blockIdx.x * NW + wave) * 16 + (j & 15);
        col = (j < 16 ? 0 : I) + (i < I ? i : I - 1);
        if (j < 16 && i < I) out_col = i;
    } else {
        const int n = (blockIdx.x * NW + wave) * 32 + j;
        col = n < p.N ? n : p.N - 1;
        if (n < p.N) out_col = n;
    }
    // element offset of this lane's first fragment and the distance between consecutive k-steps.
    // row-major: 32 rows x 32 B per wave instruction.  packed: tile-major [n_tile][k_step][lane][8]: one fully
    // contiguous KiB per wave instruction and one sequential stream per wavefront (the layout HBM likes best).
    // (a workgroup covers NW tiles; the last workgroup may reach past the last tile: re-read that one, like `col`)
    const int ntiles = (EPI == EPI_SWIGLU) ? ((p.N >> 1) + 15) / 16 : (p.N + 31) / 32;
    const int tile = min(blockIdx.x * NW + wave, ntiles - 1);
    const int64_t w_off = p.packed ? ((int64_t)tile * (p.K >> 4) + (k_beg >> 4)) * 512 + lane * 8
                                   : (int64_t)col * p.K + k_beg + kh * 8;
    const int64_t w_step = p.packed ? 512 : 16;

    // ---- activation slab staging: MP rows x 16 chunks of 16 B; thread t takes chunks t, t+NT, ... (NT = 256: exactly
    // MT * 2 each; other NT: the last round is partial)
    constexpr int XCH = (MP * 16 + NT - 1) / NT;   // chunks per thread
    constexpr bool XFULL = (MP * 16) % NT == 0;
    u32x4 xs[XCH];
    u32x4 nwv = {0u, 0u, 0u, 0u};        // PRO: the norm weights of this thread's 8 columns of the slab (c16 = tid & 15)
    const float* rstd_lds = reinterpret_cast<const float*>(lds + 2 * MP * kPitch);
    auto x_load = [&](int slab) {
        if constexpr (PRO)
            nwv = *reinterpret_cast<const u32x4*>(p.pro_w + k_beg + slab * kSlabK + (tid & 15) * 8);
#pragma unroll
        for (int q = 0; q < XCH; ++q) {
            const int c = tid + NT * q;
            const int row = c >> 4, c16 = c & 15;
            // rows >= M (and, when NT does not divide the slab, chunks past it) re-read row M-1 (branch-free: a branch per
            // load would serialise the staging, and those values are never stored)
            const int rr = row < p.M ? row : p.M - 1;
            xs[q] = *reinterpret_cast<const u32x4*>(p.x + (int64_t)rr * p.ldx + k_beg + slab * kSlabK + c16 * 8);
        }
    };
    // PRO: normalise the staged slab in registers.  Called in the MIDDLE of the current slab's MFMA steps (the loads were
    // issued at its top): the vector work runs under the matrix pipeline instead of between the last MFMA and the
    // barrier, where all four waves would wait for it (+5 us on the 8B w1|w3 at 64 rows when it sat in x_store)
    auto x_norm = [&]() {
#pragma unroll
        for (int q = 0; q < XCH; ++q) {
            const int row = XFULL ? (tid + NT * q) >> 4 : min((tid + NT * q) >> 4, MP - 1);
            const float rs = rstd_lds[row];
            u32x4 v = xs[q];
#pragma unroll
            for (int w = 0; w < 4; ++w) {                           // the arithmetic of tile_gemm_kernel's prologue
                const f32x2 t = unpack2(v[w]) * f32x2{rs, rs};
                unsigned int pk;
                asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(t[0]), "v"(t[1]));
                const f32x2 o = unpack2(pk) * unpack2(nwv[w]);
                const bf16x2 ob = {f32_to_bf16(o[0]), f32_to_bf16(o[1])};
                v[w] = *reinterpret_cast<const unsigned int*>(&ob);
            }
            xs[q] = v;
        }
    };
    auto x_store = [&](int buf) {
#pragma unroll
        for (int q = 0; q < XCH; ++q) {
            const int c = tid + NT * q;
            const int row = c >> 4, c16 = c & 15;
            if (XFULL || c < MP * 16)
                *reinterpret_cast<u32x4*>(lds + buf * (MP * kPitch) + row * kPitch + c16 * 16) = xs[q];
        }
    };

    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

    // ---- prologue: W ring (RD k-steps in flight), first activation slab
    bf16x8 wr[RD];
    const int nsteps = nslab * kStepsPerSlab;
#pragma unroll
    for (int s = 0; s < RD; ++s) wr[s] = ld_w<W8>(p.w, w_off + (int64_t)(s < nsteps ? s : nsteps - 1) * w_step);
    x_load(0);
    if constexpr (PRO) {
        // row scales, behind the first loads: rstd = rsqrt(sum_t ssq[row][t] / K + eps), 16 lanes per row, the partial
        // sums added in the tile kernel's order (strided by 16, then the 16-lane butterfly)
        float* rw = reinterpret_cast<float*>(lds + 2 * MP * kPitch);
        constexpr int RPT = MP / 16;                               // rows per thread
        const int part = tid & 15;
        float t[RPT];
        if (tid < 256) {                                           // (wave-uniform) the first four waves, as with NW = 4
        if constexpr (MT <= 2) {
            // <= 64 rows (where the policy uses this form): the partial sums of ALL rows of a thread are requested
            // before the first one is needed -- one L2 round trip per 128 tiles instead of one per 16 rows (a serial
            // loop over the row groups cost the 8B w1|w3 at 64 rows +5 us)
            const float* src[RPT];
#pragma unroll
            for (int it = 0; it < RPT; ++it) {
                const int r = it * 16 + (tid >> 4);
                src[it] = p.pro_ssq + (int64_t)(r < p.M ? r : p.M - 1) * p.pro_tiles;
                t[it] = 0.f;
            }
            for (int i0 = part; i0 < p.pro_tiles; i0 += 128) {
                float v[RPT][8];
#pragma unroll
                for (int it = 0; it < RPT; ++it)
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[it][u] = i0 + 16 * u < p.pro_tiles ? src[it][i0 + 16 * u] : 0.f;
#pragma unroll
                for (int it = 0; it < RPT; ++it)
#pragma unroll
                    for (int u = 0; u < 8; ++u) t[it] += v[it][u];          // ascending tile index: the tile kernel's order
            }
        } else {
#pragma unroll 1
            for (int it = 0; it < RPT; ++it) {
                const int r = it * 16 + (tid >> 4);
                const int gr = r < p.M ? r : p.M - 1;
                float a = 0.f;
                for (int i = part; i < p.pro_tiles; i += 16) a += p.pro_ssq[(int64_t)gr * p.pro_tiles + i];
                t[it] = a;
            }
        }
#pragma unroll
        for (int it = 0; it < RPT; ++it) {
            float a = t[it];
            a += __shfl_xor(a, 1);
            a += __shfl_xor(a, 2);
            a += __shfl_xor(a, 4);
            a += __shfl_xor(a, 8);
            if (part == 0) rw[it * 16 + (tid >> 4)] = rsqrtf(__fadd_rn(__fdiv_rn(a, (float)p.K), p.pro_eps));
        }
        }
        __syncthreads();
        x_norm();
    }
    x_store(0);
    __syncthreads();

    const unsigned char* a_base = lds + (j * kPitch + kh * 16);
    static_assert(RD % kStepsPerSlab == 0, "ring depth must be a whole number of slabs");
    constexpr int SLABS_PER_ITER = RD / kStepsPerSlab;
    for (int slab0 = 0; slab0 < nslab; slab0 += SLABS_PER_ITER) {
#pragma unroll
        for (int ss = 0; ss < SLABS_PER_ITER; ++ss) {
            const int slab = slab0 + ss;
            if (slab < nslab) {                                   // wave-uniform
                const int buf = slab & 1;
                const bool more = slab + 1 < nslab;
                if (more) x_load(slab + 1);
                const unsigned char* a_slab = a_base + buf * (MP * kPitch);
                // A fragments are double-buffered in registers: the MT ds_reads of step st+1 are issued before the MT
                // MFMAs of step st (an LDS round trip is longer than one MFMA; with one wave per SIMD at MT = 8
                // nothing else would cover it)
                bf16x8 af[2][MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    af[0][mt] = *reinterpret_cast<const bf16x8*>(a_slab + mt * 32 * kPitch);
#pragma unroll
                for (int st = 0; st < kStepsPerSlab; ++st) {
                    const int s = ss * kStepsPerSlab + st;        // ring slot (compile time)
                    const bf16x8 b = wr[s];
                    const int nxt = slab * kStepsPerSlab + st + RD;   // k-step the slot is refilled with
                    wr[s] = ld_w<W8>(p.w, w_off + (int64_t)(nxt < nsteps ? nxt : nsteps - 1) * w_step);   // tail: re-touch a hot line
                    if (st + 1 < kStepsPerSlab) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            af[(st + 1) & 1][mt] =
                                *reinterpret_cast<const bf16x8*>(a_slab + mt * 32 * kPitch + (st + 1) * 32);
                    }
                    __builtin_amdgcn_sched_barrier(0);     // keep the reads of step st+1 AHEAD of the MFMAs of step st
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[st & 1][mt], b, acc[mt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (PRO) {
                        if (st == kStepsPerSlab / 2 && more) {
                            x_norm();
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
                if (more) x_store(buf ^ 1);
                __syncthreads();
            }
        }
    }

    // ---- epilogue.  acc[mt][r] = D[row = mt*32 + (r&3) + 8*(r>>2) + 4*kh][column j of the wave]
    if (p.S > 1) {
        float* pp = p.partial + (int64_t)blockIdx.y * p.M * p.N;
        const int n = (EPI == EPI_SWIGLU) ? col : out_col;
        const bool ok = (EPI == EPI_SWIGLU) ? ((blockIdx.x * NW + wave) * 16 + (j & 15) < (p.N >> 1)) : (out_col >= 0);
        if (ok) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    if (row < p.M) pp[(int64_t)row * p.N + n] = acc[mt][r];
                }
        }
        return;
    }
    float bias = 0.f, scale = 1.f;
    if (EPI == EPI_NONE && out_col >= 0) {
        if (p.bias) bias = bf16_to_f32(p.bias[out_col]);
        if (p.scales) scale = bf16_to_f32(p.scales[out_col]);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            float v = acc[mt][r];
            if constexpr (EPI == EPI_SWIGLU) {
                float h = bf16_to_f32(f32_to_bf16(v));
                if (W8) h = bf16_to_f32(f32_to_bf16(h * bf16_to_f32(p.scales[col])));
                const float h3 = __shfl_xor(h, 16);                 // lanes j and j^16 hold w1 / w3 of one column
                v = silu_bf16(h) * h3;
            } else {
                v += bias;
                if (W8) v = bf16_to_f32(f32_to_bf16(v)) * scale;
            }
            if (out_col >= 0 && row < p.M) p.out[(int64_t)row * p.ldo + out_col] = f32_to_bf16(v);
        }
}

// Fixed-order combine of the S fp32 partials + epilogue.  One thread per 4 consecutive output columns of one row.
template <int EPI, bool W8>
__global__ __launch_bounds__(256) void skinny_reduce_kernel(const GemmParams p) {
    const int Nout = p.Nout;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int per_row = Nout / 4;
    if (t >= (int64_t)p.M * per_row) return;
    const int row = (int)(t / per_row), c0 = (int)(t % per_row) * 4;
    const int64_t plane = (int64_t)p.M * p.N;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b3 = {0.f, 0.f, 0.f, 0.f};
    // eight slices per trip, requested together and added in slice order (a load + add per loop trip is one memory round
    // trip per slice)
    for (int s0 = 0; s0 < p.S; s0 += 8) {
        f32x4 va[8], vb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (s0 + u < p.S) {                                     // uniform: no load for a slice that does not exist
                const float* pp = p.partial + (s0 + u) * plane + (int64_t)row * p.N + c0;
                va[u] = *reinterpret_cast<const f32x4*>(pp);
                if constexpr (EPI == EPI_SWIGLU) vb[u] = *reinterpret_cast<const f32x4*>(pp + (p.N >> 1));
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (s0 + u < p.S) {                                     // uniform
                a += va[u];
                if constexpr (EPI == EPI_SWIGLU) b3 += vb[u];
            }
        }
    }
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float v = a[e];
        if constexpr (EPI == EPI_SWIGLU) {
            float h1 = bf16_to_f32(f32_to_bf16(v)), h3 = bf16_to_f32(f32_to_bf16(b3[e]));
            if (W8) {
                h1 = bf16_to_f32(f32_to_bf16(h1 * bf16_to_f32(p.scales[c0 + e])));
                h3 = bf16_to_f32(f32_to_bf16(h3 * bf16_to_f32(p.scales[(p.N >> 1) + c0 + e])));
            }
            v = silu_bf16(h1) * h3;
        } else {
            if (p.bias) v += bf16_to_f32(p.bias[c0 + e]);
            if (W8) v = bf16_to_f32(f32_to_bf16(v)) * bf16_to_f32(p.scales[c0 + e]);
        }
        o[e] = f32_to_bf16(v);
    }
    *reinterpret_cast<bf16x4*>(p.out + (int64_t)row * p.ldo + c0) = o;
}

int g_target_blocks = 256;   // split-K is chosen so that about this many workgroups exist (1 per CU: measured best)
int g_force_nw = 0;          // dev knob (md_debug_set_gemm_waves): 0 = the rule below, 4 / 6 / 7 = forced

int pick_splits(int n_blocks, int K) {
    const int nslab = K / kSlabK;
    int best = 1;
    for (int s = 1; s <= nslab && s <= 64; ++s) {
        if (nslab % s) continue;
        best = s;
        if (n_blocks * s >= g_target_blocks) break;
    }
    return best;
}

// (waves per workgroup, K slices) of a product with `ntiles` wave tiles (32 GEMM columns each; SwiGLU: 16 + 16).
// Four waves and the split of pick_splits unless that leaves the most loaded CU with > 8 % more bytes than the mean AND
// six or seven waves per workgroup with the SAME split (same partial planes, same bits) bring it within 2 %.
struct SkinnyPlan { int nw, S; };
SkinnyPlan plan_of(int ntiles, int K, bool allow_nw) {
    SkinnyPlan pl;
    pl.nw = 4;
    pl.S = pick_splits((ntiles + 3) / 4, K);
    if (g_force_nw == 4 || !allow_nw) return pl;
    // compute units of the current device, queried once per device (a partitioned or smaller part must not be
    // planned as 256 CUs); unknown -> keep the four-wave shape (results are bit-identical either way)
    static int cus_of[16] = {0};
    int devid = 0;
    if (hipGetDevice(&devid) != hipSuccess || devid < 0 || devid >= 16) return pl;
    if (cus_of[devid] == 0) {
        int n = 0;
        cus_of[devid] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, devid) == hipSuccess && n > 0)
                            ? n : -1;
    }
    const int kCUs = cus_of[devid];
    if (kCUs <= 0) return pl;
    auto load = [&](int nw) {                       // bytes of the most loaded CU over the mean (round-robin placement)
        const long wgs = (long)((ntiles + nw - 1) / nw) * pl.S;
        const long per_cu = (wgs + kCUs - 1) / kCUs;
        return (double)per_cu * nw * kCUs / ((double)ntiles * pl.S);
    };
    if (g_force_nw == 6 || g_force_nw == 7) {
        pl.nw = g_force_nw;
        return pl;
    }
    if (load(4) <= 1.08) return pl;
    for (int nw : {7, 6})
        if (load(nw) <= 1.02) {
            pl.nw = nw;
            break;
        }
    return pl;
}

}  // namespace

// elementwise.hip: split-K combine + residual add + RMSNorm (compiled there, without FMA contraction, next to the
// kernels it must reproduce bit for bit)
int md_internal_launch_reduce_add_rmsnorm(const float* partial, int S, int M, int N, const void* bias, const void* scales,
                                          const void* x, int64_t ldx, const void* w, void* h_out, void* y, float eps,
                                          hipStream_t st);

namespace {

int g_force_rd = 0;          // dev knob (md_debug_set_gemm_ring): 0 = the rule in launch(), 8 / 16 = forced where instantiated

template <int MT, int EPI, bool W8, bool PRO, int NW, int RD>
int launch_rd(const GemmParams& p, int n_blocks, hipStream_t st);

// W ring depth (round 6, a measured NEGATIVE result kept as a dev knob).  The four-wave form requests 4 x 8 KiB per CU with
// the 8-deep ring, half of what the six- / seven-wave forms and the tile kernel keep in flight, so a 16-deep ring (there is
// room for it at <= 64 rows) looked like the fix for the K-split products that run four waves (the 8B w2 at 32 / 64 rows,
// the lm heads).  tools/ring_bench.py (profiles/r06_ring_depth_ab.txt): 29.6 vs 29.5, 32.9 vs 33.6, 17.8 vs 18.5, lm heads
// 101 vs 104 / 184 vs 188 us -- equal or 1-3 % slower, bits identical.  More loads per WAVE are not accepted any faster
// (the tile kernel's phase timestamps show the same: a wave that issues 32 loads is still issuing them microseconds
// later); what helps is more waves or fewer bytes per CU.  The rule stays at 8; md_debug_set_gemm_ring(16) selects the
// deep form for the A/B.
template <int MT, int EPI, bool W8, bool PRO, int NW>
int launch(const GemmParams& p, int n_blocks, hipStream_t st) {
    if constexpr (MT <= 2 && NW == 4 && !W8) {
        if (g_force_rd == 16) return launch_rd<MT, EPI, W8, PRO, NW, 16>(p, n_blocks, st);
    }
    return launch_rd<MT, EPI, W8, PRO, NW, 8>(p, n_blocks, st);
}

template <int MT, int EPI, bool W8, bool PRO, int NW, int RD>
int launch_rd(const GemmParams& p, int n_blocks, hipStream_t st) {
    const size_t lds = (size_t)2 * MT * 32 * kPitch + (PRO ? MT * 32 * 4 : 0);
    auto k = skinny_gemm_kernel<MT, EPI, W8, RD, PRO, NW>;
    if (lds > 64 * 1024) {
        static MdPerDeviceOnce once;   // per (MT, EPI, W8, PRO, NW) instantiation and per device
        if (once.first()) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds) != hipSuccess) {
                md_set_error("md_linear: hipFuncSetAttribute(%zu B LDS) failed", lds);
                return MD_ERR_LAUNCH;
            }
        }
    }
    hipLaunchKernelGGL(k, dim3(n_blocks, p.S), dim3(64 * NW), lds, st, p);
    if (p.S > 1 && !p.skip_reduce) {
        const int64_t threads = (int64_t)p.M * (p.Nout / 4);
        hipLaunchKernelGGL((skinny_reduce_kernel<EPI, W8>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, p);
    }
    return MD_OK;
}

// the 6- / 7-wave forms exist for bf16 weights at M <= 128 (nw_allowed below); everything else runs four waves
template <int MT, int EPI, bool W8, bool PRO>
int launch_nw(const GemmParams& p, int nw, int n_blocks, hipStream_t st) {
    if constexpr (!W8 && MT <= 4) {
        if (nw == 7) return launch<MT, EPI, W8, PRO, 7>(p, n_blocks, st);
        if (nw == 6) return launch<MT, EPI, W8, PRO, 6>(p, n_blocks, st);
    }
    return launch<MT, EPI, W8, PRO, 4>(p, n_blocks, st);
}

template <int EPI, bool W8, bool PRO = false>
int launch_mt(const GemmParams& p, int nw, int n_blocks, hipStream_t st) {
    if (p.M <= 32) return launch_nw<1, EPI, W8, PRO>(p, nw, n_blocks, st);
    if (p.M <= 64) return launch_nw<2, EPI, W8, PRO>(p, nw, n_blocks, st);
    if (p.M <= 128) return launch_nw<4, EPI, W8, PRO>(p, nw, n_blocks, st);
    return launch_nw<8, EPI, W8, PRO>(p, nw, n_blocks, st);
}

bool nw_allowed(int M, bool w8) { return !w8 && M <= 128; }
int wave_tiles(int N, int epilogue) { return epilogue == EPI_SWIGLU ? ((N >> 1) + 15) / 16 : (N + 31) / 32; }

}  // namespace

// blockgemm.hip: the fixed-order combine of fp32 partial slabs [S][M][N] (+ bias / SwiGLU) for md_linear_block
int md_internal_launch_skinny_reduce(const float* partial, int S, int M, int N, int epilogue, const void* bias, void* out,
                                     int64_t ldo, hipStream_t st) {
    GemmParams p = {};
    p.partial = const_cast<float*>(partial);
    p.S = S;
    p.M = M;
    p.N = N;
    p.Nout = epilogue == EPI_SWIGLU ? N / 2 : N;
    p.bias = (const bf16_t*)bias;
    p.out = (bf16_t*)out;
    p.ldo = ldo;
    const int64_t threads = (int64_t)M * (p.Nout / 4);
    const dim3 grid((unsigned)((threads + 255) / 256));
    if (epilogue == EPI_SWIGLU)
        hipLaunchKernelGGL((skinny_reduce_kernel<EPI_SWIGLU, false>), grid, dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL((skinny_reduce_kernel<EPI_NONE, false>), grid, dim3(256), 0, st, p);
    return MD_OK;
}

#ifdef MD_DEV_KNOBS
extern "C" void md_debug_set_gemm_target_blocks(int n) { g_target_blocks = n > 0 ? n : 256; }
extern "C" void md_debug_set_gemm_waves(int nw) { g_force_nw = (nw == 4 || nw == 6 || nw == 7) ? nw : 0; }
extern "C" void md_debug_set_gemm_ring(int rd) { g_force_rd = (rd == 8 || rd == 16) ? rd : 0; }
#endif

extern "C" size_t md_linear_workspace_bytes(int M, int N, int K, int epilogue) {
    if (M <= 0 || N <= 0 || K <= 0 || K % kSlabK) return 0;
    const int S = plan_of(wave_tiles(N, epilogue), K, false).S;     // the split does not depend on the waves per workgroup
    return S > 1 ? (size_t)S * M * N * 4 : 0;     // fp32 partial sums of the K slices
}

extern "C" int md_linear_supported(int M, int N, int K, int epilogue) {
    if (M < 1 || M > 256 || K < kSlabK || K % kSlabK) return 0;
    if (epilogue == EPI_SWIGLU) return (N % 32 == 0) ? 1 : 0;     // N = 2*I, I % 16 == 0
    return (N % 4 == 0) ? 1 : 0;
}

namespace {
int linear_impl(const void* x, int64_t ldx, const void* w, int w_dtype, int w_packed, const void* scales,
                const void* bias, void* out, int64_t ldo, int M, int N, int K, int epilogue, void* workspace,
                size_t workspace_bytes, md_stream_t stream, const float* pro_ssq, int pro_tiles, const void* pro_w,
                float pro_eps) {
    MD_CHECK_ARG(x && w && out, "md_linear: null pointer argument");
    MD_CHECK_ARG(md_linear_supported(M, N, K, epilogue), "md_linear: unsupported shape M=%d N=%d K=%d epilogue=%d "
                 "(need 1 <= M <= 256, K %% 128 == 0, N %% 4 == 0)", M, N, K, epilogue);
    MD_CHECK_ARG(epilogue == EPI_NONE || epilogue == EPI_SWIGLU, "md_linear: unknown epilogue %d", epilogue);
    MD_CHECK_ARG(w_dtype == MD_W_BF16 || (w_dtype == MD_W_INT8 && scales), "md_linear: w_dtype must be MD_W_BF16 or "
                 "MD_W_INT8 (with per-channel scales)");
    MD_CHECK_ARG(!(epilogue == EPI_SWIGLU && bias), "md_linear: SwiGLU epilogue takes no bias");
    MD_CHECK_ARG((((uintptr_t)x | (uintptr_t)w | (uintptr_t)out) & 15) == 0 && ldx % 8 == 0 && ldo % 4 == 0,
                 "md_linear: x / w / out must be 16-byte aligned, ldx %% 8 == 0, ldo %% 4 == 0");
    GemmParams p = {};
    p.x = (const bf16_t*)x;
    p.w = w;
    p.bias = (const bf16_t*)bias;
    p.scales = (const bf16_t*)scales;
    p.out = (bf16_t*)out;
    p.ldx = ldx;
    p.ldo = ldo;
    p.M = M;
    p.N = N;
    p.K = K;
    p.Nout = epilogue == EPI_SWIGLU ? N / 2 : N;
    p.packed = w_packed ? 1 : 0;
    p.skip_reduce = 0;
    const bool pro = pro_ssq != nullptr;
    if (pro) {
        MD_CHECK_ARG(w_dtype == MD_W_BF16, "md_linear_normed: bf16 weights only");
        MD_CHECK_ARG(pro_w && (((uintptr_t)pro_w) & 15) == 0 && (((uintptr_t)pro_ssq) & 3) == 0,
                     "md_linear_normed: the norm weight must be 16-byte aligned");
        MD_CHECK_ARG(pro_tiles > 0 && pro_tiles * 32 == K,
                     "md_linear_normed: ssq must hold K / 32 = %d partial sums per row, got %d", K / 32, pro_tiles);
        p.pro_ssq = pro_ssq;
        p.pro_w = (const bf16_t*)pro_w;
        p.pro_eps = pro_eps;
        p.pro_tiles = pro_tiles;
    }
    const bool w8 = w_dtype == MD_W_INT8;
    const int ntiles = wave_tiles(N, epilogue);
    const SkinnyPlan pl = plan_of(ntiles, K, nw_allowed(M, w8));
    const int n_blocks = (ntiles + pl.nw - 1) / pl.nw;
    p.S = pl.S;
    p.kblk = K / p.S;
    p.partial = (float*)workspace;
    if (p.S > 1) {
        MD_CHECK_ARG(workspace && workspace_bytes >= (size_t)p.S * M * N * 4 && (((uintptr_t)workspace) & 15) == 0,
                     "md_linear: workspace too small (need %zu bytes) or not 16-byte aligned", (size_t)p.S * M * N * 4);
    }
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (pro)
        rc = epilogue == EPI_SWIGLU ? launch_mt<EPI_SWIGLU, false, true>(p, pl.nw, n_blocks, st)
                                    : launch_mt<EPI_NONE, false, true>(p, pl.nw, n_blocks, st);
    else if (epilogue == EPI_SWIGLU)
        rc = w8 ? launch_mt<EPI_SWIGLU, true>(p, pl.nw, n_blocks, st) : launch_mt<EPI_SWIGLU, false>(p, pl.nw, n_blocks, st);
    else
        rc = w8 ? launch_mt<EPI_NONE, true>(p, pl.nw, n_blocks, st) : launch_mt<EPI_NONE, false>(p, pl.nw, n_blocks, st);
    if (rc != MD_OK) return rc;
    MD_CHECK_LAUNCH("md_linear");
    return MD_OK;
}
}  // namespace

extern "C" int md_linear(const void* x, int64_t ldx, const void* w, int w_dtype, int w_packed, const void* scales,
                         const void* bias, void* out, int64_t ldo, int M, int N, int K, int epilogue, void* workspace,
                         size_t workspace_bytes, md_stream_t stream) {
    return linear_impl(x, ldx, w, w_dtype, w_packed, scales, bias, out, ldo, M, N, K, epilogue, workspace, workspace_bytes,
                       stream, nullptr, 0, nullptr, 0.f);
}

extern "C" int md_linear_normed(const void* h, int64_t ldh, const float* ssq, int ssq_tiles, const void* norm_weight,
                                float eps, const void* w, int w_packed, const void* bias, void* out, int64_t ldo, int M,
                                int N, int K, int epilogue, void* workspace, size_t workspace_bytes, md_stream_t stream) {
    MD_CHECK_ARG(ssq && norm_weight, "md_linear_normed: null pointer argument");
    return linear_impl(h, ldh, w, MD_W_BF16, w_packed, nullptr, bias, out, ldo, M, N, K, epilogue, workspace,
                       workspace_bytes, stream, ssq, ssq_tiles, norm_weight, eps);
}

extern "C" int md_linear_add_rmsnorm_supported(int M, int N, int K) {
    if (!md_linear_supported(M, N, K, EPI_NONE) || N % 8 || N > 8192) return 0;
    return plan_of(wave_tiles(N, EPI_NONE), K, false).S > 1 ? 1 : 0;      // needs the split-K combine launch to fuse into
}

extern "C" int md_linear_add_rmsnorm(const void* x, int64_t ldx, const void* w, int w_dtype, int w_packed,
                                     const void* scales, const void* bias, const void* resid, int64_t ldr,
                                     const void* norm_weight, float eps, void* h_out, void* y_out, int M, int N, int K,
                                     void* workspace, size_t workspace_bytes, md_stream_t stream) {
    MD_CHECK_ARG(x && w && resid && norm_weight && h_out && y_out, "md_linear_add_rmsnorm: null pointer argument");
    MD_CHECK_ARG(md_linear_add_rmsnorm_supported(M, N, K),
                 "md_linear_add_rmsnorm: unsupported shape M=%d N=%d K=%d (md_linear shape with a split K, N %% 8 == 0, "
                 "N <= 8192)", M, N, K);
    MD_CHECK_ARG(w_dtype == MD_W_BF16 || (w_dtype == MD_W_INT8 && scales), "md_linear_add_rmsnorm: w_dtype must be "
                 "MD_W_BF16 or MD_W_INT8 (with per-channel scales)");
    MD_CHECK_ARG((((uintptr_t)x | (uintptr_t)w | (uintptr_t)resid | (uintptr_t)norm_weight | (uintptr_t)h_out |
                   (uintptr_t)y_out) & 15) == 0 && ldx % 8 == 0 && ldr % 8 == 0,
                 "md_linear_add_rmsnorm: pointers must be 16-byte aligned, ldx %% 8 == 0, ldr %% 8 == 0");
    GemmParams p = {};
    p.x = (const bf16_t*)x;
    p.w = w;
    p.bias = nullptr;
    p.scales = nullptr;
    p.out = nullptr;
    p.ldx = ldx;
    p.ldo = N;
    p.M = M;
    p.N = N;
    p.K = K;
    p.Nout = N;
    p.packed = w_packed ? 1 : 0;
    p.skip_reduce = 1;
    const bool w8 = w_dtype == MD_W_INT8;
    const int ntiles = wave_tiles(N, EPI_NONE);
    const SkinnyPlan pl = plan_of(ntiles, K, nw_allowed(M, w8));
    const int n_blocks = (ntiles + pl.nw - 1) / pl.nw;
    p.S = pl.S;
    p.kblk = K / p.S;
    p.partial = (float*)workspace;
    MD_CHECK_ARG(workspace && workspace_bytes >= (size_t)p.S * M * N * 4 && (((uintptr_t)workspace) & 15) == 0,
                 "md_linear_add_rmsnorm: workspace too small (need %zu bytes) or not 16-byte aligned",
                 (size_t)p.S * M * N * 4);
    hipStream_t st = (hipStream_t)stream;
    int rc = w8 ? launch_mt<EPI_NONE, true>(p, pl.nw, n_blocks, st) : launch_mt<EPI_NONE, false>(p, pl.nw, n_blocks, st);
    if (rc != MD_OK) return rc;
    rc = md_internal_launch_reduce_add_rmsnorm(p.partial, p.S, M, N, bias, w8 ? scales : nullptr, resid, ldr,
                                               norm_weight, h_out, y_out, eps, st);
    if (rc != MD_OK) return rc;
    MD_CHECK_LAUNCH("md_linear_add_rmsnorm");
    return MD_OK;
}
