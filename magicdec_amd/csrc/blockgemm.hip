// Block-tile GEMM for the 129..256-row linears of a verify step (gfx950):  out = epilogue(x[M][K] . W[N][K]^T + bias)
//
// replaces: the nn.Linear calls of the (gamma+1)-token verify pass at B x (gamma+1) = 256 rows
//           (Engine/SnapKV/model.py:288-289 wqkv / wo, :446-455 w1 / w3 / w2, :175-177 output), which rounds 1-3 left
//           to hipBLASLt -- 5.1 ms of a 31 ms iteration at 20-43 % of HBM / <= 35 % of the MFMA peak
//           (profiles/r03_bench_cfg3_iter_breakdown.csv).
//
// Why a third GEMM kernel.  At M = 256 the products sit on the MFMA / HBM ridge (256 flop per weight byte): md_linear
// (gemm.hip: one wave = 32 columns x all rows, W straight into registers, x re-read from LDS per 32 columns) pays
// 1 + 1/8 LDS fragment reads per MFMA and is LDS-bound there; md_linear_fused (tilegemm.hip: 32 x 32 tiles) re-reads
// the weights 8 times through L2.  This kernel is the classic block-tile form, built around what bounds it on CDNA4
// -- LDS bytes per flop and the per-CU load path:
//   * one workgroup = 256 rows (all of M) x 128 columns x a K range; 4 MFMA wavefronts as 2 (M) x 2 (N), one per SIMD,
//     each owning a 128 x 64 output tile = 4 x 2 accumulators of v_mfma_f32_32x32x16_bf16 (128 accumulator registers):
//     6 fragment reads per 8 MFMAs, half of md_linear's LDS traffic per flop; 4 LOADER wavefronts beside them;
//   * both operands go global -> LDS by DMA (buffer_load ... lds, 16 B per lane, no VGPR round trip), three 48-KB
//     stages of 64 k (144 of the 160 KB), the loads of stage s+2 issued right after the one barrier of stage s and
//     retired with a COUNTED vmcnt two stages later -- loads stay in flight across the barrier;
//   * W is read in the streaming layout of md_linear ([N/32][K/16][64 lanes][8]: the MFMA B fragment of 32 columns x
//     16 k is one contiguous KiB): one DMA instruction = one fragment, the LDS image is lane-linear, fragment reads are
//     conflict-free ds_read_b128, and no second packed copy of the weights is needed;
//   * x (row-major, L2-resident, re-read by every column tile) is fetched in full 128-B lines; the LDS image is
//     [row][8 chunks of 16 B] with chunk' = chunk ^ ((row >> 1) & 7): the DMA writes LDS linearly, so the permutation
//     is applied to the per-lane SOURCE address and again on the fragment read -- every 16-lane group of a
//     ds_read_b128 then covers all 64 banks once;
//   * narrow products (N = 4096..6144: 32-48 column tiles) split K over workgroups so that ~240-256 of them exist;
//     fp32 partial tiles go to the workspace and the combine launch applies the epilogue in a fixed slice order
//     (deterministic) -- the SAME combine kernels as md_linear (bias / SwiGLU / residual add + RMSNorm);
//   * the un-split product (w1|w3: 224 tiles) finishes in the kernel: accumulators -> LDS (fp32, the ring is free by
//     then) -> 16-byte row-contiguous stores, SwiGLU with the reference's rounding points.
#include "md_common.h"
#include <type_traits>

// gemm.hip / elementwise.hip: the split-K combine launches shared with md_linear
int md_internal_launch_skinny_reduce(const float* partial, int S, int M, int N, int epilogue, const void* bias, void* out,
                                     int64_t ldo, hipStream_t st);
int md_internal_launch_reduce_add_rmsnorm(const float* partial, int S, int M, int N, const void* bias, const void* scales,
                                          const void* x, int64_t ldx, const void* w, void* h_out, void* y, float eps,
                                          hipStream_t st);

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_ptr_t;

constexpr int BM = 256, BN = 128, BK = 64;
constexpr int kStageA = BM * BK * 2;            // 32 KiB: [256 rows][8 chunks x 16 B], chunk-swizzled
constexpr int kStageB = BN * BK * 2;            // 16 KiB: [4 column tiles][4 k-steps][64 lanes x 16 B]
constexpr int kEpiPitch = BN + 4;               // fp32 epilogue tile [256][132]: 135 168 B inside the rings
constexpr int lds_bytes(int XS, int WS) { return XS * kStageA + WS * kStageB; }

enum { BE_NONE = 0, BE_SWIGLU = 1 };

struct BlockParams {
    const bf16_t* x;
    const bf16_t* w;        // streaming layout [N/32][K/16][64][8]
    const bf16_t* bias;     // [N] or null (un-split BE_NONE only; split products get it in the combine launch)
    bf16_t* out;            // un-split: [M][N] (BE_NONE) or [M][N/2] (BE_SWIGLU), row stride ldo
    float* partial;         // split: [S][M][N] fp32, GEMM-column order ([w1; w3] for BE_SWIGLU)
    int64_t ldx, ldo;
    int M, N, K, S, nsteps;     // nsteps = K / 64
    unsigned int x_bytes;       // bytes of x a row index may address: rows >= M read zeros through the buffer bound
};

__device__ __forceinline__ float silu_bf16(float h1) {
    return bf16_to_f32(f32_to_bf16(h1 / (1.0f + expf(-h1))));     // same expression as md_silu_mul / md_linear
}

// LLVM SchedGroupMask bits for __builtin_amdgcn_sched_group_barrier
#define SG_MFMA 0x8
#define SG_VMEM 0x10
#define SG_DSR 0x100

// WNT: weights with the non-temporal cache policy (the tile owns all 256 rows: a weight byte is read by one workgroup)
//
// Eight wavefronts, two per SIMD with different jobs:
//   waves 0-3  CONSUMERS: fragment reads + MFMAs only, software-pipelined one k-step ahead;
//   waves 4-5  x LOADERS (16 DMA pieces per stage each), waves 6-7  W LOADERS (8 pieces per stage each).
// What bounds it (w1|w3 at M = 256, timing experiments of GPU calls 1-9, profiles/r04_block_*; the experiment switches
// are gone from the code): the MFMA + LDS skeleton alone runs 40.3 us (1.49 PFLOP/s), the loaders alone 59.5 us -- x only
// (L2 hits) 30.5 = DMA-issue-bound at ~56 cycles per piece, W only (HBM) 47.1 = 5.0 TB/s, and the two TOGETHER 59.5:
// a CU ingests ~50 GB/s once HBM misses are in the mix, whoever issues the DMA (the four MFMA waves themselves: 70.7 us;
// dedicated loader waves: 70.5), however deep the W queue (3, 4 or 6 stages; 4 stages in the loaders' registers), and
// wherever x comes from (one hot 32 KB every stage: 55).  Per-CU ingest for this tile is 3 MB (2 MB of it x, re-read by
// every column tile), so ~60 us is this decomposition's floor at N = 28672 and the kernel sits at 67-72; PMC: HBM
// fetch = 1.05 x the weight bytes, L2 hit rate of x 100 %, LDS conflicts 2 %, matrix pipe 46 % busy.
// One s_barrier per stage joins everybody: a loader arrives when ITS pieces of stage it+1 have landed, a consumer when
// its last fragment read of stage it has returned -- past the barrier stage it+1 is readable and the buffers of stage it
// are free for x stage it+XS / W stage it+WS.
template <int EPI, bool SPLIT, bool WNT, int XS, int WS>
__global__ __launch_bounds__(512) void block_gemm_kernel(const BlockParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];     // the ONLY LDS object (cdna guide 5.4(a))
    constexpr int kWRing = XS * kStageA;                                    // LDS offset of the W ring
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // block id -> (column tile, K slice); with S = 8 a slice (= the x columns it reads) stays on one XCD's L2
    const int slice = blockIdx.x % p.S, tn = blockIdx.x / p.S;
    const int s0 = (int)((int64_t)slice * p.nsteps / p.S), s1 = (int)((int64_t)(slice + 1) * p.nsteps / p.S);
    const int n = s1 - s0;
    const int wm = (wave >> 1) & 1, wn = wave & 1;          // consumers: 2 (M) x 2 (N)
    const int j = lane & 31, kh = lane >> 5;

    float* tile = reinterpret_cast<float*>(lds);            // fp32 epilogue tile (the rings are free by then)

    // A stage past the end of the slice is issued all the same with its offsets pushed beyond the descriptor's bound
    // (bit 31): zero fill, no memory traffic, and every vmcnt is the same count.  (The instruction's immediate offset
    // would move the LDS address too: not used.)
    if (wave >= 6) {
        // ------------------------------------------------------------------ W loader: column tiles 2*lw, 2*lw+1
        const int lw = wave - 6;
        const int64_t wtile = (int64_t)(p.K >> 4) * 512;                   // elements of one 32-column tile
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<bf16_t*>(p.w + (int64_t)tn * 4 * wtile), 0, (unsigned int)(4 * wtile * 2), 0x00020000);
        // one piece = one MFMA B fragment (32 columns x 16 k = one contiguous KiB of the streaming layout)
        const unsigned int bvo0 = (unsigned int)(2 * lw) * (unsigned int)(wtile * 2) + (unsigned int)lane * 16u;
        const unsigned int bvo1 = bvo0 + (unsigned int)(wtile * 2);
        auto issue = [&](int it, int slot) {
            const unsigned int oob = it < n ? 0u : 0x80000000u;
            const int so = (s0 + it) * 4096;                  // byte offset of the stage's 4 fragments in a column tile
            unsigned char* dst = lds + kWRing + slot * kStageB + lw * 8192;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t*)(dst + q * 1024), 16,
                                                         ((q >> 2) ? bvo1 : bvo0) | oob, so + (q & 3) * 1024, 0,
                                                         WNT ? 2 : 0);
        };
#pragma unroll
        for (int i = 0; i < WS; ++i) issue(i, i);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((WS - 1) * 8) : "memory");      // stage 0 has landed
        __builtin_amdgcn_s_barrier();
        int slot = 0;
        for (int it = 0; it < n; ++it) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((WS - 2) * 8) : "memory");  // stage it+1 has landed
            __builtin_amdgcn_s_barrier();                                         // the consumers are done with stage it
            issue(it + WS, slot);
            slot = slot + 1 == WS ? 0 : slot + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // zero-fill pieces of the stages past the end
        __builtin_amdgcn_s_barrier();                           // every wave is done with the rings
    } else if (wave >= 4) {
        // ------------------------------------------------------------------ x loader: rows lw*128 .. +127
        const int lw = wave - 4;
        const __amdgpu_buffer_rsrc_t rx =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x), 0, p.x_bytes, 0x00020000);
        // piece q covers rows (lw*16+q)*8 .. +7, 8 lanes per 128-B line; LDS image [row][8 chunks], chunk c of row r
        // stored at chunk c ^ ((r >> 1) & 7): the DMA writes LDS linearly (base + lane * 16), so the lane FETCHES the
        // chunk that belongs at its slot.  Rows >= M lie beyond the descriptor's bound: zeros.
        unsigned int avo[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int r = (lw * 16 + q) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            avo[q] = (unsigned int)r * (unsigned int)p.ldx * 2u + (unsigned int)c * 16u;
        }
        auto issue = [&](int it, int slot) {
            const unsigned int oob = it < n ? 0u : 0x80000000u;
            const int so = (s0 + it) * (BK * 2);              // byte offset of the stage's k range in a row of x
            unsigned char* dst = lds + slot * kStageA + lw * 16384;
#pragma unroll
            for (int q = 0; q < 16; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t*)(dst + q * 1024), 16, avo[q] | oob, so, 0, 0);
        };
#pragma unroll
        for (int i = 0; i < XS; ++i) issue(i, i);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((XS - 1) * 16) : "memory");
        __builtin_amdgcn_s_barrier();
        int slot = 0;
        for (int it = 0; it < n; ++it) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((XS - 2) * 16) : "memory");
            __builtin_amdgcn_s_barrier();
            issue(it + XS, slot);
            slot = slot + 1 == XS ? 0 : slot + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    } else {
        // ------------------------------------------------------------------ consumer
        const int f = (j >> 1) & 7;
        const unsigned char* a_lane = lds + (wm * 128 + j) * 128;
        int axo[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) axo[ks] = ((2 * ks + kh) ^ f) * 16;
        const unsigned char* b_lane = lds + kWRing + wn * 8192 + lane * 16;
        f32x16 acc[4][2];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

        bf16x8 fa[2][4], fb[2][2];                             // fragment double buffer (compile-time indices only)
        auto read_frags = [&](int set, int xoff, int woff, int ks) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                fa[set][mt] = *reinterpret_cast<const bf16x8*>(a_lane + xoff + axo[ks] + mt * 4096);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                fb[set][nt] = *reinterpret_cast<const bf16x8*>(b_lane + woff + (nt * 4 + ks) * 1024);
        };
        auto mfmas = [&](int set) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[set][mt], fb[set][nt], acc[mt][nt], 0, 0, 0);
        };
        // the issue order of one k-step: 8 MFMAs (fragments read one k-step ago) with the 6 fragment reads of the NEXT
        // k-step in the shadows of the first six
        auto interleave = [&]() {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
                __builtin_amdgcn_sched_group_barrier(SG_DSR, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, 2, 0);
        };
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        read_frags(0, 0, 0, 0);
        int xs = 0, ws = 0;
        for (int it = 0; it < n; ++it) {
            const int xoff = xs * kStageA, woff = ws * kStageB;
            xs = xs + 1 == XS ? 0 : xs + 1;
            ws = ws + 1 == WS ? 0 : ws + 1;
            read_frags(1, xoff, woff, 1);
            mfmas(0);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            read_frags(0, xoff, woff, 2);
            mfmas(1);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            read_frags(1, xoff, woff, 3);
            mfmas(0);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my last reads of this stage have returned
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            read_frags(0, xs * kStageA, ws * kStageB, 0);
            mfmas(1);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- accumulators -> fp32 tile in LDS
        // acc[mt][nt][r] = D[row wm*128 + mt*32 + (r&3) + 8*(r>>2) + 4*kh][column wn*64 + nt*32 + j]
        __builtin_amdgcn_s_barrier();                           // every wave is done with the rings (no DMA pending)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * 128 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    tile[row * kEpiPitch + wn * 64 + nt * 32 + j] = acc[mt][nt][r];
                }
    }
    // ---- epilogue: row-contiguous 16-byte stores by all 512 threads
    __syncthreads();

    if constexpr (SPLIT) {
        float* pp = p.partial + (int64_t)slice * p.M * p.N;
#pragma unroll 4
        for (int q = 0; q < 16; ++q) {
            const int c = tid + 512 * q;
            const int row = c >> 5, c4 = (c & 31) * 4;                        // 4 consecutive tile columns
            if (row >= p.M) continue;
            const f32x4 v = *reinterpret_cast<const f32x4*>(tile + row * kEpiPitch + c4);
            int col;
            if constexpr (EPI == BE_SWIGLU) {                                 // packed tile: 16 rows of w1, 16 of w3
                const int t = tn * 4 + (c4 >> 5), jj = c4 & 31;
                col = jj < 16 ? t * 16 + jj : (p.N >> 1) + t * 16 + jj - 16;
            } else {
                col = tn * BN + c4;
            }
            *reinterpret_cast<f32x4*>(pp + (int64_t)row * p.N + col) = v;
        }
    } else if constexpr (EPI == BE_SWIGLU) {
#pragma unroll 4
        for (int q = 0; q < 4; ++q) {
            const int c = tid + 512 * q;
            const int row = c >> 3, t = (c >> 1) & 3, i8 = (c & 1) * 8;       // 8 outputs of packed tile t
            if (row >= p.M) continue;
            const float* src = tile + row * kEpiPitch + t * 32 + i8;
            const f32x4 h1a = *reinterpret_cast<const f32x4*>(src), h1b = *reinterpret_cast<const f32x4*>(src + 4);
            const f32x4 h3a = *reinterpret_cast<const f32x4*>(src + 16), h3b = *reinterpret_cast<const f32x4*>(src + 20);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = f32_to_bf16(silu_bf16(bf16_to_f32(f32_to_bf16(h1a[e]))) * bf16_to_f32(f32_to_bf16(h3a[e])));
                o[4 + e] = f32_to_bf16(silu_bf16(bf16_to_f32(f32_to_bf16(h1b[e]))) * bf16_to_f32(f32_to_bf16(h3b[e])));
            }
            *reinterpret_cast<bf16x8*>(p.out + (int64_t)row * p.ldo + (tn * 4 + t) * 16 + i8) = o;
        }
    } else {
        const int c8 = (tid & 15) * 8;                                        // the thread's 8 columns: the same for every q
        f32x8 bv = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (p.bias) bv = __builtin_convertvector(*reinterpret_cast<const bf16x8*>(p.bias + tn * BN + c8), f32x8);
#pragma unroll 4
        for (int q = 0; q < 8; ++q) {
            const int row = (tid + 512 * q) >> 4;
            if (row >= p.M) continue;
            const float* src = tile + row * kEpiPitch + c8;
            const f32x4 va = *reinterpret_cast<const f32x4*>(src), vb = *reinterpret_cast<const f32x4*>(src + 4);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = f32_to_bf16(va[e] + bv[e]);
                o[4 + e] = f32_to_bf16(vb[e] + bv[4 + e]);
            }
            *reinterpret_cast<bf16x8*>(p.out + (int64_t)row * p.ldo + tn * BN + c8) = o;
        }
    }
}

int g_target_blocks = 256;     // K is split so that about this many workgroups exist (md_debug_set_block_gemm)
int g_wnt = 1;                 // non-temporal weight DMA

int pick_splits(int n_tiles, int nsteps) {
    if (n_tiles >= 160) return 1;                                   // >= 62 % of the CUs busy without partial sums
    int s = g_target_blocks / n_tiles;
    if (s < 1) s = 1;
    if (s > 16) s = 16;
    if (s > nsteps / 4) s = nsteps / 4 > 0 ? nsteps / 4 : 1;        // at least 4 stages (256 k) per workgroup
    return s;
}

// x ring 3 x 32 KB + W ring 4 x 16 KB = the CU's 160 KB.  Measured alternatives (same process, w1|w3 at M = 256,
// profiles/r04_block_*): W ring 3 deep: equal; x 2 + W 6: 81 vs 70.5 us (one x stage in flight is too few); the loaders'
// register files as a 3-4 stage deep FIFO in front of a two-slot LDS image for BOTH operands (plain loads counted by
// hipcc + ds_write_b128): 76-81 us; for the W stream only, 4-6 stages deep with inline-asm loads and a counted vmcnt
// (96 KB of W in flight per CU instead of 48): 64.1-68.9 against 66.5 -- noise (r04_block_wreg_fifo_rejected_call28.txt).
template <int EPI, bool SPLIT, bool WNT>
int launch_cfg(const BlockParams& p, int grid, hipStream_t st) {
    constexpr int XS = 3, WS = 4;
    constexpr int kLds = lds_bytes(XS, WS);
    static_assert(BM * kEpiPitch * 4 <= kLds && kLds <= 160 * 1024, "rings must hold the epilogue tile and fit the CU");
    auto k = block_gemm_kernel<EPI, SPLIT, WNT, XS, WS>;
    static MdPerDeviceOnce once;
    if (once.first()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, kLds) !=
            hipSuccess) {
            once.undo();
            md_set_error("md_linear_block: hipFuncSetAttribute(%d B LDS) failed", kLds);
            return MD_ERR_LAUNCH;
        }
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), kLds, st, p);
    return MD_OK;
}

template <int EPI>
int launch_epi(const BlockParams& p, int grid, hipStream_t st) {
    if (p.S > 1 || p.partial)
        return g_wnt ? launch_cfg<EPI, true, true>(p, grid, st) : launch_cfg<EPI, true, false>(p, grid, st);
    return g_wnt ? launch_cfg<EPI, false, true>(p, grid, st) : launch_cfg<EPI, false, false>(p, grid, st);
}

bool shape_ok(int M, int N, int K) { return M >= 1 && M <= BM && N >= BN && N % BN == 0 && K >= BK && K % BK == 0; }

// fills the GEMM part of the parameters; `force_split`: partial sums even with one K slice (the combine launch applies an
// epilogue the kernel does not have)
int setup(BlockParams& p, const void* x, int64_t ldx, const void* w, int M, int N, int K, bool force_split,
          void* workspace, size_t workspace_bytes, const char* who) {
    MD_CHECK_ARG(x && w, "%s: null pointer argument", who);
    MD_CHECK_ARG(shape_ok(M, N, K), "%s: unsupported shape M=%d N=%d K=%d (need 1 <= M <= 256, N %% 128 == 0, K %% 64 == 0)",
                 who, M, N, K);
    MD_CHECK_ARG((((uintptr_t)x | (uintptr_t)w) & 15) == 0 && ldx % 8 == 0 && ldx >= K,
                 "%s: x / w must be 16-byte aligned, ldx %% 8 == 0", who);
    const int64_t xb = ((int64_t)(M - 1) * ldx + K) * 2;
    MD_CHECK_ARG(xb < ((int64_t)1 << 31) && (int64_t)255 * ldx * 2 + K * 2 < ((int64_t)1 << 31) && (int64_t)K * 256 < ((int64_t)1 << 31),
                 "%s: x spans more than 2 GiB (M=%d, ldx=%lld): pass a compact activation tensor", who, M, (long long)ldx);
    p.x = (const bf16_t*)x;
    p.w = (const bf16_t*)w;
    p.bias = nullptr;
    p.out = nullptr;
    p.ldx = ldx;
    p.ldo = 0;
    p.M = M;
    p.N = N;
    p.K = K;
    p.nsteps = K / BK;
    p.S = pick_splits(N / BN, p.nsteps);
    p.x_bytes = (unsigned int)xb;
    p.partial = nullptr;
    if (p.S > 1 || force_split) {
        const size_t need = (size_t)p.S * M * N * 4;
        MD_CHECK_ARG(workspace && workspace_bytes >= need && (((uintptr_t)workspace) & 15) == 0,
                     "%s: workspace too small (need %zu bytes) or not 16-byte aligned", who, need);
        p.partial = (float*)workspace;
    }
    return MD_OK;
}

}  // namespace

#ifdef MD_DEV_KNOBS
extern "C" void md_debug_set_block_gemm(int target_blocks, int weights_nontemporal) {
    g_target_blocks = target_blocks > 0 ? target_blocks : 256;
    g_wnt = weights_nontemporal ? 1 : 0;
}
#endif

extern "C" int md_linear_block_supported(int M, int N, int K, int epilogue) {
    if (!shape_ok(M, N, K)) return 0;
    return (epilogue == BE_NONE || epilogue == BE_SWIGLU) ? 1 : 0;
}

extern "C" size_t md_linear_block_workspace_bytes(int M, int N, int K, int force_split) {
    if (!shape_ok(M, N, K)) return 0;
    const int S = pick_splits(N / BN, K / BK);
    return (S > 1 || force_split) ? (size_t)S * M * N * 4 : 0;
}

extern "C" int md_linear_block(const void* x, int64_t ldx, const void* w_packed, const void* bias, void* out, int64_t ldo,
                               int M, int N, int K, int epilogue, void* workspace, size_t workspace_bytes,
                               md_stream_t stream) {
    MD_CHECK_ARG(out && (((uintptr_t)out) & 15) == 0 && ldo % 8 == 0, "md_linear_block: out must be 16-byte aligned, ldo %% 8 == 0");
    MD_CHECK_ARG(epilogue == BE_NONE || epilogue == BE_SWIGLU, "md_linear_block: unknown epilogue %d", epilogue);
    MD_CHECK_ARG(!(epilogue == BE_SWIGLU && bias), "md_linear_block: the SwiGLU epilogue takes no bias");
    MD_CHECK_ARG(!bias || (((uintptr_t)bias) & 15) == 0, "md_linear_block: bias must be 16-byte aligned (read as bf16x8)");
    BlockParams p;
    int rc = setup(p, x, ldx, w_packed, M, N, K, false, workspace, workspace_bytes, "md_linear_block");
    if (rc != MD_OK) return rc;
    p.out = (bf16_t*)out;
    p.ldo = ldo;
    p.bias = p.S > 1 ? nullptr : (const bf16_t*)bias;
    hipStream_t st = (hipStream_t)stream;
    const int grid = (N / BN) * p.S;
    rc = epilogue == BE_SWIGLU ? launch_epi<BE_SWIGLU>(p, grid, st) : launch_epi<BE_NONE>(p, grid, st);
    if (rc != MD_OK) return rc;
    if (p.S > 1) {
        rc = md_internal_launch_skinny_reduce(p.partial, p.S, M, N, epilogue, bias, out, ldo, st);
        if (rc != MD_OK) return rc;
    }
    MD_CHECK_LAUNCH("md_linear_block");
    return MD_OK;
}

extern "C" int md_linear_block_add_rmsnorm(const void* x, int64_t ldx, const void* w_packed, const void* bias,
                                           const void* resid, int64_t ldr, const void* norm_weight, float eps, void* h_out,
                                           void* y_out, int M, int N, int K, void* workspace, size_t workspace_bytes,
                                           md_stream_t stream) {
    MD_CHECK_ARG(resid && norm_weight && h_out && y_out, "md_linear_block_add_rmsnorm: null pointer argument");
    MD_CHECK_ARG(!bias || (((uintptr_t)bias) & 15) == 0, "md_linear_block_add_rmsnorm: bias must be 16-byte aligned");
    MD_CHECK_ARG((((uintptr_t)resid | (uintptr_t)norm_weight | (uintptr_t)h_out | (uintptr_t)y_out) & 15) == 0 &&
                     ldr % 8 == 0 && N % 8 == 0 && N <= 8192,
                 "md_linear_block_add_rmsnorm: pointers must be 16-byte aligned, ldr %% 8 == 0, N %% 8 == 0, N <= 8192");
    BlockParams p;
    int rc = setup(p, x, ldx, w_packed, M, N, K, true, workspace, workspace_bytes, "md_linear_block_add_rmsnorm");
    if (rc != MD_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    rc = launch_epi<BE_NONE>(p, (N / BN) * p.S, st);
    if (rc != MD_OK) return rc;
    rc = md_internal_launch_reduce_add_rmsnorm(p.partial, p.S, M, N, bias, nullptr, resid, ldr, norm_weight, h_out, y_out,
                                               eps, st);
    if (rc != MD_OK) return rc;
    MD_CHECK_LAUNCH("md_linear_block_add_rmsnorm");
    return MD_OK;
}
