// SnapKV select (K6, row a8 of SURVEY.md section 8) for gfx950.
//
// reference: Attention.gen_draft_kv  Engine/SnapKV/model.py:389-439
// restated (bit-exact vs the reference on CPU) in oracle/magicdec_ref.py:snapkv_scores.
//
// Rounding sequence reproduced (g = H/KH, W = window, S = ctx_len, L = g*W rows
// per kv head ordered (r,l)):
//   for each chunk of 8g rows:  s = bf16(q.k)  (UNSCALED);  the last W rows x last
//   W columns of the chunk get the WxW causal mask;  p = bf16(softmax_fp32(s));
//   rows regrouped (r'=g, l'=8): gs = bf16(sum_l' p[..., :S-W]);  acc[r'] = bf16(acc[r'] + gs)
//   pooled = bf16(avg_pool1d(acc, k, pad k/2, /k));  score = bf16(sum_r' pooled)
//   indices = top-(budget-W) of score, descending score, ties -> lowest index
//   draft rows = K,V[indices] ++ K,V[S-W:S]
//
// Four small launches, none on the timed decode path (once per prefill per layer):
//   stats   : row max / sum-exp partials per 1024-column chunk   (MFMA scores; float64 sums)
//   accum   : recompute scores, p, 8-row group sums, bf16 chunk accumulation
//   select  : pool + group sum + exact radix select + bitonic sort (one WG per b,kvh)
//   gather  : copy the selected rows into the draft pages
#include "md_common.h"

namespace {

constexpr int kChunkCols = 1024;

struct SnapParams {
    const bf16_t* q;      // [B*W, H, D]
    const void* cache;    // paged full KV (bf16, or e4m3fn bytes with per-head scales)
    const float* k_scale;
    const int32_t* page_indices;
    const int32_t* page_indptr;
    double* partials;       // [B*KH][L][nch][2] = (row max, sum of exp) per 1024-column chunk, float64
    unsigned short* aws;    // [B][H][S-W] bf16 bits
    int B, H, KH, g, W, S, L, nch, page_size;
    int64_t page_stride;
    int slot_stride;      // elements between rows of a page: KH*D (NHD) or D (HND)
    int head_stride;      // elements between kv heads: D (NHD) or page_size*D (HND)
};

// K fragment (A operand) of 16 consecutive columns (kv positions) col0..col0+15 for head kvh:
// lane (lq,lc) holds K[col0+lq][ks*32 + lc*8 .. +8]
template <int D, bool FP8>
__device__ __forceinline__ void load_kfrag(const SnapParams& p, int b, int kvh, int col0, int lq, int lc,
                                           bf16x8 (&kf)[D / 32]) {
    const int col = col0 + lq;
    const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    if (col < p.S) {
        const int page = col / p.page_size, slot = col - page * p.page_size;
        const int64_t pid = p.page_indices[p.page_indptr[b] + page];
        const int64_t off = pid * p.page_stride + (int64_t)slot * p.slot_stride + (int64_t)kvh * p.head_stride + lc * 8;   // elements
        if constexpr (FP8) {
            const unsigned char* kp = reinterpret_cast<const unsigned char*>(p.cache) + off;
#pragma unroll
            for (int ks = 0; ks < D / 32; ++ks) {
                const u32x2 w = *reinterpret_cast<const u32x2*>(kp + ks * 32);   // 8 e4m3fn bytes -> 8 bf16 (exact)
                bf16x8 r;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const bf16x2 a = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w[h], 1.0f, false);
                    const bf16x2 c = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w[h], 1.0f, true);
                    r[4 * h] = a[0];
                    r[4 * h + 1] = a[1];
                    r[4 * h + 2] = c[0];
                    r[4 * h + 3] = c[1];
                }
                kf[ks] = r;
            }
        } else {
            const bf16_t* kp = reinterpret_cast<const bf16_t*>(p.cache) + off;
#pragma unroll
            for (int ks = 0; ks < D / 32; ++ks) kf[ks] = *reinterpret_cast<const bf16x8*>(kp + ks * 32);
        }
    } else {
#pragma unroll
        for (int ks = 0; ks < D / 32; ++ks) kf[ks] = z;
    }
}

// scores of row tile rt (16 rows) x 16 columns: returns bf16-rounded, masked scores for
// (row rt*16+lq, col col0+lc*4+j); masked / out-of-range entries are -inf
template <int D>
__device__ __forceinline__ f32x4 score_tile(const SnapParams& p, int b, int kvh, int rt, int col0, int lq, int lc,
                                            const bf16x8 (&kf)[D / 32], float kscale) {
    const int rg = rt * 16 + lq;  // row within the kv head, ordered (r,l)
    const int r = rg / p.W, l = rg - r * p.W;
    const bf16_t* qp = p.q + ((int64_t)(b * p.W + l) * p.H + kvh * p.g + r) * D + lc * 8;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < D / 32; ++ks) {
        const bf16x8 qf = *reinterpret_cast<const bf16x8*>(qp + ks * 32);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[ks], qf, acc, 0, 0, 0);
    }
    const int chunk_rows = 8 * p.g;
    const int rho = rg % chunk_rows;
    const int mrow = rho - (chunk_rows - p.W);  // row of the WxW mask (valid if >= 0)
    f32x4 s;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = col0 + lc * 4 + j;
        float v = bf16_to_f32(f32_to_bf16(acc[j] * kscale));      // kscale = 1 for a bf16 cache
        const int jcol = col - (p.S - p.W);
        if (col >= p.S || (mrow >= 0 && jcol > mrow)) v = -INFINITY;
        s[j] = v;
    }
    return s;
}

template <int D, bool FP8>
__global__ __launch_bounds__(256) void snapkv_stats_kernel(const SnapParams p) {
    // The softmax denominator and the normalised probabilities are evaluated in float64 so that p = bf16(softmax)
    // is the CORRECTLY ROUNDED value: the reference's fp32 CPU softmax agrees with the correctly rounded one on every
    // element of the fixtures (tests/test_gpu_ops.py measures both), an fp32 GPU evaluation with another sum order
    // does not (a ~1e-7 relative error of Z flips ~1e-4 of the bf16 roundings, which the 8-row sums, the pooling and
    // the group sum then spread over ~1 % of the final scores).  Runs once per prefill: the float64 cost is nil.
    extern __shared__ __attribute__((aligned(16))) double smd[];  // [4 waves][L][2]
    const int chunk = blockIdx.x, kvh = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq = lane & 15, lc = lane >> 4;
    double* ms = smd + wave * p.L * 2;
    for (int i = lane; i < p.L; i += 64) {
        ms[i * 2] = -INFINITY;
        ms[i * 2 + 1] = 0.0;
    }
    const int RT = p.L / 16;
    for (int cg = wave; cg < kChunkCols / 16; cg += 4) {
        const int col0 = chunk * kChunkCols + cg * 16;
        if (col0 >= p.S) break;
        bf16x8 kf[D / 32];
        load_kfrag<D, FP8>(p, b, kvh, col0, lq, lc, kf);
        const float kscale = FP8 ? p.k_scale[kvh] : 1.0f;
        for (int rt = 0; rt < RT; ++rt) {
            const f32x4 s = score_tile<D>(p, b, kvh, rt, col0, lq, lc, kf, kscale);
            float mx = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            double sum = 0.0;
            if (mx > -INFINITY) {
#pragma unroll
                for (int j = 0; j < 4; ++j) sum += exp((double)s[j] - (double)mx);
            }
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            if (lc == 0 && mx > -INFINITY) {
                const int row = rt * 16 + lq;
                const double mo = ms[row * 2], zo = ms[row * 2 + 1];
                const double mn = fmax(mo, (double)mx);
                ms[row * 2] = mn;
                ms[row * 2 + 1] = (mo > -INFINITY ? zo * exp(mo - mn) : 0.0) + sum * exp((double)mx - mn);
            }
        }
    }
    __syncthreads();
    for (int row = tid; row < p.L; row += 256) {
        double M = -INFINITY;
        for (int w = 0; w < 4; ++w) M = fmax(M, smd[(w * p.L + row) * 2]);
        double Z = 0.0;
        if (M > -INFINITY)
            for (int w = 0; w < 4; ++w) {
                const double mw = smd[(w * p.L + row) * 2];
                if (mw > -INFINITY) Z += smd[(w * p.L + row) * 2 + 1] * exp(mw - M);
            }
        double* out = p.partials + (((int64_t)(b * p.KH + kvh) * p.L + row) * p.nch + chunk) * 2;
        out[0] = M;
        out[1] = Z;
    }
}

template <int D, bool FP8>
__global__ __launch_bounds__(256) void snapkv_accum_kernel(const SnapParams p) {
    extern __shared__ __attribute__((aligned(16))) double smd[];  // [L][2] row stats (f64), then [4][g][16] f32 accumulators
    const int ctile = blockIdx.x, kvh = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq = lane & 15, lc = lane >> 4;
    double* MZ = smd;
    float* accs = reinterpret_cast<float*>(smd + p.L * 2) + wave * p.g * 16;
    for (int row = tid; row < p.L; row += 256) {
        const double* pp = p.partials + ((int64_t)(b * p.KH + kvh) * p.L + row) * p.nch * 2;
        double M = -INFINITY;
        for (int c = 0; c < p.nch; ++c) M = fmax(M, pp[c * 2]);
        double Z = 0.0;
        for (int c = 0; c < p.nch; ++c)
            if (pp[c * 2] > -INFINITY) Z += pp[c * 2 + 1] * exp(pp[c * 2] - M);
        MZ[row * 2] = M;
        MZ[row * 2 + 1] = 1.0 / Z;
    }
    for (int i = lane; i < p.g * 16; i += 64) accs[i] = 0.f;
    __syncthreads();
    const int N = p.S - p.W;
    const int col0 = ctile * 64 + wave * 16;
    if (col0 >= N) return;
    bf16x8 kf[D / 32];
    load_kfrag<D, FP8>(p, b, kvh, col0, lq, lc, kf);
    const float kscale = FP8 ? p.k_scale[kvh] : 1.0f;
    const int RT = p.L / 16;
    for (int rt = 0; rt < RT; ++rt) {
        const f32x4 s = score_tile<D>(p, b, kvh, rt, col0, lq, lc, kf, kscale);
        const int row = rt * 16 + lq;
        const double M = MZ[row * 2], rZ = MZ[row * 2 + 1];
        f32x4 gs;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // softmax -> bf16 (model.py:416; float64 here, see the stats kernel), then the 8-row group sum in
            // fp32 -> bf16 (:418)
            float pj = (float)(exp((double)s[j] - M) * rZ);
            pj = bf16_to_f32(f32_to_bf16(pj));
            pj += __shfl_xor(pj, 1);
            pj += __shfl_xor(pj, 2);
            pj += __shfl_xor(pj, 4);
            gs[j] = bf16_to_f32(f32_to_bf16(pj));
        }
        if ((lq & 7) == 0) {
            const int G = rt * 2 + (lq >> 3);  // global 8-row group; chunk = G / g, pseudo head r' = G % g
            const int rp = G % p.g;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float* a = accs + rp * 16 + lc * 4 + j;
                *a = bf16_to_f32(f32_to_bf16(*a + gs[j]));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // write bf16 accumulators: aws[b][kvh*g + r'][col]
    for (int i = lane; i < p.g * 16; i += 64) {
        const int rp = i / 16, c = i % 16;
        const int col = col0 + c;
        if (col < N) {
            const bf16_t v = f32_to_bf16(accs[i]);
            p.aws[((int64_t)b * p.H + kvh * p.g + rp) * N + col] = *reinterpret_cast<const unsigned short*>(&v);
        }
    }
}

__device__ __forceinline__ unsigned int ordered16(unsigned short bits) {
    return (bits & 0x8000u) ? (unsigned int)(~bits & 0xffffu) : (unsigned int)(bits | 0x8000u);
}

// one workgroup (1024 threads) per (kv head, request)
__global__ __launch_bounds__(1024) void snapkv_select_kernel(const unsigned short* __restrict__ aws, int H, int KH, int g,
                                                             int N, int ksz, int topk, int32_t* idx_out,
                                                             unsigned short* score_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    unsigned short* keys = reinterpret_cast<unsigned short*>(smraw);  // [N] ordered keys
    const int npad_keys = (N + 7) / 8 * 8;
    unsigned int* hist = reinterpret_cast<unsigned int*>(smraw + npad_keys * 2);  // [256]
    unsigned int* sel = hist + 256;                                               // [kpad]
    unsigned int* misc = sel + 1024;                                              // scalars / wave counts
    const int kvh = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x;
    const int half = ksz / 2;

    // 1. pooled + group-summed score per column -> ordered 16-bit key
    for (int s = tid; s < N; s += 1024) {
        float tot = 0.f;
        for (int rp = 0; rp < g; ++rp) {
            const unsigned short* row = aws + ((int64_t)b * H + kvh * g + rp) * N;
            float acc = 0.f;
            for (int k = -half; k <= half; ++k) {
                const int c = s + k;
                if (c >= 0 && c < N) acc += bf16_bits_to_f32(row[c]);
            }
            const float pooled = bf16_to_f32(f32_to_bf16(acc / (float)ksz));
            tot += pooled;
        }
        const bf16_t tb = f32_to_bf16(tot);
        const unsigned short bits = *reinterpret_cast<const unsigned short*>(&tb);
        keys[s] = (unsigned short)ordered16(bits);
        if (score_out) score_out[((int64_t)b * KH + kvh) * N + s] = bits;
    }
    // 2. exact k-th largest key by two 8-bit histogram rounds
    unsigned int prefix_hi = 0, kth_rem = topk;
    for (int round = 0; round < 2; ++round) {
        for (int i = tid; i < 256; i += 1024) hist[i] = 0;
        __syncthreads();
        for (int s = tid; s < N; s += 1024) {
            const unsigned int k = keys[s];
            if (round == 0)
                atomicAdd(&hist[k >> 8], 1u);
            else if ((k >> 8) == prefix_hi)
                atomicAdd(&hist[k & 0xff], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned int rem = kth_rem;
            int bin = 255;
            for (; bin > 0; --bin) {
                if (hist[bin] >= rem) break;
                rem -= hist[bin];
            }
            misc[0] = bin;
            misc[1] = rem;  // rank of the k-th inside this bin (1-based from the top)
        }
        __syncthreads();
        if (round == 0) prefix_hi = misc[0];
        kth_rem = misc[1];
        __syncthreads();
    }
    const unsigned int T = (prefix_hi << 8) | misc[0];
    const unsigned int need_eq = kth_rem;  // how many keys == T are selected (lowest indices first)
    __syncthreads();
    // 3. compaction: > T unordered (sorted later); == T in index order
    if (tid == 0) {
        misc[2] = 0;  // count of > T written
        misc[3] = 0;  // running count of == T
    }
    for (int i = tid; i < 1024; i += 1024) sel[i] = 0;
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    unsigned int* wcnt = misc + 8;  // [16]
    for (int base = 0; base < N; base += 1024) {
        const int s = base + tid;
        const unsigned int k = s < N ? keys[s] : 0u;
        const bool gt = s < N && k > T;
        const bool eq = s < N && k == T;
        if (gt) {
            const unsigned int slot = atomicAdd(&misc[2], 1u);
            sel[slot] = (k << 16) | (0xffffu - (unsigned int)s);
        }
        const unsigned long long bal = __ballot(eq);
        const unsigned int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wcnt[wave] = __popcll(bal);
        __syncthreads();
        unsigned int woff = misc[3];
        for (int w = 0; w < wave; ++w) woff += wcnt[w];
        if (eq) {
            const unsigned int rank = woff + before;
            if (rank < need_eq) sel[(topk - need_eq) + rank] = (k << 16) | (0xffffu - (unsigned int)s);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned int t = misc[3];
            for (int w = 0; w < 16; ++w) t += wcnt[w];
            misc[3] = t;
        }
        __syncthreads();
    }
    // 4. bitonic sort (descending) of the padded selection
    int kpad = 1;
    while (kpad < topk) kpad <<= 1;
    for (int size = 2; size <= kpad; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < kpad / 2; i += 1024) {
                const int lo = (i / stride) * stride * 2 + (i % stride);
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const unsigned int a = sel[lo], c = sel[hi];
                if ((a < c) == desc) {
                    sel[lo] = c;
                    sel[hi] = a;
                }
            }
            __syncthreads();
        }
    }
    for (int j = tid; j < topk; j += 1024)
        idx_out[((int64_t)b * KH + kvh) * topk + j] = (int32_t)(0xffffu - (sel[j] & 0xffffu));
}

struct GatherParams {
    const void* cache;
    const float* k_scale;
    const float* v_scale;
    const int32_t* page_indices;
    const int32_t* page_indptr;
    bf16_t* dcache;
    const int32_t* dindices;
    const int32_t* dindptr;
    const int32_t* dlast;
    const int32_t* idx;  // [B][KH][topk]
    int KH, D, page_size, S, W, budget, topk;
    int src_hnd;         // source pages are [2][KH][page_size][D] instead of [2][page_size][KH][D]
};

template <bool FP8>
__global__ __launch_bounds__(64) void snapkv_gather_kernel(const GatherParams p) {
    const int j = blockIdx.x, kvh = blockIdx.y, b = blockIdx.z;
    const int src = j < p.topk ? p.idx[((int64_t)b * p.KH + kvh) * p.topk + j] : p.S - p.W + (j - p.topk);
    const int row_elems = p.KH * p.D;
    const int sp = src / p.page_size, ss = src - sp * p.page_size;
    const int64_t spid = p.page_indices[p.page_indptr[b] + sp];
    const int64_t soff = p.src_hnd ? spid * 2 * p.page_size * row_elems + ((int64_t)kvh * p.page_size + ss) * p.D
                                   : (spid * 2 * p.page_size + ss) * row_elems + kvh * p.D;
    const int dp0 = p.dindptr[b];
    const int dnp = p.dindptr[b + 1] - dp0;
    const int dlen = dnp > 0 ? (dnp - 1) * p.page_size + p.dlast[b] : 0;
    const int pos = dlen - p.budget + j;
    if (pos < 0 || pos >= dnp * p.page_size) return;
    const int dpg = pos / p.page_size, dsl = pos - dpg * p.page_size;
    const int64_t dpid = p.dindices[dp0 + dpg];
    const int64_t doff = (dpid * 2 * p.page_size + dsl) * row_elems + kvh * p.D;
    const int64_t half = (int64_t)p.page_size * row_elems;
    const int nv = p.D / 8;
    for (int i = threadIdx.x; i < 2 * nv; i += 64) {
        const int64_t h = i < nv ? 0 : half;
        const int c = i < nv ? i : i - nv;
        if constexpr (FP8) {   // dequantise into the bf16 draft cache: bf16(byte * scale)
            const float sc = i < nv ? p.k_scale[kvh] : p.v_scale[kvh];
            const u32x2 w = *reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned char*>(p.cache) + soff + h + c * 8);
            bf16x8 r;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8(w[hh], false);
                const f32x2 cc = __builtin_amdgcn_cvt_pk_f32_fp8(w[hh], true);
                r[4 * hh] = f32_to_bf16(a[0] * sc);
                r[4 * hh + 1] = f32_to_bf16(a[1] * sc);
                r[4 * hh + 2] = f32_to_bf16(cc[0] * sc);
                r[4 * hh + 3] = f32_to_bf16(cc[1] * sc);
            }
            *reinterpret_cast<bf16x8*>(p.dcache + doff + h + c * 8) = r;
        } else {
            *reinterpret_cast<u32x4*>(p.dcache + doff + h + c * 8) =
                *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(p.cache) + soff + h + c * 8);
        }
    }
}

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace

extern "C" size_t md_snapkv_workspace_bytes(int B, int H, int KH, int ctx_len, int window) {
    if (B <= 0 || H <= 0 || KH <= 0 || H % KH || ctx_len <= window) return 0;
    const int g = H / KH;
    const size_t L = (size_t)g * window;
    const size_t nch = (ctx_len + kChunkCols - 1) / kChunkCols;
    const size_t partials = align_up((size_t)B * KH * L * nch * 2 * 8, 256);
    const size_t aws = align_up((size_t)B * H * (ctx_len - window) * 2, 256);
    const size_t scores = align_up((size_t)B * KH * (ctx_len - window) * 2, 256);
    return partials + aws + scores + 256;
}

extern "C" size_t md_snapkv_scores_offset(int B, int H, int KH, int ctx_len, int window) {
    if (B <= 0 || H <= 0 || KH <= 0 || H % KH || ctx_len <= window) return 0;
    const int g = H / KH;
    const size_t L = (size_t)g * window;
    const size_t nch = (ctx_len + kChunkCols - 1) / kChunkCols;
    return align_up((size_t)B * KH * L * nch * 2 * 8, 256) + align_up((size_t)B * H * (ctx_len - window) * 2, 256);
}

extern "C" int md_snapkv_select(const void* q_win, const void* cache, const int32_t* page_indices,
                                const int32_t* page_indptr, int B, int H, int KH, int D, int page_size, int ctx_len,
                                int window, int budget, int pool_kernel, void* draft_cache,
                                const int32_t* draft_page_indices, const int32_t* draft_page_indptr,
                                const int32_t* draft_last_page_len, int32_t* idx_out, int kv_dtype,
                                const float* k_scale, const float* v_scale, void* workspace,
                                size_t workspace_bytes, md_stream_t stream) {
    MD_CHECK_ARG(q_win && cache && page_indices && page_indptr && draft_cache && draft_page_indices &&
                     draft_page_indptr && draft_last_page_len && idx_out && workspace,
                 "md_snapkv_select: null pointer argument");
    MD_CHECK_ARG(B > 0 && H > 0 && KH > 0 && H % KH == 0, "md_snapkv_select: bad heads H=%d KH=%d", H, KH);
    if (D != 64 && D != 128) {
        md_set_error("md_snapkv_select: head_dim %d unsupported (64 or 128)", D);
        return MD_ERR_UNSUPPORTED;
    }
    const int g = H / KH;
    const int L = g * window;
    const int topk = budget - window;
    const int N = ctx_len - window;
    // the reference itself raises for 8g < window (model.py:415: 8g-row chunk vs WxW mask)
    MD_CHECK_ARG(8 * g >= window, "md_snapkv_select: needs 8*(H/KH) >= window (reference raises for g=%d)", g);
    MD_CHECK_ARG(window > 0 && L % 16 == 0 && L % (8 * g) == 0, "md_snapkv_select: window %d unsupported", window);
    MD_CHECK_ARG(topk > 0 && topk <= N && topk <= 1024, "md_snapkv_select: budget-window=%d must be in [1, min(ctx-window,1024)]", topk);
    MD_CHECK_ARG(ctx_len <= 65536, "md_snapkv_select: ctx_len %d > 65536 unsupported", ctx_len);
    MD_CHECK_ARG(pool_kernel >= 1 && (pool_kernel & 1), "md_snapkv_select: pool kernel must be odd");
    MD_CHECK_ARG(workspace_bytes >= md_snapkv_workspace_bytes(B, H, KH, ctx_len, window) &&
                     (((uintptr_t)workspace) & 255) == 0,
                 "md_snapkv_select: workspace too small or not 256-byte aligned");
    MD_CHECK_ARG((kv_dtype & ~(MD_KV_DTYPE_MASK | MD_KV_LAYOUT_HND)) == 0, "md_snapkv_select: unknown kv_dtype flags");
    const bool hnd = (kv_dtype & MD_KV_LAYOUT_HND) != 0;
    kv_dtype &= MD_KV_DTYPE_MASK;
    MD_CHECK_ARG(kv_dtype == MD_KV_BF16 || (kv_dtype == MD_KV_FP8_E4M3 && k_scale && v_scale),
                 "md_snapkv_select: kv_dtype must be MD_KV_BF16 or MD_KV_FP8_E4M3 (with per-head scales)");
    const bool fp8 = kv_dtype == MD_KV_FP8_E4M3;
    hipStream_t st = (hipStream_t)stream;

    SnapParams p;
    p.q = (const bf16_t*)q_win;
    p.cache = cache;
    p.k_scale = k_scale;
    p.page_indices = page_indices;
    p.page_indptr = page_indptr;
    p.B = B;
    p.H = H;
    p.KH = KH;
    p.g = g;
    p.W = window;
    p.S = ctx_len;
    p.L = L;
    p.nch = (ctx_len + kChunkCols - 1) / kChunkCols;
    p.page_size = page_size;
    p.slot_stride = hnd ? D : KH * D;
    p.head_stride = hnd ? page_size * D : D;
    p.page_stride = 2 * (int64_t)page_size * KH * D;
    unsigned char* ws = (unsigned char*)workspace;
    p.partials = (double*)ws;
    const size_t partials_b = align_up((size_t)B * KH * L * p.nch * 2 * 8, 256);
    p.aws = (unsigned short*)(ws + partials_b);
    const size_t aws_b = align_up((size_t)B * H * N * 2, 256);
    unsigned short* scores = (unsigned short*)(ws + partials_b + aws_b);

    const size_t lds1 = (size_t)4 * L * 2 * 8;
    const size_t lds2 = (size_t)L * 2 * 8 + (size_t)4 * g * 16 * 4;
#define MD_SNAP_LAUNCH(DD, FP)                                                                                   \
    do {                                                                                                         \
        hipLaunchKernelGGL((snapkv_stats_kernel<DD, FP>), dim3(p.nch, KH, B), dim3(256), lds1, st, p);            \
        hipLaunchKernelGGL((snapkv_accum_kernel<DD, FP>), dim3((N + 63) / 64, KH, B), dim3(256), lds2, st, p);    \
    } while (0)
    if (D == 128) {
        if (fp8) MD_SNAP_LAUNCH(128, true); else MD_SNAP_LAUNCH(128, false);
    } else {
        if (fp8) MD_SNAP_LAUNCH(64, true); else MD_SNAP_LAUNCH(64, false);
    }
#undef MD_SNAP_LAUNCH
    MD_CHECK_LAUNCH("md_snapkv_select(scores)");

    const size_t lds3 = (size_t)((N + 7) / 8 * 8) * 2 + (256 + 1024 + 32) * 4;
    static MdPerDeviceOnce attr_once;
    if (attr_once.first()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&snapkv_select_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) {
            md_set_error("md_snapkv_select: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return MD_ERR_LAUNCH;
        }
    }
    hipLaunchKernelGGL(snapkv_select_kernel, dim3(KH, B), dim3(1024), lds3, st, p.aws, H, KH, g, N, pool_kernel, topk,
                       idx_out, scores);
    MD_CHECK_LAUNCH("md_snapkv_select(select)");

    GatherParams gp;
    gp.cache = cache;
    gp.k_scale = k_scale;
    gp.v_scale = v_scale;
    gp.page_indices = page_indices;
    gp.page_indptr = page_indptr;
    gp.dcache = (bf16_t*)draft_cache;
    gp.dindices = draft_page_indices;
    gp.dindptr = draft_page_indptr;
    gp.dlast = draft_last_page_len;
    gp.idx = idx_out;
    gp.KH = KH;
    gp.D = D;
    gp.page_size = page_size;
    gp.S = ctx_len;
    gp.W = window;
    gp.budget = budget;
    gp.topk = topk;
    gp.src_hnd = hnd ? 1 : 0;
    if (fp8)
        hipLaunchKernelGGL((snapkv_gather_kernel<true>), dim3(budget, KH, B), dim3(64), 0, st, gp);
    else
        hipLaunchKernelGGL((snapkv_gather_kernel<false>), dim3(budget, KH, B), dim3(64), 0, st, gp);
    MD_CHECK_LAUNCH("md_snapkv_select(gather)");
    return MD_OK;
}
