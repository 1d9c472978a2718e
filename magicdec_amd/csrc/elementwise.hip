// RMSNorm (+residual), SiLU*mul, argmax over a vocab shard, TP argmax merge
// (K9, K10 of SURVEY.md section 2.3) for gfx950.
//
// reference: RMSNorm        Engine/SnapKV/model.py:458-469
//            FeedForward    Engine/SnapKV/model.py:451-455
//            argmax / merge Engine/SnapKV/model.py:175-188
//
// All HBM/L2-streaming kernels: bf16x8 (16 B) per lane, one workgroup per row,
// wave-shuffle + LDS reductions.  bf16 rounding points follow the reference:
//   norm:  y = bf16( float(x) * rsqrt(mean(float(x)^2) + eps) ) * w   (product rounded to bf16)
//   silu:  y = bf16( silu(float(a)) ) * b                             (product rounded to bf16)
#include "md_common.h"

namespace {

__device__ __forceinline__ float block_reduce_sum(float v, float* red) {
    v = wave_reduce_sum(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nw = blockDim.x >> 6;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < nw; ++w) t += red[w];
    __syncthreads();
    return t;
}

// HAS_RES: h = x + r (bf16 add, stored), norm of h
template <bool HAS_RES>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ r,
                                                      const bf16_t* __restrict__ w, bf16_t* h_out, bf16_t* y,
                                                      int dim, float eps) {
    __shared__ float red[8];
    const int64_t row = blockIdx.x;
    const bf16x8* xv = reinterpret_cast<const bf16x8*>(x + row * dim);
    const bf16x8* rv = HAS_RES ? reinterpret_cast<const bf16x8*>(r + row * dim) : nullptr;
    const bf16x8* wv = reinterpret_cast<const bf16x8*>(w);
    const int nvec = dim / 8;
    // dims up to 8192 -> at most 4 vectors per thread; keep them in registers
    constexpr int MAXV = 4;
    f32x8 vals[MAXV];
    float ss = 0.f;
#pragma unroll
    for (int it = 0; it < MAXV; ++it) {
        const int i = threadIdx.x + it * 256;
        if (i < nvec) {
            f32x8 f = __builtin_convertvector(xv[i], f32x8);
            if (HAS_RES) {
                const f32x8 rf = __builtin_convertvector(rv[i], f32x8);
                const bf16x8 hb = __builtin_convertvector(f + rf, bf16x8);  // bf16 add
                reinterpret_cast<bf16x8*>(h_out + row * dim)[i] = hb;
                f = __builtin_convertvector(hb, f32x8);
            }
            vals[it] = f;
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += f[e] * f[e];
        }
    }
    const float tot = block_reduce_sum(ss, red);
    const float rs = rsqrtf(tot / (float)dim + eps);
#pragma unroll
    for (int it = 0; it < MAXV; ++it) {
        const int i = threadIdx.x + it * 256;
        if (i < nvec) {
            const bf16x8 nb = __builtin_convertvector(vals[it] * rs, bf16x8);
            const f32x8 nf = __builtin_convertvector(nb, f32x8);
            const f32x8 wf = __builtin_convertvector(wv[i], f32x8);
            reinterpret_cast<bf16x8*>(y + row * dim)[i] = __builtin_convertvector(nf * wf, bf16x8);
        }
    }
}

// Split-K combine of md_linear + residual add + RMSNorm in one launch (md_linear_add_rmsnorm): one workgroup per row,
// the thread -> column mapping, the order of every sum and every rounding point are those of skinny_reduce_kernel
// (csrc/gemm.hip) followed by rmsnorm_kernel<true> above, so (h, y) are bit-identical to md_linear -> md_add_rmsnorm.
//   o = bf16(sum_s partial[s] + bias)   [int8 weights: bf16(bf16(sum) * scale)];  h = bf16(x + o);  y = rmsnorm(h) * w
template <bool W8>
__global__ __launch_bounds__(256) void reduce_add_rmsnorm_kernel(const float* __restrict__ partial, int S, int M, int N,
                                                                 const bf16_t* __restrict__ bias,
                                                                 const bf16_t* __restrict__ scales,
                                                                 const bf16_t* __restrict__ x, int64_t ldx,
                                                                 const bf16_t* __restrict__ w, bf16_t* h_out,
                                                                 bf16_t* y, float eps) {
    __shared__ float red[8];
    const int64_t row = blockIdx.x;
    const int nvec = N / 8;
    const int64_t plane = (int64_t)M * N;
    constexpr int MAXV = 4;
    f32x8 vals[MAXV];
    float ss = 0.f;
#pragma unroll
    for (int it = 0; it < MAXV; ++it) {
        const int i = threadIdx.x + it * 256;
        if (i < nvec) {
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
            // eight slices per trip, requested together and added in slice order: a load + add per loop trip was one
            // memory round trip per slice (8.3 us for 32 rows x 8 slices; skinny_reduce_kernel does the same)
            for (int s0 = 0; s0 < S; s0 += 8) {
                f32x4 v0[8], v1[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (s0 + u < S) {                               // uniform: no load for a slice that does not exist
                        const float* pp = partial + (s0 + u) * plane + row * N + i * 8;
                        v0[u] = *reinterpret_cast<const f32x4*>(pp);
                        v1[u] = *reinterpret_cast<const f32x4*>(pp + 4);
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (s0 + u < S) {                               // uniform
                        a0 += v0[u];
                        a1 += v1[u];
                    }
                }
            }
            f32x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = e < 4 ? a0[e] : a1[e - 4];
                if (bias) v += bf16_to_f32(bias[i * 8 + e]);
                if (W8) v = bf16_to_f32(f32_to_bf16(v)) * bf16_to_f32(scales[i * 8 + e]);
                o[e] = bf16_to_f32(f32_to_bf16(v));
            }
            const f32x8 xf = __builtin_convertvector(reinterpret_cast<const bf16x8*>(x + row * ldx)[i], f32x8);
            const bf16x8 hb = __builtin_convertvector(xf + o, bf16x8);      // bf16 add
            reinterpret_cast<bf16x8*>(h_out + row * N)[i] = hb;
            const f32x8 f = __builtin_convertvector(hb, f32x8);
            vals[it] = f;
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += f[e] * f[e];
        }
    }
    const float tot = block_reduce_sum(ss, red);
    const float rs = rsqrtf(tot / (float)N + eps);
    const bf16x8* wv = reinterpret_cast<const bf16x8*>(w);
#pragma unroll
    for (int it = 0; it < MAXV; ++it) {
        const int i = threadIdx.x + it * 256;
        if (i < nvec) {
            const bf16x8 nb = __builtin_convertvector(vals[it] * rs, bf16x8);
            const f32x8 nf = __builtin_convertvector(nb, f32x8);
            const f32x8 wf = __builtin_convertvector(wv[i], f32x8);
            reinterpret_cast<bf16x8*>(y + row * N)[i] = __builtin_convertvector(nf * wf, bf16x8);
        }
    }
}

__global__ __launch_bounds__(256) void silu_mul_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                       int64_t as, int64_t bs, bf16_t* y, int rows, int dim) {
    const int nvec = dim / 8;
    const int64_t total = (int64_t)rows * nvec;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / nvec;
        const int c = (int)(i - row * nvec);
        const f32x8 af = __builtin_convertvector(*reinterpret_cast<const bf16x8*>(a + row * as + c * 8), f32x8);
        const f32x8 bf = __builtin_convertvector(*reinterpret_cast<const bf16x8*>(b + row * bs + c * 8), f32x8);
        f32x8 s;
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = af[e] / (1.0f + expf(-af[e]));
        const bf16x8 sb = __builtin_convertvector(s, bf16x8);
        const f32x8 sf = __builtin_convertvector(sb, f32x8);
        *reinterpret_cast<bf16x8*>(y + row * dim + c * 8) = __builtin_convertvector(sf * bf, bf16x8);
    }
}

// one workgroup of NT threads per row; lowest index among equal maxima.  Decode steps have few rows (64..256) of a
// 128K vocabulary: with NT = 1024 (16 waves, 4 loads in flight each) a row is limited by its CU's load issue rate
// instead of by one wave's memory latency (49 -> ~15 us for 64 rows)
// slots > 0 (md_argmax_tp_slots): the outputs are [rows, slots] and the row's result goes to column `slot`, zeros to the
// others -- the one-hot-slot tensors the reference builds with torch.zeros + an index assignment before its two
// all-reduces (Engine/SnapKV/model.py:178-184), in the argmax launch itself
template <int NT>
__global__ __launch_bounds__(NT) void argmax_kernel(const bf16_t* __restrict__ logits, int64_t row_stride, int vocab,
                                                    int64_t index_offset, bf16_t* max_val_out, int64_t* idx_out,
                                                    int slots, int slot) {
    __shared__ float sv[NT / 64];
    __shared__ int si[NT / 64];
    const int64_t row = blockIdx.x;
    const bf16_t* p = logits + row * row_stride;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    const int nvec = vocab / 8;
    const bool vec_ok = (((uintptr_t)p) & 15) == 0;
    if (vec_ok) {
#pragma unroll 4
        for (int i = threadIdx.x; i < nvec; i += NT) {
            const f32x8 f = __builtin_convertvector(reinterpret_cast<const bf16x8*>(p)[i], f32x8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = f[e];
                const int id = i * 8 + e;
                if (v > best || (v == best && id < bi)) {
                    best = v;
                    bi = id;
                }
            }
        }
        for (int id = nvec * 8 + threadIdx.x; id < vocab; id += NT) {
            const float v = bf16_to_f32(p[id]);
            if (v > best || (v == best && id < bi)) {
                best = v;
                bi = id;
            }
        }
    } else {
        for (int id = threadIdx.x; id < vocab; id += NT) {
            const float v = bf16_to_f32(p[id]);
            if (v > best || (v == best && id < bi)) {
                best = v;
                bi = id;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > best || (ov == best && oi < bi)) {
            best = ov;
            bi = oi;
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        sv[wave] = best;
        si[wave] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < NT / 64; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) {
                best = sv[w];
                bi = si[w];
            }
        if (bi == 0x7fffffff) bi = 0;  // all -inf / NaN row: torch.argmax returns an index too; pick 0
        if (slots > 0) {
            for (int s = 0; s < slots; ++s) {
                idx_out[row * slots + s] = s == slot ? (int64_t)bi + index_offset : (int64_t)0;
                max_val_out[row * slots + s] = f32_to_bf16(s == slot ? best : 0.f);
            }
        } else {
            idx_out[row] = (int64_t)bi + index_offset;
            if (max_val_out) max_val_out[row] = f32_to_bf16(best);
        }
    }
}

__global__ void tp_merge_kernel(const bf16_t* __restrict__ vals, const int64_t* __restrict__ idx, int rows, int tp,
                                int64_t* out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    float best = bf16_to_f32(vals[(int64_t)r * tp]);
    int br = 0;
    for (int t = 1; t < tp; ++t) {
        const float v = bf16_to_f32(vals[(int64_t)r * tp + t]);
        if (v > best) {
            best = v;
            br = t;
        }
    }
    out[r] = idx[(int64_t)r * tp + br];
}

}  // namespace

// launched by md_linear_add_rmsnorm (csrc/gemm.hip) behind its split-K main kernel
int md_internal_launch_reduce_add_rmsnorm(const float* partial, int S, int M, int N, const void* bias, const void* scales,
                                          const void* x, int64_t ldx, const void* w, void* h_out, void* y, float eps,
                                          hipStream_t st) {
    if (scales)
        hipLaunchKernelGGL((reduce_add_rmsnorm_kernel<true>), dim3(M), dim3(256), 0, st, partial, S, M, N,
                           (const bf16_t*)bias, (const bf16_t*)scales, (const bf16_t*)x, ldx, (const bf16_t*)w,
                           (bf16_t*)h_out, (bf16_t*)y, eps);
    else
        hipLaunchKernelGGL((reduce_add_rmsnorm_kernel<false>), dim3(M), dim3(256), 0, st, partial, S, M, N,
                           (const bf16_t*)bias, (const bf16_t*)scales, (const bf16_t*)x, ldx, (const bf16_t*)w,
                           (bf16_t*)h_out, (bf16_t*)y, eps);
    return MD_OK;
}


extern "C" int md_rmsnorm(const void* x, const void* weight, void* y, int rows, int dim, float eps,
                          md_stream_t stream) {
    MD_CHECK_ARG(x && weight && y, "md_rmsnorm: null pointer argument");
    MD_CHECK_ARG(rows > 0 && dim > 0 && dim % 8 == 0 && dim <= 8192, "md_rmsnorm: dim %d must be a multiple of 8, <= 8192", dim);
    MD_CHECK_ARG((((uintptr_t)x | (uintptr_t)weight | (uintptr_t)y) & 15) == 0, "md_rmsnorm: 16-byte alignment required");
    hipLaunchKernelGGL((rmsnorm_kernel<false>), dim3(rows), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       (const bf16_t*)nullptr, (const bf16_t*)weight, (bf16_t*)nullptr, (bf16_t*)y, dim, eps);
    MD_CHECK_LAUNCH("md_rmsnorm");
    return MD_OK;
}

extern "C" int md_add_rmsnorm(const void* x, const void* r, const void* weight, void* h_out, void* y, int rows,
                              int dim, float eps, md_stream_t stream) {
    MD_CHECK_ARG(x && r && weight && h_out && y, "md_add_rmsnorm: null pointer argument");
    MD_CHECK_ARG(rows > 0 && dim > 0 && dim % 8 == 0 && dim <= 8192, "md_add_rmsnorm: dim %d must be a multiple of 8, <= 8192", dim);
    MD_CHECK_ARG((((uintptr_t)x | (uintptr_t)r | (uintptr_t)weight | (uintptr_t)h_out | (uintptr_t)y) & 15) == 0,
                 "md_add_rmsnorm: 16-byte alignment required");
    hipLaunchKernelGGL((rmsnorm_kernel<true>), dim3(rows), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       (const bf16_t*)r, (const bf16_t*)weight, (bf16_t*)h_out, (bf16_t*)y, dim, eps);
    MD_CHECK_LAUNCH("md_add_rmsnorm");
    return MD_OK;
}

extern "C" int md_silu_mul(const void* a, const void* b, int64_t a_row_stride, int64_t b_row_stride, void* y,
                           int rows, int dim, md_stream_t stream) {
    MD_CHECK_ARG(a && b && y, "md_silu_mul: null pointer argument");
    MD_CHECK_ARG(rows > 0 && dim > 0 && dim % 8 == 0 && a_row_stride % 8 == 0 && b_row_stride % 8 == 0,
                 "md_silu_mul: dim and row strides must be multiples of 8");
    MD_CHECK_ARG((((uintptr_t)a | (uintptr_t)b | (uintptr_t)y) & 15) == 0, "md_silu_mul: 16-byte alignment required");
    const int64_t total = (int64_t)rows * (dim / 8);
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(silu_mul_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a,
                       (const bf16_t*)b, a_row_stride, b_row_stride, (bf16_t*)y, rows, dim);
    MD_CHECK_LAUNCH("md_silu_mul");
    return MD_OK;
}

extern "C" int md_argmax(const void* logits, int64_t row_stride, int rows, int vocab, int64_t index_offset,
                         void* max_val_out, int64_t* idx_out, md_stream_t stream) {
    MD_CHECK_ARG(logits && idx_out, "md_argmax: null pointer argument");
    MD_CHECK_ARG(rows > 0 && vocab > 0, "md_argmax: bad shape rows=%d vocab=%d", rows, vocab);
    if (rows <= 512 && vocab >= 16384)
        hipLaunchKernelGGL((argmax_kernel<1024>), dim3(rows), dim3(1024), 0, (hipStream_t)stream, (const bf16_t*)logits,
                           row_stride, vocab, index_offset, (bf16_t*)max_val_out, idx_out, 0, 0);
    else
        hipLaunchKernelGGL((argmax_kernel<256>), dim3(rows), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)logits,
                           row_stride, vocab, index_offset, (bf16_t*)max_val_out, idx_out, 0, 0);
    MD_CHECK_LAUNCH("md_argmax");
    return MD_OK;
}

extern "C" int md_argmax_tp_slots(const void* logits, int64_t row_stride, int rows, int vocab, int64_t index_offset,
                                  int tp_rank, int tp_world, void* vals_out, int64_t* idx_out, md_stream_t stream) {
    MD_CHECK_ARG(logits && vals_out && idx_out, "md_argmax_tp_slots: null pointer argument");
    MD_CHECK_ARG(rows > 0 && vocab > 0, "md_argmax_tp_slots: bad shape rows=%d vocab=%d", rows, vocab);
    MD_CHECK_ARG(tp_world >= 1 && tp_world <= 64 && tp_rank >= 0 && tp_rank < tp_world,
                 "md_argmax_tp_slots: need 1 <= tp_world <= 64 and 0 <= tp_rank < tp_world (got %d of %d)", tp_rank, tp_world);
    if (rows <= 512 && vocab >= 16384)
        hipLaunchKernelGGL((argmax_kernel<1024>), dim3(rows), dim3(1024), 0, (hipStream_t)stream, (const bf16_t*)logits,
                           row_stride, vocab, index_offset, (bf16_t*)vals_out, idx_out, tp_world, tp_rank);
    else
        hipLaunchKernelGGL((argmax_kernel<256>), dim3(rows), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)logits,
                           row_stride, vocab, index_offset, (bf16_t*)vals_out, idx_out, tp_world, tp_rank);
    MD_CHECK_LAUNCH("md_argmax_tp_slots");
    return MD_OK;
}

extern "C" int md_tp_argmax_merge(const void* vals, const int64_t* idx, int rows, int tp, int64_t* out,
                                  md_stream_t stream) {
    MD_CHECK_ARG(vals && idx && out && rows > 0 && tp > 0, "md_tp_argmax_merge: bad arguments");
    hipLaunchKernelGGL(tp_merge_kernel, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)vals, idx, rows, tp, out);
    MD_CHECK_LAUNCH("md_tp_argmax_merge");
    return MD_OK;
}
