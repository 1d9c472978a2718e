// Paged-KV append and RoPE for gfx950 (K4, K5 of SURVEY.md section 2.3).
//
//   md_append_paged_kv  <- flashinfer.append_paged_kv_cache   (Engine/utils.py:36-54)
//   md_rope             <- flashinfer.rope.apply_rope / apply_llama31_rope, interleave=True
//                          (Engine/SnapKV/model.py:133-156)
//   md_rope_append      <- the two fused (one launch per layer instead of two)
//
// All three are pure streaming kernels: one workgroup per (token row, request),
// 16 B per lane, rows are whole contiguous KH*D / H*D segments.  RoPE uses a
// host-precomputed fp32 cos/sin table (on-device powf/sincosf would make it
// VALU-bound) and un-fused fp32 multiplies/adds so the result is bit-identical
// to the oracle's numpy arithmetic (x_e*cos - x_o*sin, x_o*cos + x_e*sin).
#include "md_common.h"
#include <math.h>

// rows dropped by an append because they fell beyond the request's mapped pages (md_page_overflow_count)
__device__ unsigned int g_md_page_overflow = 0;

namespace {

__device__ __forceinline__ void note_overflow() {
    if (threadIdx.x == 0) atomicAdd(&g_md_page_overflow, 1u);
}

__device__ __forceinline__ int req_len(const int32_t* indptr, const int32_t* last, int b, int page_size,
                                       int* pg0) {
    const int p0 = indptr[b];
    const int np = indptr[b + 1] - p0;
    *pg0 = p0;
    return np > 0 ? (np - 1) * page_size + last[b] : 0;
}

// Rows a request may hold: an over-long last_page_len (> page_size: the caller never mapped the next page -- the
// reference's page tables do not grow during decode, Engine/SnapKV/backend.py:147) must not index past the
// request's own page list; such rows are dropped (the host raises, harness.check_page_bounds).
__device__ __forceinline__ int req_capacity(const int32_t* indptr, int b, int page_size) {
    return (indptr[b + 1] - indptr[b]) * page_size;
}

// destination (element offset) of row `pos` of request b in a paged cache, K half
__device__ __forceinline__ int64_t page_row_offset(const int32_t* indices, int pg0, int pos, int page_size,
                                                   int KH, int D) {
    const int page = pos / page_size;
    const int slot = pos - page * page_size;
    const int64_t pid = indices[pg0 + page];
    return (pid * 2 * page_size + slot) * (int64_t)(KH * D);
}

// element offset of 8-element chunk c of kv head h of row `pos` of a request (K half; V half = + page_size*KH*D)
__device__ __forceinline__ int64_t kv_chunk_offset(const int32_t* indices, int pg0, int pos, int page_size, int KH,
                                                   int D, int h, int c, bool hnd) {
    const int page = pos / page_size;
    const int slot = pos - page * page_size;
    const int64_t base = (int64_t)indices[pg0 + page] * 2 * page_size * KH * D;
    return hnd ? base + ((int64_t)h * page_size + slot) * D + c * 8 : base + ((int64_t)slot * KH + h) * D + c * 8;
}

// 8 bf16 -> 8 OCP e4m3fn bytes: q = rne_e4m3(clamp(x * inv_scale, +-448))   (oracle: flashinfer_ref.quantize_fp8)
__device__ __forceinline__ u32x2 quant8_fp8(const u32x4 x, float inv_scale) {
    float f[8];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        f[2 * w] = __uint_as_float(x[w] << 16);
        f[2 * w + 1] = __uint_as_float(x[w] & 0xffff0000u);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = fminf(fmaxf(__fmul_rn(f[e], inv_scale), -448.f), 448.f);
    u32x2 r = {0u, 0u};
    r[0] = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], r[0], false);
    r[0] = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], r[0], true);
    r[1] = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], r[1], false);
    r[1] = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], r[1], true);
    return r;
}

// store 8 consecutive cache elements starting at element offset `off` (bf16 or fp8 cache)
template <bool FP8>
__device__ __forceinline__ void store8(void* cache, int64_t off, const u32x4 v, float inv_scale) {
    if constexpr (FP8)
        *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned char*>(cache) + off) = quant8_fp8(v, inv_scale);
    else
        *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(cache) + off) = v;
}

template <bool FP8>
__global__ __launch_bounds__(256) void append_kernel(const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                     int64_t ks, int64_t vs, const int32_t* append_indptr,
                                                     void* cache, const int32_t* indices,
                                                     const int32_t* indptr, const int32_t* last, int KH, int D,
                                                     int page_size, const float* k_scale, const float* v_scale,
                                                     int hnd) {
    const int b = blockIdx.y, j = blockIdx.x;
    const int a0 = append_indptr[b];
    const int n_b = append_indptr[b + 1] - a0;
    if (j >= n_b) return;
    int pg0;
    const int len = req_len(indptr, last, b, page_size, &pg0);
    const int pos = len - n_b + j;
    if (pos < 0) return;
    if (pos >= req_capacity(indptr, b, page_size)) {
        note_overflow();
        return;
    }
    const int64_t half = (int64_t)page_size * KH * D;
    const int nvec = KH * D / 8;
    const u32x4* ksrc = reinterpret_cast<const u32x4*>(k + (int64_t)(a0 + j) * ks);
    const u32x4* vsrc = reinterpret_cast<const u32x4*>(v + (int64_t)(a0 + j) * vs);
    const int cpr = D / 8;
    for (int i = threadIdx.x; i < nvec; i += blockDim.x) {
        const int h = i / cpr;
        const int64_t dst = kv_chunk_offset(indices, pg0, pos, page_size, KH, D, h, i - h * cpr, hnd != 0);
        store8<FP8>(cache, dst, ksrc[i], FP8 ? 1.0f / k_scale[h] : 1.f);
        store8<FP8>(cache, dst + half, vsrc[i], FP8 ? 1.0f / v_scale[h] : 1.f);
    }
}

// rotate 8 consecutive bf16 (4 interleaved pairs) by table entries cs[0..7] = (cos,sin) x4
__device__ __forceinline__ u32x4 rope8(const u32x4 x, const float* __restrict__ cs) {
    const f32x4 c0 = *reinterpret_cast<const f32x4*>(cs);      // cos0 sin0 cos1 sin1
    const f32x4 c1 = *reinterpret_cast<const f32x4*>(cs + 4);  // cos2 sin2 cos3 sin3
    u32x4 r;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float xe = __uint_as_float(x[w] << 16);
        const float xo = __uint_as_float(x[w] & 0xffff0000u);
        const float co = w < 2 ? c0[(w & 1) * 2] : c1[(w & 1) * 2];
        const float si = w < 2 ? c0[(w & 1) * 2 + 1] : c1[(w & 1) * 2 + 1];
        const float ye = __fsub_rn(__fmul_rn(xe, co), __fmul_rn(xo, si));
        const float yo = __fadd_rn(__fmul_rn(xo, co), __fmul_rn(xe, si));
        const bf16x2 pk = {f32_to_bf16(ye), f32_to_bf16(yo)};
        r[w] = *reinterpret_cast<const unsigned int*>(&pk);
    }
    return r;
}

// q_out/k_out contiguous.  k may be null.
__global__ __launch_bounds__(256) void rope_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                   int64_t qs, int64_t ks, bf16_t* q_out, bf16_t* k_out,
                                                   const int32_t* indptr, const int32_t* offsets, int H, int KH,
                                                   int D, const float* __restrict__ cos_sin, int max_pos) {
    const int b = blockIdx.y, j = blockIdx.x;
    const int a0 = indptr[b];
    const int n_b = indptr[b + 1] - a0;
    if (j >= n_b) return;
    int pos = offsets[b] + j;
    pos = pos < 0 ? 0 : (pos >= max_pos ? max_pos - 1 : pos);
    const float* cs = cos_sin + (int64_t)pos * D;  // [D/2][2]
    const int cpr = D / 8;                         // 16-B chunks per head
    const int row = a0 + j;
    const int nq = H * cpr, nk = k ? KH * cpr : 0;
    for (int i = threadIdx.x; i < nq + nk; i += blockDim.x) {
        if (i < nq) {
            const int c = i % cpr;
            const u32x4 x = *reinterpret_cast<const u32x4*>(q + (int64_t)row * qs + i * 8);
            *reinterpret_cast<u32x4*>(q_out + (int64_t)row * H * D + i * 8) = rope8(x, cs + c * 8);
        } else {
            const int ii = i - nq;
            const int c = ii % cpr;
            const u32x4 x = *reinterpret_cast<const u32x4*>(k + (int64_t)row * ks + ii * 8);
            *reinterpret_cast<u32x4*>(k_out + (int64_t)row * KH * D + ii * 8) = rope8(x, cs + c * 8);
        }
    }
}

struct PageTable {
    void* cache;
    const int32_t* indices;
    const int32_t* indptr;
    const int32_t* last;
};

template <bool FP8>
__global__ __launch_bounds__(256) void rope_append_kernel(const bf16_t* __restrict__ q,
                                                          const bf16_t* __restrict__ k,
                                                          const bf16_t* __restrict__ v, int64_t qs, int64_t ks,
                                                          int64_t vs, bf16_t* q_out, const int32_t* indptr,
                                                          const int32_t* offsets, int H, int KH, int D,
                                                          const float* __restrict__ cos_sin, int max_pos,
                                                          PageTable t1, PageTable t2, int page_size,
                                                          const float* k_scale, const float* v_scale, int hnd) {
    const int b = blockIdx.y, j = blockIdx.x;
    const int a0 = indptr[b];
    const int n_b = indptr[b + 1] - a0;
    if (j >= n_b) return;
    int pos = offsets[b] + j;
    pos = pos < 0 ? 0 : (pos >= max_pos ? max_pos - 1 : pos);
    const float* cs = cos_sin + (int64_t)pos * D;
    const int cpr = D / 8;
    const int row = a0 + j;
    const int nq = H * cpr, nk = KH * cpr;
    const int64_t half = (int64_t)page_size * KH * D;

    int pg0;
    const int len1 = req_len(t1.indptr, t1.last, b, page_size, &pg0);
    const int p1 = len1 - n_b + j;
    const bool over1 = p1 >= req_capacity(t1.indptr, b, page_size);
    const bool ok1 = p1 >= 0 && !over1;
    const int pg1 = pg0;
    if (over1) note_overflow();
    int64_t d2 = -1;
    if (t2.cache) {
        const int len2 = req_len(t2.indptr, t2.last, b, page_size, &pg0);
        const int p2 = len2 - n_b + j;
        const bool over2 = p2 >= req_capacity(t2.indptr, b, page_size);
        d2 = (p2 >= 0 && !over2) ? page_row_offset(t2.indices, pg0, p2, page_size, KH, D) : -1;
        if (over2) note_overflow();
    }
    for (int i = threadIdx.x; i < nq + 2 * nk; i += blockDim.x) {
        if (i < nq) {
            const int c = i % cpr;
            const u32x4 x = *reinterpret_cast<const u32x4*>(q + (int64_t)row * qs + i * 8);
            *reinterpret_cast<u32x4*>(q_out + (int64_t)row * H * D + i * 8) = rope8(x, cs + c * 8);
        } else if (i < nq + nk) {
            const int ii = i - nq;
            const int c = ii % cpr;
            const u32x4 x = *reinterpret_cast<const u32x4*>(k + (int64_t)row * ks + ii * 8);
            const u32x4 y = rope8(x, cs + c * 8);
            // the first cache may be fp8 (target); a second cache (self-spec draft cache) is always bf16
            const int h = ii / cpr;
            if (ok1)
                store8<FP8>(t1.cache, kv_chunk_offset(t1.indices, pg1, p1, page_size, KH, D, h, c, hnd != 0), y,
                            FP8 ? 1.0f / k_scale[h] : 1.f);
            if (d2 >= 0) store8<false>(t2.cache, d2 + ii * 8, y, 1.f);
        } else {
            const int ii = i - nq - nk;
            const u32x4 x = *reinterpret_cast<const u32x4*>(v + (int64_t)row * vs + ii * 8);
            const int h = ii / cpr;
            if (ok1)
                store8<FP8>(t1.cache,
                            kv_chunk_offset(t1.indices, pg1, p1, page_size, KH, D, h, ii - h * cpr, hnd != 0) + half,
                            x, FP8 ? 1.0f / v_scale[h] : 1.f);
            if (d2 >= 0) store8<false>(t2.cache, d2 + half + ii * 8, x, 1.f);
        }
    }
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

// device address of the dropped-row counter, for kernels of other translation units (csrc/tilegemm.hip, blockgemm.hip).
// Resolved once per device: the callers sit on the launch-bound path (one call per layer per step, also inside graph
// capture), where a runtime call per launch is avoidable host latency.
unsigned int* md_page_overflow_counter_device() {
    static unsigned int* cached[MD_MAX_DEVICES] = {};
    int d = 0;
    const bool slot = hipGetDevice(&d) == hipSuccess && d >= 0 && d < MD_MAX_DEVICES;
    if (slot && cached[d]) return cached[d];
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_md_page_overflow)) != hipSuccess) return nullptr;
    if (slot) cached[d] = (unsigned int*)p;
    return (unsigned int*)p;
}

extern "C" int md_page_overflow_count(unsigned int* count_host, int reset) {
    MD_CHECK_ARG(count_host, "md_page_overflow_count: null pointer argument");
    hipError_t e = hipMemcpyFromSymbol(count_host, HIP_SYMBOL(g_md_page_overflow), sizeof(unsigned int));
    if (e == hipSuccess && reset && *count_host) {
        const unsigned int z = 0;
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_md_page_overflow), &z, sizeof(z));
    }
    if (e != hipSuccess) {
        md_set_error("md_page_overflow_count: %s", hipGetErrorString(e));
        return MD_ERR_LAUNCH;
    }
    return MD_OK;
}

extern "C" int md_append_paged_kv(const void* k, const void* v, int64_t k_row_stride, int64_t v_row_stride,
                                  const int32_t* append_indptr, void* cache, const int32_t* page_indices,
                                  const int32_t* page_indptr, const int32_t* last_page_len, int B, int n_max,
                                  int KH, int D, int page_size, int kv_dtype, const float* k_scale,
                                  const float* v_scale, md_stream_t stream) {
    MD_CHECK_ARG(k && v && append_indptr && cache && page_indices && page_indptr && last_page_len,
                 "md_append_paged_kv: null pointer argument");
    MD_CHECK_ARG(B > 0 && n_max > 0 && KH > 0 && D > 0 && D % 8 == 0 && page_size > 0,
                 "md_append_paged_kv: bad shape B=%d n_max=%d KH=%d D=%d", B, n_max, KH, D);
    MD_CHECK_ARG(aligned16(k) && aligned16(v) && aligned16(cache) && k_row_stride % 8 == 0 && v_row_stride % 8 == 0,
                 "md_append_paged_kv: k/v/cache must be 16-byte aligned, row strides multiples of 8");
    const int hnd = (kv_dtype & MD_KV_LAYOUT_HND) ? 1 : 0;
    MD_CHECK_ARG((kv_dtype & ~(MD_KV_DTYPE_MASK | MD_KV_LAYOUT_HND)) == 0, "md_append_paged_kv: unknown kv_dtype flags");
    kv_dtype &= MD_KV_DTYPE_MASK;
    MD_CHECK_ARG(kv_dtype == MD_KV_BF16 || (kv_dtype == MD_KV_FP8_E4M3 && k_scale && v_scale),
                 "md_append_paged_kv: kv_dtype must be MD_KV_BF16 or MD_KV_FP8_E4M3 (with per-head scales)");
    if (kv_dtype == MD_KV_FP8_E4M3)
        hipLaunchKernelGGL((append_kernel<true>), dim3(n_max, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)k,
                           (const bf16_t*)v, k_row_stride, v_row_stride, append_indptr, cache, page_indices,
                           page_indptr, last_page_len, KH, D, page_size, k_scale, v_scale, hnd);
    else
        hipLaunchKernelGGL((append_kernel<false>), dim3(n_max, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)k,
                           (const bf16_t*)v, k_row_stride, v_row_stride, append_indptr, cache, page_indices,
                           page_indptr, last_page_len, KH, D, page_size, k_scale, v_scale, hnd);
    MD_CHECK_LAUNCH("md_append_paged_kv");
    return MD_OK;
}

extern "C" int md_rope(const void* q, const void* k, int64_t q_row_stride, int64_t k_row_stride, void* q_out,
                       void* k_out, const int32_t* indptr, const int32_t* offsets, int B, int n_max, int H, int KH,
                       int D, const float* cos_sin, int max_pos, md_stream_t stream) {
    MD_CHECK_ARG(q && q_out && indptr && offsets && cos_sin, "md_rope: null pointer argument");
    MD_CHECK_ARG((k == nullptr) == (k_out == nullptr), "md_rope: k and k_out must both be given or both NULL");
    MD_CHECK_ARG(B > 0 && n_max > 0 && H > 0 && KH >= 0 && D > 0 && D % 8 == 0 && max_pos > 0,
                 "md_rope: bad shape B=%d n_max=%d H=%d KH=%d D=%d", B, n_max, H, KH, D);
    MD_CHECK_ARG(aligned16(q) && aligned16(q_out) && aligned16(cos_sin) && q_row_stride % 8 == 0 &&
                     (!k || (aligned16(k) && aligned16(k_out) && k_row_stride % 8 == 0)),
                 "md_rope: tensors must be 16-byte aligned, row strides multiples of 8");
    hipLaunchKernelGGL(rope_kernel, dim3(n_max, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q,
                       (const bf16_t*)k, q_row_stride, k_row_stride, (bf16_t*)q_out, (bf16_t*)k_out, indptr, offsets,
                       H, KH, D, cos_sin, max_pos);
    MD_CHECK_LAUNCH("md_rope");
    return MD_OK;
}

extern "C" int md_rope_fill_table_host(float* table_host, int max_pos, int D, double theta, double rope_scale,
                                       double low_freq_factor, double high_freq_factor,
                                       double old_context_len) {
    MD_CHECK_ARG(table_host && max_pos > 0 && D > 0 && D % 2 == 0, "md_rope_fill_table_host: bad arguments");
    const int half = D / 2;
    const double two_pi = 6.283185307179586476925286766559;
    for (int i = 0; i < half; ++i) {
        double f = pow(theta, -2.0 * (double)i / (double)D);
        if (low_freq_factor > 0.0 && high_freq_factor > 0.0) {
            double s = (old_context_len * f / two_pi - low_freq_factor) / (high_freq_factor - low_freq_factor);
            s = s < 0.0 ? 0.0 : (s > 1.0 ? 1.0 : s);
            f = (1.0 - s) * f / rope_scale + s * f;
        } else {
            f = f / rope_scale;
        }
        for (int p = 0; p < max_pos; ++p) {
            const double a = (double)p * f;
            table_host[((size_t)p * half + i) * 2 + 0] = (float)cos(a);
            table_host[((size_t)p * half + i) * 2 + 1] = (float)sin(a);
        }
    }
    return MD_OK;
}

extern "C" int md_rope_append(const void* q, const void* k, const void* v, int64_t q_row_stride,
                              int64_t k_row_stride, int64_t v_row_stride, void* q_out, const int32_t* indptr,
                              const int32_t* offsets, int B, int n_max, int H, int KH, int D, const float* cos_sin,
                              int max_pos, void* cache, const int32_t* page_indices, const int32_t* page_indptr,
                              const int32_t* last_page_len, void* cache2, const int32_t* page_indices2,
                              const int32_t* page_indptr2, const int32_t* last_page_len2, int page_size,
                              int kv_dtype, const float* k_scale, const float* v_scale, md_stream_t stream) {
    MD_CHECK_ARG(q && k && v && q_out && indptr && offsets && cos_sin && cache && page_indices && page_indptr &&
                     last_page_len,
                 "md_rope_append: null pointer argument");
    MD_CHECK_ARG(!cache2 || (page_indices2 && page_indptr2 && last_page_len2),
                 "md_rope_append: second cache needs its page table");
    MD_CHECK_ARG(B > 0 && n_max > 0 && H > 0 && KH > 0 && D > 0 && D % 8 == 0 && max_pos > 0 && page_size > 0,
                 "md_rope_append: bad shape");
    MD_CHECK_ARG(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(q_out) && aligned16(cache) &&
                     aligned16(cos_sin) && (!cache2 || aligned16(cache2)) && q_row_stride % 8 == 0 &&
                     k_row_stride % 8 == 0 && v_row_stride % 8 == 0,
                 "md_rope_append: tensors must be 16-byte aligned, row strides multiples of 8");
    const int hnd = (kv_dtype & MD_KV_LAYOUT_HND) ? 1 : 0;
    MD_CHECK_ARG((kv_dtype & ~(MD_KV_DTYPE_MASK | MD_KV_LAYOUT_HND)) == 0, "md_rope_append: unknown kv_dtype flags");
    kv_dtype &= MD_KV_DTYPE_MASK;
    MD_CHECK_ARG(kv_dtype == MD_KV_BF16 || (kv_dtype == MD_KV_FP8_E4M3 && k_scale && v_scale),
                 "md_rope_append: kv_dtype must be MD_KV_BF16 or MD_KV_FP8_E4M3 (with per-head scales)");
    PageTable t1{cache, page_indices, page_indptr, last_page_len};
    PageTable t2{cache2, page_indices2, page_indptr2, last_page_len2};
    if (kv_dtype == MD_KV_FP8_E4M3)
        hipLaunchKernelGGL((rope_append_kernel<true>), dim3(n_max, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q,
                           (const bf16_t*)k, (const bf16_t*)v, q_row_stride, k_row_stride, v_row_stride, (bf16_t*)q_out,
                           indptr, offsets, H, KH, D, cos_sin, max_pos, t1, t2, page_size, k_scale, v_scale, hnd);
    else
        hipLaunchKernelGGL((rope_append_kernel<false>), dim3(n_max, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q,
                           (const bf16_t*)k, (const bf16_t*)v, q_row_stride, k_row_stride, v_row_stride, (bf16_t*)q_out,
                           indptr, offsets, H, KH, D, cos_sin, max_pos, t1, t2, page_size, k_scale, v_scale, hnd);
    MD_CHECK_LAUNCH("md_rope_append");
    return MD_OK;
}
