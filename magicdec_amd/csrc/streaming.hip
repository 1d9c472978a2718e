// StreamingLLM sink+window eviction for the draft KV cache (K7, row a9 of
// SURVEY.md section 8), gfx950.
//
// reference: KVCache.prefill        Engine/StreamingLLM/model_draft.py:102-143
//            KVCache.prefill_draft  Engine/StreamingLLM/model.py:116-157
//
// The reference, per 128-token prefill chunk and per layer, builds
//   new_k = cat(cache[:, sink:kv_len], k_chunk)[:, -(kv_len-sink):]
// re-appends it, clones the WHOLE draft cache, rotates the clone's keys with
// cache-relative positions and runs attention on the clone.  Here:
//   md_streaming_shift_append : the same resulting bytes, in place (each lane
//       owns one 16-B column of the request's rows and walks the slots in
//       ascending order, so a row is always read before it is overwritten);
//   md_streaming_rotate       : rows [0,valid_len) of every request, K rotated
//       by position = slot index, V copied, into a scratch cache (or in place
//       on the last chunk == the reference's kv_cache.copy_(rotated_kv)).
// Bug-for-bug: the shift covers the whole [sink,kv_len) slot range even while
// the tail slot has never been written (the zero row of the 2nd->3rd chunk
// transition travels exactly as in the reference).
#include "md_common.h"

namespace {

__device__ __forceinline__ int64_t slot_offset(int b, int s, int ppr, int page_size, int row_elems) {
    const int page = b * ppr + s / page_size;
    const int slot = s % page_size;
    return ((int64_t)page * 2 * page_size + slot) * row_elems;
}

// grid (ceil(2*cols/256), B): thread -> (k|v, 16-B column) of request b
__global__ __launch_bounds__(256) void shift_append_kernel(const bf16_t* __restrict__ k_new,
                                                           const bf16_t* __restrict__ v_new, int64_t ks, int64_t vs,
                                                           bf16_t* cache, int n_new, int kv_len, int sink, int ppr,
                                                           int KH, int D, int page_size) {
    const int b = blockIdx.y;
    const int row_elems = KH * D;
    const int cols = row_elems / 8;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * cols) return;
    const bool isv = t >= cols;
    const int c = isv ? t - cols : t;
    const int64_t half = isv ? (int64_t)page_size * row_elems : 0;
    const bf16_t* nw = isv ? v_new : k_new;
    const int64_t ns = isv ? vs : ks;
    constexpr int U = 8;
    for (int s0 = sink; s0 < kv_len; s0 += U) {
        u32x4 buf[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int s = s0 + u;
            if (s < kv_len) {
                const int src = s + n_new;
                if (src < kv_len)
                    buf[u] = *reinterpret_cast<const u32x4*>(cache + slot_offset(b, src, ppr, page_size, row_elems) + half + c * 8);
                else
                    buf[u] = *reinterpret_cast<const u32x4*>(nw + (int64_t)(b * n_new + (src - kv_len)) * ns + c * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int s = s0 + u;
            if (s < kv_len)
                *reinterpret_cast<u32x4*>(cache + slot_offset(b, s, ppr, page_size, row_elems) + half + c * 8) = buf[u];
        }
    }
}

// grid (valid_len, B); K rotated by pos = slot (same arithmetic as kvops.hip rope8), V copied
__global__ __launch_bounds__(256) void rotate_kernel(const bf16_t* __restrict__ cache, bf16_t* rot, int ppr, int KH, int D,
                                                     int page_size, const float* __restrict__ cos_sin, int max_pos) {
    const int b = blockIdx.y, s = blockIdx.x;
    const int row_elems = KH * D;
    const int cols = row_elems / 8;
    const int cpr = D / 8;
    const int64_t off = slot_offset(b, s, ppr, page_size, row_elems);
    const int64_t half = (int64_t)page_size * row_elems;
    const int pos = s < max_pos ? s : max_pos - 1;
    const float* cs = cos_sin + (int64_t)pos * D;
    const bool inplace = (rot == cache);
    for (int i = threadIdx.x; i < 2 * cols; i += blockDim.x) {
        if (i < cols) {
            const u32x4 x = *reinterpret_cast<const u32x4*>(cache + off + i * 8);
            const float* c = cs + (i % cpr) * 8;
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(c);
            const f32x4 c1 = *reinterpret_cast<const f32x4*>(c + 4);
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
            *reinterpret_cast<u32x4*>(rot + off + i * 8) = r;
        } else if (!inplace) {
            const int ii = i - cols;
            *reinterpret_cast<u32x4*>(rot + off + half + ii * 8) =
                *reinterpret_cast<const u32x4*>(cache + off + half + ii * 8);
        }
    }
}

}  // namespace

extern "C" int md_streaming_shift_append(const void* k_new, const void* v_new, int64_t k_row_stride,
                                         int64_t v_row_stride, void* cache, int B, int n_new, int kv_len, int sink,
                                         int pages_per_req, int KH, int D, int page_size, md_stream_t stream) {
    MD_CHECK_ARG(k_new && v_new && cache, "md_streaming_shift_append: null pointer argument");
    MD_CHECK_ARG(B > 0 && n_new > 0 && kv_len > sink && sink >= 0 && KH > 0 && D > 0 && (KH * D) % 8 == 0 &&
                     page_size > 0 && pages_per_req * page_size >= kv_len,
                 "md_streaming_shift_append: bad shape n_new=%d kv_len=%d sink=%d pages_per_req=%d", n_new, kv_len,
                 sink, pages_per_req);
    MD_CHECK_ARG((((uintptr_t)k_new | (uintptr_t)v_new | (uintptr_t)cache) & 15) == 0 && k_row_stride % 8 == 0 &&
                     v_row_stride % 8 == 0,
                 "md_streaming_shift_append: 16-byte alignment / stride multiple of 8 required");
    const int cols2 = 2 * KH * D / 8;
    hipLaunchKernelGGL(shift_append_kernel, dim3((cols2 + 255) / 256, B), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)k_new, (const bf16_t*)v_new, k_row_stride, v_row_stride, (bf16_t*)cache, n_new,
                       kv_len, sink, pages_per_req, KH, D, page_size);
    MD_CHECK_LAUNCH("md_streaming_shift_append");
    return MD_OK;
}

extern "C" int md_streaming_rotate(const void* cache, void* rot_cache, int B, int valid_len, int pages_per_req, int KH,
                                   int D, int page_size, const float* cos_sin, int max_pos, md_stream_t stream) {
    MD_CHECK_ARG(cache && rot_cache && cos_sin, "md_streaming_rotate: null pointer argument");
    MD_CHECK_ARG(B > 0 && valid_len > 0 && KH > 0 && D > 0 && D % 8 == 0 && page_size > 0 &&
                     pages_per_req * page_size >= valid_len && max_pos > 0,
                 "md_streaming_rotate: bad shape valid_len=%d pages_per_req=%d", valid_len, pages_per_req);
    MD_CHECK_ARG((((uintptr_t)cache | (uintptr_t)rot_cache | (uintptr_t)cos_sin) & 15) == 0,
                 "md_streaming_rotate: 16-byte alignment required");
    hipLaunchKernelGGL(rotate_kernel, dim3(valid_len, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)cache,
                       (bf16_t*)rot_cache, pages_per_req, KH, D, page_size, cos_sin, max_pos);
    MD_CHECK_LAUNCH("md_streaming_rotate");
    return MD_OK;
}
