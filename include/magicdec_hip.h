/*
 * magicdec_hip.h -- C ABI of libmagicdec_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary for MagicDec's speculative draft/verify decode
 * path.  In the reference every device kernel on that path sits behind seven
 * `torch.library` operators ("mylib::*") whose bodies call flashinfer, plus a
 * handful of ATen ops (RMSNorm, SiLU*mul, argmax, SnapKV select, StreamingLLM
 * evict, the accept/rollback loop).  Each entry point below names the
 * reference interface it replaces (file:line under the reference checkout).
 *
 * Conventions
 *   - plain C, no C++/torch types; every pointer is a DEVICE pointer unless
 *     the parameter name ends in `_host`;
 *   - bf16 tensors are passed as `const void*` / `void*` (2-byte elements);
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*);
 *     nothing synchronises, allocates or frees device memory (except the
 *     md_ar_* set-up / tear-down calls);
 *   - return value: MD_OK (0) or a negative MD_ERR_* code; the message for
 *     the calling thread's last error is md_last_error_string();
 *   - page tables use the reference's (flashinfer 0.1.x) triple
 *     (indices, indptr, last_page_len), int32 on device, NHD page layout
 *     cache[page][2][page_size][KH][D] (Engine/SnapKV/model.py:84).
 *   - sequence length of request b:
 *        len_b = (indptr[b+1]-indptr[b]-1)*page_size + last_page_len[b]
 *   - safe to capture into a hipGraph: no host reads of device memory, all
 *     lengths are read on the device at execution time.
 *   - no process-global tuning state: the development knobs (md_debug_set_*) are declared in
 *     magicdec_hip_dev.h and exist only in a library built with -DMD_DEV_KNOBS.
 */
#ifndef MAGICDEC_HIP_H
#define MAGICDEC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MD_OK 0
#define MD_ERR_INVALID_ARG (-1)
#define MD_ERR_UNSUPPORTED (-2)
#define MD_ERR_LAUNCH (-3)
#define MD_ERR_WORKSPACE (-4)

typedef void* md_stream_t; /* hipStream_t */

/* KV-cache element type.  MD_KV_FP8_E4M3 = OCP e4m3fn bytes (CDNA4 native, NOT the MI300 fnuz format) with one
 * fp32 dequantisation scale per kv head for K and for V (value = byte * scale); the reference has no fp8 path
 * (row "next" 8f-2 of SURVEY.md, BASELINE configs[4]); accuracy is gated in tests/test_gpu_fp8.py. */
#define MD_KV_BF16 0
#define MD_KV_FP8_E4M3 1
/* Page layout flag, OR-ed into the kv_dtype argument of md_append_paged_kv / md_rope_append (first cache) /
 * md_paged_attn / md_snapkv_select (source cache).  Default (flag absent) is the reference's NHD page
 * cache[page][2][page_size][KH][D]; MD_KV_LAYOUT_HND selects cache[page][2][KH][page_size][D] (flashinfer's other
 * layout): the rows of one kv head are contiguous, so a (request, kv head) stream is read in 128-row runs instead of
 * D*sizeof-byte pieces at a KH*D*sizeof stride -- at KH = 8 that is 128-byte pieces for an fp8 cache, whose loads-only
 * ceiling is 72.6 % of the HBM peak.  A second cache (md_rope_append), the SnapKV draft cache
 * and the StreamingLLM ring are always NHD.  Results are bit-identical between the layouts. */
#define MD_KV_LAYOUT_HND 0x100
#define MD_KV_DTYPE_MASK 0xff

/* ABI version (bumped on any signature change). */
int md_abi_version(void);
/* Message of the calling thread's most recent error ("" if none). Host. */
const char* md_last_error_string(void);
/* Returns and clears the calling thread's sticky HIP runtime error (hipGetLastError) -- used by the hipGraph capture
 * fall-back (magicdec_amd/Engine/graph.py): an invalidated capture otherwise fails the next, unrelated launch. Host. */
int md_clear_last_hip_error(void);

/* ------------------------------------------------------------------------
 * K4  mylib::update_kv  ->  flashinfer.append_paged_kv_cache
 *     reference: Engine/utils.py:31-54, callers Engine/SnapKV/model.py:90-112
 * Row j of request b (n_b = append_indptr[b+1]-append_indptr[b] rows) is
 * written to position len_b - n_b + j of the request (the page table must
 * already include the appended rows).  k/v: [nnz, KH, D] with a row stride in
 * elements (a slice of the fused wqkv output can be passed without a copy).
 * n_max >= max_b n_b (host upper bound, sizes the grid).
 * ---------------------------------------------------------------------- */
int md_append_paged_kv(const void* k, const void* v, int64_t k_row_stride, int64_t v_row_stride,
                       const int32_t* append_indptr, void* cache, const int32_t* page_indices,
                       const int32_t* page_indptr, const int32_t* last_page_len, int B, int n_max,
                       int KH, int D, int page_size, int kv_dtype, const float* k_scale,
                       const float* v_scale, md_stream_t stream);

/* Number of KV rows the append kernels (md_append_paged_kv, md_rope_append, md_snapkv_select's gather) dropped
 * since the last reset because their position lay beyond the request's mapped pages (last_page_len > page_size:
 * the reference's page tables never grow during decode, Engine/SnapKV/backend.py:147, and flashinfer would have
 * written into the next request's page).  Host call, synchronises with the device; `reset` != 0 zeroes it. */
int md_page_overflow_count(unsigned int* count_host, int reset);

/* ------------------------------------------------------------------------
 * K5  mylib::rope / mylib::draft_rope -> flashinfer.rope.apply_rope /
 *     apply_llama31_rope (interleave=True)
 *     reference: Engine/SnapKV/model.py:133-156
 * Token j of request b has position offsets[b]+j; the pair (x[2i],x[2i+1]) is
 * rotated by angle pos*freq[i].  `cos_sin` is a host-precomputed device table
 * float32 [max_pos][D/2][2] (cos, sin) -- see md_rope_fill_table_host for the
 * frequency definition.  Positions are clamped to [0, max_pos).  q_out/k_out are contiguous
 * [nnz,H,D] / [nnz,KH,D]; q/k carry a row stride in elements.  k/k_out may
 * be NULL (rotate q only).
 * ---------------------------------------------------------------------- */
int md_rope(const void* q, const void* k, int64_t q_row_stride, int64_t k_row_stride, void* q_out,
            void* k_out, const int32_t* indptr, const int32_t* offsets, int B, int n_max, int H,
            int KH, int D, const float* cos_sin, int max_pos, md_stream_t stream);

/* Host helper: fills table[max_pos][D/2][2] with cos/sin of pos*freq[i],
 * freq[i] = theta^(-2i/D); plain rope: freq/rope_scale; Llama-3.1 smoothing
 * when low_freq_factor>0 (Engine/SnapKV/model.py:140: rope_scale,
 * low/high_freq_factor, old_context_len).  Angles and trig in float64,
 * rounded once to float32. */
int md_rope_fill_table_host(float* table_host, int max_pos, int D, double theta, double rope_scale,
                            double low_freq_factor, double high_freq_factor, double old_context_len);

/* Fused K5+K4 for the decode/verify/prefill step (our Engine's fast path;
 * same results as md_rope followed by md_append_paged_kv on the rotated k):
 * reads q,k,v rows of the fused wqkv output, writes rotated q to q_out
 * (contiguous) and rotated k plus v straight into the pages.  When
 * cache2 != NULL the same rows are also appended to a second cache with its
 * own page table (self-spec verify writes target and draft caches,
 * Engine/SnapKV/model.py:347-348).  kv_dtype applies to `cache`; `cache2` is always bf16. */
int md_rope_append(const void* q, const void* k, const void* v, int64_t q_row_stride,
                   int64_t k_row_stride, int64_t v_row_stride, void* q_out,
                   const int32_t* indptr, const int32_t* offsets, int B, int n_max, int H, int KH,
                   int D, const float* cos_sin, int max_pos, void* cache,
                   const int32_t* page_indices, const int32_t* page_indptr,
                   const int32_t* last_page_len, void* cache2, const int32_t* page_indices2,
                   const int32_t* page_indptr2, const int32_t* last_page_len2, int page_size,
                   int kv_dtype, const float* k_scale, const float* v_scale, md_stream_t stream);

/* ------------------------------------------------------------------------
 * K1/K2/K3  mylib::target_decode / target_prefill / draft_decode /
 *     draft_prefill -> flashinfer BatchPrefillWithPagedKVCacheWrapper.plan+run
 *     reference: Engine/SnapKV/backend.py:56-107,146-159; backend_draft.py:42-92
 * q: [nnz, H, D] (row stride in elements), out: [nnz, H, D] contiguous.
 * Query row i of request b (m_b rows) attends kv positions <= len_b-m_b+i
 * (causal) or all len_b positions; q head h reads kv head h/(H/KH);
 * softmax(q.k * sm_scale) in fp32; bf16 in/out.  The kernel variant
 * (verify / draft / chunked prefill) is chosen from (n_max, H/KH).
 * max_pages_per_req: host upper bound of pages of any request (sizes the
 * split-KV decomposition; no host read of the page table is made).
 * workspace: md_paged_attn_workspace_bytes() bytes of scratch.
 * ---------------------------------------------------------------------- */
size_t md_paged_attn_workspace_bytes(int B, int n_max, int H, int KH, int D,
                                     int max_pages_per_req, int page_size);
/* measurement (bench.py's roofline): while enabled, every decode / verify launch of md_paged_attn with n_max ==
 * n_rows query rows per request is bracketed by the kernel's OWN begin / end timestamps (hipExtLaunchKernel start /
 * stop events = what a rocprofv3 kernel trace reports; stream events around a launch also see the dispatch overhead).
 * md_debug_attn_timing_read synchronises the device, writes the durations in ms (launch order, at most `cap`) and
 * returns their count; the list is cleared.  Eager launches only (not inside a graph capture). */
void md_debug_attn_timing(int enable, int n_rows);
int md_debug_attn_timing_read(float* ms_out, int cap);
int md_paged_attn(const void* q, int64_t q_row_stride, const void* cache, void* out,
                  const int32_t* qo_indptr, const int32_t* page_indices,
                  const int32_t* page_indptr, const int32_t* last_page_len, int B, int n_max,
                  int H, int KH, int D, int page_size, int causal, float sm_scale,
                  int max_pages_per_req, int kv_dtype, const float* k_scale, const float* v_scale,
                  void* workspace, size_t workspace_bytes, md_stream_t stream);

/* ------------------------------------------------------------------------
 * K6  Attention.gen_draft_kv  (SnapKV select)
 *     reference: Engine/SnapKV/model.py:389-439 (+ caller :381-382)
 * q_win: rotated queries of the last chunk [B*window, H, D] (contiguous);
 * cache/page table: the request's full KV (ctx_len tokens each, same for all
 * requests as in the reference: context_len = offsets[0]+seqlen).
 * Reproduces the reference's rounding sequence (bf16 scores, fp32 softmax ->
 * bf16, bf16 partial sums in chunk order, avg_pool1d k=kernel, group sum,
 * top-(budget-window) in descending-score order) and appends, per request
 * and kv head, the selected rows followed by the last `window` rows to the
 * draft cache at positions [draft_len_b - budget, draft_len_b).
 * idx_out: [B, KH, budget-window] int32 (selected positions, reference order).
 * kv_dtype describes `cache`; the draft cache is bf16 (fp8 rows are dequantised while gathering).
 * ---------------------------------------------------------------------- */
size_t md_snapkv_workspace_bytes(int B, int H, int KH, int ctx_len, int window);
/* byte offset, inside the (256-byte aligned) workspace, of the pooled group
 * scores bf16 [B, KH, ctx_len-window] the top-k ran on (valid after the call;
 * exposed so tests can compare scores, not only indices). */
size_t md_snapkv_scores_offset(int B, int H, int KH, int ctx_len, int window);
int md_snapkv_select(const void* q_win, const void* cache, const int32_t* page_indices,
                     const int32_t* page_indptr, int B, int H, int KH, int D, int page_size,
                     int ctx_len, int window, int budget, int pool_kernel, void* draft_cache,
                     const int32_t* draft_page_indices, const int32_t* draft_page_indptr,
                     const int32_t* draft_last_page_len, int32_t* idx_out, int kv_dtype,
                     const float* k_scale, const float* v_scale, void* workspace,
                     size_t workspace_bytes, md_stream_t stream);

/* ------------------------------------------------------------------------
 * K7  KVCache.prefill / prefill_draft  (StreamingLLM sink+window eviction)
 *     reference: Engine/StreamingLLM/model_draft.py:102-143, model.py:116-157
 * In-place equivalent of: keep rows [0,sink) and the most recent
 * (kv_len-sink) of (old rows [sink,kv_len) ++ the n new rows) in the
 * request's contiguous slot range [0,kv_len).  `cache` holds UN-rotated keys.
 * rot_cache (same shape) receives, for every request, rows [0,valid_len)
 * with K rotated to cache-relative position = slot index and V copied
 * (the reference's per-chunk rotated clone); pass rot_cache == cache on the
 * last chunk to reproduce `self.kv_cache.copy_(rotated_kv)`.
 * Requests own pages_per_req consecutive pages starting at b*pages_per_req
 * (Engine/StreamingLLM/backend_draft.py:160).
 * ---------------------------------------------------------------------- */
int md_streaming_shift_append(const void* k_new, const void* v_new, int64_t k_row_stride,
                              int64_t v_row_stride, void* cache, int B, int n_new, int kv_len,
                              int sink, int pages_per_req, int KH, int D, int page_size,
                              md_stream_t stream);
int md_streaming_rotate(const void* cache, void* rot_cache, int B, int valid_len,
                        int pages_per_req, int KH, int D, int page_size, const float* cos_sin,
                        int max_pos, md_stream_t stream);

/* ------------------------------------------------------------------------
 * K9  RMSNorm / SiLU*mul / residual   reference: Engine/SnapKV/model.py:451-469
 * rmsnorm: y = bf16( x_f32 * rsqrt(mean(x^2)+eps) ) * w   (weight multiply in
 * bf16, model.py:467-469).  add_rmsnorm: h = x + r (bf16 add), then the same
 * norm of h; writes h_out and y.  silu_mul: y = bf16(silu(a)) * b with the
 * reference's bf16 rounding points (F.silu in bf16, product in bf16).
 * ---------------------------------------------------------------------- */
int md_rmsnorm(const void* x, const void* weight, void* y, int rows, int dim, float eps,
               md_stream_t stream);
int md_add_rmsnorm(const void* x, const void* r, const void* weight, void* h_out, void* y,
                   int rows, int dim, float eps, md_stream_t stream);
int md_silu_mul(const void* a, const void* b, int64_t a_row_stride, int64_t b_row_stride, void* y,
                int rows, int dim, md_stream_t stream);

/* ------------------------------------------------------------------------
 * K8  the decode / verify linears: out[M][N] = x[M][K] . W[N][K]^T  (+ epilogue), M <= 256
 *     reference: nn.Linear calls of a decode step -- Attention.wqkv / wo, FeedForward.w1 / w3 / w2, Transformer.output
 *     (Engine/SnapKV/model.py:288-289,446-455,175-177) and WeightOnlyInt8Linear.forward (Engine/quantize.py:84-86)
 * Weight-streaming skinny GEMM (csrc/gemm.hip): W rows go global -> registers in MFMA operand layout, activations
 * through LDS, fp32 accumulation, split-K with a fixed-order combine (deterministic).
 *   w_dtype  MD_W_BF16: w is bf16 [N][K];  MD_W_INT8: w is int8 [N][K] with bf16 per-row `scales` [N]
 *            (out = bf16(bf16(x.w^T) * scale), the reference's rounding points)
 *   epilogue MD_EPI_NONE: out[M][N] = bf16(acc + bias)   (bias bf16 [N] or NULL)
 *            MD_EPI_SWIGLU: w = [w1; w3] (N = 2*I rows): out[M][I] = bf16(bf16(silu(bf16(h1))) * bf16(h3))
 *   x row stride ldx, out row stride ldo (elements).  workspace: md_linear_workspace_bytes() bytes (16-B aligned).
 *   w_packed 0: w is the nn.Linear row-major [N][K] tensor.
 *            1: w is the STREAMING layout [ceil(N/32)][K/16][64][8]: element (t, s, lane, e) =
 *               W[32*t + lane%32][16*s + 8*(lane/32) + e] (rows >= N zero) -- every wavefront load instruction is one
 *               contiguous KiB and every wavefront reads one sequential stream; for MD_EPI_SWIGLU tile t holds rows
 *               16t..16t+15 of w1 followed by rows 16t..16t+15 of w3 ([ceil(I/16)] tiles).  The row-major form makes
 *               32-byte pieces of 32 different rows per instruction and streams at about half the rate.
 * md_linear_supported() == 0 -> use a library GEMM (M > 256, K % 128 != 0, ...).
 * ---------------------------------------------------------------------- */
#define MD_W_BF16 0
#define MD_W_INT8 1
#define MD_EPI_NONE 0
#define MD_EPI_SWIGLU 1
int md_linear_supported(int M, int N, int K, int epilogue);
size_t md_linear_workspace_bytes(int M, int N, int K, int epilogue);
int md_linear(const void* x, int64_t ldx, const void* w, int w_dtype, int w_packed, const void* scales,
              const void* bias, void* out, int64_t ldo, int M, int N, int K, int epilogue, void* workspace,
              size_t workspace_bytes, md_stream_t stream);
/* md_linear_normed: md_linear over rmsnorm(h) * norm_weight WITHOUT a launch for the norm -- the consumer half of the
 * deferred RMSNorm of md_linear_fused on the weight-streaming kernel (Engine/SnapKV/model.py:260-278,464-469: the
 * ffn_norm in front of w1|w3, the final norm in front of the lm head).  h [M][K] is the un-normalised hidden state the
 * residual epilogue of the producing linear wrote (MD_FL_RESID with ssq_out), ssq [M][ssq_tiles] its partial sums of
 * squares (ssq_tiles * 32 == K); every workgroup forms rstd per row (the fixed order of md_linear_fused) and feeds
 * y = bf16(bf16(h * rstd) * norm_weight) into the product.  bf16 weights; shapes / workspace / epilogues as md_linear. */
int md_linear_normed(const void* h, int64_t ldh, const float* ssq, int ssq_tiles, const void* norm_weight, float eps,
                     const void* w, int w_packed, const void* bias, void* out, int64_t ldo, int M, int N, int K,
                     int epilogue, void* workspace, size_t workspace_bytes, md_stream_t stream);

/* md_linear_add_rmsnorm: the output projection of a sub-layer on the weight-streaming kernel TOGETHER with what follows
 * it (Engine/SnapKV/model.py:260-278: h = x + wo(...), out = h + w2(...); :464-469 RMSNorm): the split-K combine launch
 * of md_linear also adds the residual and normalises --
 *   o = bf16(x.W^T + bias) [int8: bf16(bf16(x.W^T) * scale)];  h_out[M][N] = bf16(resid + o);  y_out = rmsnorm(h) * w
 * bit-identical to md_linear followed by md_add_rmsnorm, one launch and one pass over h fewer.  Needs a shape whose K
 * md_linear splits (md_linear_add_rmsnorm_supported), N % 8 == 0, N <= 8192; workspace as md_linear. */
int md_linear_add_rmsnorm_supported(int M, int N, int K);
int md_linear_add_rmsnorm(const void* x, int64_t ldx, const void* w, int w_dtype, int w_packed, const void* scales,
                          const void* bias, const void* resid, int64_t ldr, const void* norm_weight, float eps,
                          void* h_out, void* y_out, int M, int N, int K, void* workspace, size_t workspace_bytes,
                          md_stream_t stream);

/* ------------------------------------------------------------------------
 * K8c  block-tile GEMM for the 129..256-row linears of a verify step (csrc/blockgemm.hip)
 *     reference: the same nn.Linear calls as K8 at M = B x (gamma+1) = 256 rows -- Attention.wqkv / wo,
 *     FeedForward.w1 / w3 / w2, Transformer.output (Engine/SnapKV/model.py:288-289,446-455,175-177)
 * out = epilogue(x[M][K] . W[N][K]^T + bias), W bf16 in the STREAMING layout of md_linear (w_packed = 1; packed for
 * MD_EPI_SWIGLU when that is the epilogue), M <= 256, N % 128 == 0, K % 64 == 0.  One workgroup = all rows x 128 columns
 * x a K slice, both operands global -> LDS by DMA through a three-stage ring, v_mfma_f32_32x32x16_bf16; narrow products
 * split K over workgroups (fp32 partial tiles in `workspace`, combined in slice order by the launch that applies the
 * epilogue: deterministic).  Epilogues and rounding points are md_linear's (MD_EPI_NONE / MD_EPI_SWIGLU);
 * md_linear_block_add_rmsnorm is md_linear_add_rmsnorm on this kernel ((h, y) = (resid + o, rmsnorm(h) * w): the combine
 * launch of md_linear_add_rmsnorm, same rounding points).  The narrow projections of an M = 256 step (wqkv, wo) stay on
 * md_linear_fused / the library: a K split over ~256 workgroups costs them more than it saves.
 * workspace: md_linear_block_workspace_bytes(M, N, K, force_split) bytes, 16-B aligned (force_split = 1 for the
 * add_rmsnorm form, whose epilogue always runs in the combine launch). */
int md_linear_block_supported(int M, int N, int K, int epilogue);
size_t md_linear_block_workspace_bytes(int M, int N, int K, int force_split);
int md_linear_block(const void* x, int64_t ldx, const void* w_packed, const void* bias, void* out, int64_t ldo, int M,
                    int N, int K, int epilogue, void* workspace, size_t workspace_bytes, md_stream_t stream);
int md_linear_block_add_rmsnorm(const void* x, int64_t ldx, const void* w_packed, const void* bias, const void* resid,
                                int64_t ldr, const void* norm_weight, float eps, void* h_out, void* y_out, int M, int N,
                                int K, void* workspace, size_t workspace_bytes, md_stream_t stream);

/* ------------------------------------------------------------------------
 * K8b  fused small-problem linear: a linear of a decode / verify step TOGETHER with the op that consumes its output,
 *      one launch (csrc/tilegemm.hip; no split-K across workgroups, no workspace, deterministic)
 *     reference: Engine/SnapKV/model.py:322-336 (wqkv -> apply_rope -> update_kv), :260-278 (h = x + attention(...),
 *                out = h + feed_forward(...)), :451-455 (w2(silu(w1 x) * w3 x))
 * out = epilogue(x[M][K] . W[N][K]^T + bias), W in the streaming layout of md_linear (w_packed = 1, bf16), M <= 256,
 * K % 128 == 0, N % 32 == 0.  Epilogues (the reference's rounding points: the linear's output is rounded to bf16 first):
 *   MD_FL_NONE         out[M][N]   = bf16(acc + bias)
 *   MD_FL_SWIGLU       out[M][N/2] = bf16(bf16(silu(bf16(h1))) * bf16(h3)); W packed for MD_EPI_SWIGLU, no bias
 *   MD_FL_RESID        out[M][N]   = bf16(resid + bf16(acc + bias))                 (the bf16 residual add)
 *   MD_FL_ROPE_APPEND  N = (H + 2 KH) * D fused qkv rows [q; k; v]: q columns -> interleaved RoPE -> out = q_rot
 *                      [M][H*D] (contiguous); k columns -> RoPE -> paged cache; v columns -> paged cache.  Request b
 *                      owns rows [b * rows_per_req, (b+1) * rows_per_req) (uniform append, what a decode step has);
 *                      row j of request b has RoPE position offsets[b] + j and cache position len_b - rows_per_req + j
 *                      (page table already advanced, as md_rope_append); kv_dtype / scales / cache2 as md_rope_append.
 *                      Results are bit-identical to md_linear_fused(MD_FL_NONE) followed by md_rope_append.
 * Deferred RMSNorm: see the last five fields of md_fused_linear_args.
 * ---------------------------------------------------------------------- */
#define MD_FL_NONE 0
#define MD_FL_SWIGLU 1
#define MD_FL_RESID 2
#define MD_FL_ROPE_APPEND 3
typedef struct md_fused_linear_args {
    const void* x;            /* [M][K] bf16, row stride ldx (elements) */
    const void* w_packed;     /* streaming layout (md_linear, w_packed = 1) */
    const void* bias;         /* bf16 [N] or NULL */
    void* out;                /* row stride ldo */
    const void* resid;        /* MD_FL_RESID: bf16 [M][N], row stride ldr */
    int64_t ldx, ldo, ldr;
    int M, N, K, epilogue;
    /* MD_FL_ROPE_APPEND */
    int H, KH, D, rows_per_req, max_pos, page_size, kv_dtype;
    const int32_t* offsets;   /* [M / rows_per_req] RoPE position of each request's first row */
    const float* cos_sin;     /* [max_pos][D/2][2] fp32 table (md_rope_fill_table_host) */
    void* cache;
    const int32_t* page_indices;
    const int32_t* page_indptr;
    const int32_t* last_page_len;
    void* cache2;             /* optional second cache (bf16, NHD) or NULL */
    const int32_t* page_indices2;
    const int32_t* page_indptr2;
    const int32_t* last_page_len2;
    const float* k_scale;     /* fp8 pages: per-kv-head scales */
    const float* v_scale;
    /* deferred RMSNorm (both optional).  MD_FL_RESID with ssq_out: also writes ssq_out[M][N/32] = per row, the sum of
     * squares of each 32-column tile of `out`.  MD_FL_SWIGLU / MD_FL_ROPE_APPEND with pro_ssq: x is the UN-normalised
     * hidden state h [M][K]; the kernel forms rstd = rsqrt(sum_t pro_ssq[row][t] / K + pro_eps) (t < pro_tiles, fixed
     * order) and feeds y = bf16(bf16(h * rstd) * pro_norm_w) into the product -- md_rmsnorm's arithmetic, only the
     * order of the fp32 sum of squares differs. */
    float* ssq_out;
    const float* pro_ssq;
    const void* pro_norm_w;   /* bf16 [K] */
    float pro_eps;
    int pro_tiles;
} md_fused_linear_args;
int md_linear_fused_supported(int M, int N, int K, int epilogue);
int md_linear_fused(const md_fused_linear_args* args, md_stream_t stream);

/* md_linear_fused_split (round 6): the K8b tile kernel with the K range ALSO split over S workgroups, for the deep narrow
 * products of a <= 64-row draft step (the 1B model's w2: N = 2048, K = 8192 -- Engine/SnapKV/model.py:451-455 followed by
 * the residual add and the next RMSNorm, :260-278,464-469).  A workgroup owns a (32|64)-row x (32|64)-column tile and K / S
 * of the depth, its 8 wavefronts K / (8 S) each -- at the 1B w2 that is 128 KB of W + 128 KB of x per CU, all requested
 * by the first instructions of the launch; fp32 partial planes [S][M][N] go to `workspace` and a second launch adds them
 * IN SLICE ORDER (deterministic) and applies the epilogue -- the combine launches of md_linear:
 *   md_linear_fused_split              out = bf16(sum_s partial_s + bias)
 *   md_linear_fused_split_add_rmsnorm  o as above; h_out = bf16(resid + o); y_out = rmsnorm(h) * norm_weight
 *                                      -- bit-identical to md_linear_fused_split followed by md_add_rmsnorm.
 * Shapes as md_linear_fused (M <= 256, K % 128 == 0, N % 32 == 0; bf16 weights in the streaming layout, not packed for
 * SwiGLU); the add_rmsnorm form needs N % 8 == 0, N <= 8192.  workspace: md_linear_fused_split_workspace_bytes() bytes,
 * 16-byte aligned. */
size_t md_linear_fused_split_workspace_bytes(int M, int N, int K);
int md_linear_fused_split(const void* x, int64_t ldx, const void* w_packed, const void* bias, void* out, int64_t ldo,
                          int M, int N, int K, void* workspace, size_t workspace_bytes, md_stream_t stream);
int md_linear_fused_split_add_rmsnorm(const void* x, int64_t ldx, const void* w_packed, const void* bias,
                                      const void* resid, int64_t ldr, const void* norm_weight, float eps, void* h_out,
                                      void* y_out, int M, int N, int K, void* workspace, size_t workspace_bytes,
                                      md_stream_t stream);

/* ------------------------------------------------------------------------
 * K10  argmax over a vocab shard / TP merge
 *     reference: Engine/SnapKV/model.py:175-188
 * argmax: per row, max bf16 logit and its lowest index (+index_offset).
 * argmax_tp_slots (round 5): the same argmax written as the one-hot-slot
 * tensors the reference all-reduces (model.py:178-184: torch.zeros + index
 * assignment): vals_out / idx_out [rows, tp_world], the row's result in
 * column tp_rank, zeros elsewhere -- one launch instead of five.
 * tp_argmax_merge: vals/idx [rows, tp] -> token of the lowest rank among
 * equal maxima (torch.argmax over [..,tp], model.py:185).
 * ---------------------------------------------------------------------- */
int md_argmax(const void* logits, int64_t row_stride, int rows, int vocab, int64_t index_offset,
              void* max_val_out /* bf16 [rows], may be NULL */, int64_t* idx_out,
              md_stream_t stream);
int md_argmax_tp_slots(const void* logits, int64_t row_stride, int rows, int vocab, int64_t index_offset,
                       int tp_rank, int tp_world, void* vals_out /* bf16 [rows,tp_world] */,
                       int64_t* idx_out /* [rows,tp_world] */, md_stream_t stream);
int md_tp_argmax_merge(const void* vals /* bf16 [rows,tp] */, const int64_t* idx /* [rows,tp] */,
                       int rows, int tp, int64_t* out, md_stream_t stream);

/* ------------------------------------------------------------------------
 * a1  the verify loop body (accept / rollback / scatter / bonus / terminate)
 *     reference: tests/SnapKV/longspec_benchmark.py:208-281 (= StreamingLLM
 *     twin :205-278).  One launch, all-integer, no host sync.
 * Inputs: tokens_buffer [B,gamma+1], target_tokens [B,gamma+1] (int64).
 * In/out: output [B,out_cols] int64, num_nodes [B] int64, target
 *   cachelens/last_page_len [B] int32 (already advanced by gamma+1 by the
 *   verify call), draft cachelens/last_page_len [B] int32 (already advanced
 *   by gamma; may be NULL), draft_rollback = gamma (longspec) or gamma+1
 *   (selfspec), draft_cap = max tokens the draft keeps (gamma for longspec:
 *   min(accept,gamma); gamma+1 for selfspec).
 * Outputs: accept_nums [B] int64, bonus [B] int64, double_buffer [B,2]
 *   int64, cachelens_update [B] int64, flags[0]=terminal, flags[1]=any row
 *   accepted all gamma (next_double).  On terminal the bonus token is also
 *   written to output[b, num_nodes[b]] and num_nodes += 1 (:283-285); else
 *   tokens_buffer[:,0] = bonus.
 * ---------------------------------------------------------------------- */
int md_accept_rollback(int64_t* tokens_buffer, const int64_t* target_tokens, int64_t* output,
                       int out_cols, int64_t* num_nodes, int32_t* cachelens,
                       int32_t* last_page_len, int32_t* draft_cachelens,
                       int32_t* draft_last_page_len, int B, int gamma, int draft_rollback,
                       int draft_cap, int64_t eot_1, int64_t eot_2, int64_t max_nodes,
                       int64_t* accept_nums, int64_t* bonus, int64_t* double_buffer,
                       int64_t* cachelens_update, int32_t* flags, md_stream_t stream);

/* ------------------------------------------------------------------------
 * C1  the per-layer sum-all-reduce of the tensor-parallel decode path over xGMI peer-mapped buffers
 *     reference: dist.all_reduce at Engine/SnapKV/model.py:334-335,453-454 (and the StreamingLLM twins), NCCL with
 *     PyTorch's intra-node one-/two-shot kernels (README.md:59 ENABLE_INTRA_NODE_COMM=1); the fused form also
 *     replaces the residual add + RMSNorm that consumes the result (model.py:260-278,464-469).
 *   md_ar_create      allocates this rank's registered buffer (4 x max_bytes: two halves of published partials, two
 *                     halves of two-shot results) and its signal area;
 *   md_ar_get_handles writes 2 IPC handles (data, signal; MD_AR_HANDLE_BYTES each) to handles_host;
 *                     the caller exchanges them over its bootstrap transport (RCCL / gloo all-gather);
 *   md_ar_open_peers  takes the world x 2 handles in rank order and maps the peers' buffers.
 * md_allreduce: out = sum over ranks of in (count bf16, multiple of 8, count*2 <= max_bytes; in == out allowed).
 *   algo MD_AR_ALGO_ONESHOT: every rank reads all N partials ((N-1) x message inbound, one hop);
 *        MD_AR_ALGO_TWOSHOT: reduce-scatter + all-gather through the registered buffers (2(N-1)/N x message, two hops);
 *        MD_AR_ALGO_AUTO:    two-shot for world >= 4 and messages > 512 KiB, else one-shot.
 * md_allreduce_add_rmsnorm: h = bf16(resid + allreduce(partial)), y = rmsnorm(h) * weight ([rows, dim] bf16, dim <= 8192)
 *   in the same launch (a wavefront owns a row).
 * Every rank adds in rank order with fp32 accumulation and one rounding: all ranks get the same bits.  Calls are
 * asynchronous on `stream`, graph-capturable (call counters live in device memory), and must be issued in the same
 * order with the same sizes on every rank.  Spins are bounded (~2 s): on a time-out the kernel sets the status word
 * (md_ar_status: 0 = ok, 1 = timed out) and writes NaN into the rows it could not complete -- a missing peer is
 * never papered over (two-shot: the rows the timed-out rank owns are NaN in every rank's output); the host reads the
 * status word with every iteration's flag read (md_ar_status_async: a 4-byte copy into pinned host memory queued on
 * `stream`, valid after the stream is synchronised) and raises.
 * md_ar_set_publish: how a rank makes its partial visible to its peers before it raises their flags --
 *   MD_AR_PUBLISH_WRITE_THROUGH (default): 16-byte system-scope write-through stores drained by every wave (fast; rests
 *        on a write-through store being peer-visible when it retires);
 *   MD_AR_PUBLISH_FENCE: the same stores followed by the memory model's system-scope release fence (L2 write-back)
 *        before the flags -- slower per hop, assumes nothing beyond the documented model; the fallback arm a multi-GPU
 *        run selects when the write-through arm fails its bit-exact stress on real links (bench.py collectives_us,
 *        MAGICDEC_AR_PUBLISH=fence).  Every rank of a communicator must use the same mode; takes effect at the next call.
 * ---------------------------------------------------------------------- */
#define MD_AR_HANDLE_BYTES 64
#define MD_AR_MAX_RANKS 8
#define MD_AR_ALGO_AUTO 0
#define MD_AR_ALGO_ONESHOT 1
#define MD_AR_ALGO_TWOSHOT 2
#define MD_AR_PUBLISH_WRITE_THROUGH 0
#define MD_AR_PUBLISH_FENCE 1
typedef struct md_ar_comm md_ar_comm;
int md_ar_create(int rank, int world, size_t max_bytes, md_ar_comm** comm_out);
int md_ar_get_handles(md_ar_comm* comm, void* handles_host /* 2 * MD_AR_HANDLE_BYTES */);
int md_ar_open_peers(md_ar_comm* comm, const void* all_handles_host /* world * 2 * MD_AR_HANDLE_BYTES */);
int md_allreduce(md_ar_comm* comm, const void* in, void* out, size_t count, int algo, md_stream_t stream);
int md_allreduce_oneshot(md_ar_comm* comm, const void* in, void* out, size_t count, md_stream_t stream);
int md_allreduce_add_rmsnorm(md_ar_comm* comm, const void* partial, const void* resid, const void* weight,
                             void* out_h, void* out_y, int rows, int dim, float eps, int algo, md_stream_t stream);
int md_ar_set_publish(md_ar_comm* comm, int mode);
int md_ar_status(md_ar_comm* comm, int* status_host);
int md_ar_status_async(md_ar_comm* comm, int* status_host_pinned, md_stream_t stream);
int md_ar_destroy(md_ar_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* MAGICDEC_HIP_H */
