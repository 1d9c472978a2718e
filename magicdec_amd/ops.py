"""Tensor-level wrappers over the C ABI (include/magicdec_hip.h).

These are the seven ``mylib::*`` operators of the reference (Engine/utils.py:31-66,
Engine/SnapKV/backend.py:56-107, Engine/SnapKV/model.py:133-156) plus the fused
small ops of the decode loop, with the same tensor-level signatures, backed by
hand-written gfx950 kernels.  Differences from the reference's op boundary:

* no hidden ``plan()`` state: the page table is passed to the attention op
  explicitly and read on the device (no host sync, graph-capturable);
* tensors must live on the GPU; there is no CPU implementation here
  (``MagicDecHipError`` if the library is missing, ``ValueError`` for CPU tensors).
"""
from __future__ import annotations

import ctypes
import math

import torch

from . import _lib
from ._lib import MagicDecHipError, check  # noqa: F401

PAGE_SIZE = 128
MD_KV_BF16, MD_KV_FP8_E4M3 = 0, 1      # include/magicdec_hip.h
MD_KV_LAYOUT_HND = 0x100
FP8_DTYPE = torch.float8_e4m3fn
KV_LAYOUTS = ("NHD", "HND")


def _kv_geom(kv_cache, kv_layout):
    """(page_size, KH, D) of a paged cache [pages, 2, page_size, KH, D] ("NHD", the reference's layout:
    Engine/SnapKV/backend.py:30) or [pages, 2, KH, page_size, D] ("HND", rows of a kv head contiguous)."""
    if kv_layout not in KV_LAYOUTS:
        raise ValueError(f"kv_layout must be one of {KV_LAYOUTS}, got {kv_layout!r}")
    if kv_cache.dim() != 5 or kv_cache.shape[1] != 2 or not kv_cache.is_contiguous():
        raise ValueError("paged KV cache must be a contiguous [pages, 2, ., ., D] tensor")
    if kv_layout == "HND":
        return kv_cache.shape[3], kv_cache.shape[2], kv_cache.shape[4]
    return kv_cache.shape[2], kv_cache.shape[3], kv_cache.shape[4]


def _kv_args(kv_cache, kv_scales, kv_layout="NHD"):
    """(kv_dtype | layout flag, k_scale ptr, v_scale ptr) for a paged cache: bf16 caches take no scales, e4m3fn
    caches need per-kv-head float32 dequantisation scales `(k_scale[KH], v_scale[KH])`."""
    flag = MD_KV_LAYOUT_HND if kv_layout == "HND" else 0
    if kv_cache.dtype == torch.bfloat16:
        return MD_KV_BF16 | flag, None, None
    if kv_cache.dtype != FP8_DTYPE:
        raise TypeError(f"paged KV cache must be bfloat16 or float8_e4m3fn, got {kv_cache.dtype}")
    if kv_scales is None:
        raise ValueError("an fp8 KV cache needs kv_scales=(k_scale, v_scale)")
    ks, vs = kv_scales
    KH = _kv_geom(kv_cache, kv_layout)[1]
    for t in (ks, vs):
        if t.dtype != torch.float32 or t.numel() != KH or not t.is_contiguous() or not t.is_cuda:
            raise ValueError("kv scales must be contiguous float32 [KH] tensors on the GPU")
    return MD_KV_FP8_E4M3 | flag, _p(ks), _p(vs)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise ValueError("magicdec_amd.ops: tensors must be on the GPU (no CPU path exists)")


def _i32(t):
    if t.dtype != torch.int32:
        raise TypeError("page-table tensors must be int32")
    return t


def _row_stride(t):
    """[rows, heads, D] view whose last two dims are dense; returns the row stride in elements."""
    if t.dim() != 3 or t.stride(2) != 1 or t.stride(1) != t.shape[2]:
        raise ValueError("expected a [rows, heads, D] tensor with dense (heads, D)")
    return t.stride(0)


# ----------------------------------------------------------------------------- K4
def update_kv(k, v, kv_append_indptr, kv_cache, kv_page_indices, kv_page_indptr, kv_page_lastlen, n_max=None,
              kv_scales=None, kv_layout="NHD"):
    """mylib::update_kv (Engine/utils.py:31-54).  In place on kv_cache."""
    _gpu(k, v, kv_cache)
    B = kv_page_indptr.numel() - 1
    page_size, KH, D = _kv_geom(kv_cache, kv_layout)
    if n_max is None:
        n_max = (k.shape[0] + B - 1) // B if B > 0 else 0
        n_max = max(n_max, 1)
    lib = _lib.load()
    check(lib.md_append_paged_kv(_p(k), _p(v), _row_stride(k), _row_stride(v), _p(_i32(kv_append_indptr)),
                                 _p(kv_cache), _p(_i32(kv_page_indices)), _p(_i32(kv_page_indptr)),
                                 _p(_i32(kv_page_lastlen)), B, n_max, KH, D, page_size,
                                 *_kv_args(kv_cache, kv_scales, kv_layout), _stream()),
          "md_append_paged_kv")


def page_overflow_count(reset=True):
    """KV rows dropped by the append kernels because their request's last page was full (see
    md_page_overflow_count).  Synchronises with the device: call it per batch, not per step."""
    n = ctypes.c_uint(0)
    check(_lib.load().md_page_overflow_count(ctypes.byref(n), 1 if reset else 0), "md_page_overflow_count")
    return int(n.value)


# ----------------------------------------------------------------------------- K5
class RopeTable:
    """Host-precomputed cos/sin table on the device (float32 [max_pos, D/2, 2])."""

    def __init__(self, max_pos, head_dim, rope_theta, rope_scale, low_freq_factor=None, high_freq_factor=None,
                 old_context_len=None, device="cuda"):
        lib = _lib.load()
        host = torch.empty(max_pos, head_dim // 2, 2, dtype=torch.float32)
        llama31 = low_freq_factor is not None and high_freq_factor is not None
        check(lib.md_rope_fill_table_host(ctypes.c_void_p(host.data_ptr()), max_pos, head_dim, float(rope_theta),
                                          float(rope_scale), float(low_freq_factor) if llama31 else 0.0,
                                          float(high_freq_factor) if llama31 else 0.0,
                                          float(old_context_len) if llama31 else 0.0),
              "md_rope_fill_table_host")
        self.table = host.to(device)
        self.max_pos = max_pos
        self.head_dim = head_dim


def rope(q, k, indptr, offsets, table: RopeTable, n_max=None):
    """mylib::rope / mylib::draft_rope (Engine/SnapKV/model.py:133-156): returns new (q, k)."""
    _gpu(q, k)
    B = indptr.numel() - 1
    H, D = q.shape[1], q.shape[2]
    KH = k.shape[1] if k is not None else 0
    if n_max is None:
        n_max = max((q.shape[0] + B - 1) // B, 1)
    q_out = torch.empty((q.shape[0], H, D), dtype=q.dtype, device=q.device)
    k_out = torch.empty((k.shape[0], KH, D), dtype=k.dtype, device=k.device) if k is not None else None
    lib = _lib.load()
    check(lib.md_rope(_p(q), _p(k), _row_stride(q), _row_stride(k) if k is not None else 0, _p(q_out), _p(k_out),
                      _p(_i32(indptr)), _p(_i32(offsets)), B, n_max, H, KH, D, _p(table.table), table.max_pos,
                      _stream()), "md_rope")
    return q_out, k_out


def rope_append(q, k, v, indptr, offsets, table: RopeTable, kv_cache, page_indices, page_indptr, last_page_len,
                kv_cache2=None, page_indices2=None, page_indptr2=None, last_page_len2=None, n_max=None,
                kv_scales=None, kv_layout="NHD"):
    """Fused mylib::rope + mylib::update_kv (+ second cache for self-spec verify).  Returns rotated q.
    kv_cache may be fp8 (with kv_scales) and/or HND (kv_layout); kv_cache2 is always bf16 NHD."""
    _gpu(q, k, v, kv_cache, kv_cache2)
    B = indptr.numel() - 1
    H, D = q.shape[1], q.shape[2]
    KH = k.shape[1]
    if n_max is None:
        n_max = max((q.shape[0] + B - 1) // B, 1)
    q_out = torch.empty((q.shape[0], H, D), dtype=q.dtype, device=q.device)
    lib = _lib.load()
    check(lib.md_rope_append(_p(q), _p(k), _p(v), _row_stride(q), _row_stride(k), _row_stride(v), _p(q_out),
                             _p(_i32(indptr)), _p(_i32(offsets)), B, n_max, H, KH, D, _p(table.table), table.max_pos,
                             _p(kv_cache), _p(_i32(page_indices)), _p(_i32(page_indptr)), _p(_i32(last_page_len)),
                             _p(kv_cache2), _p(page_indices2), _p(page_indptr2), _p(last_page_len2),
                             _kv_geom(kv_cache, kv_layout)[0], *_kv_args(kv_cache, kv_scales, kv_layout), _stream()),
          "md_rope_append")
    return q_out


# ----------------------------------------------------------------------------- K1/K2/K3
class AttnWorkspace:
    """Scratch for split-KV partials, grown on demand (never inside a captured graph)."""

    def __init__(self, device="cuda"):
        self.device = device
        self.buf = torch.empty(1 << 20, dtype=torch.uint8, device=device)
        self._retired = []     # outgrown buffers stay allocated: captured hipGraphs have their addresses baked in

    def get(self, nbytes):
        if self.buf.numel() < nbytes:
            self._retired.append(self.buf)
            self.buf = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=self.device)
        return self.buf


def paged_attention(q, kv_cache, qo_indptr, page_indices, page_indptr, last_page_len, n_max, max_pages_per_req,
                    workspace: AttnWorkspace, causal=True, sm_scale=None, out=None, kv_scales=None,
                    kv_layout="NHD"):
    """mylib::target_decode / target_prefill / draft_decode / draft_prefill
    (Engine/SnapKV/backend.py:56-107): flashinfer BatchPrefillWithPagedKVCacheWrapper.run with the
    plan() arguments passed explicitly."""
    _gpu(q, kv_cache)
    B = page_indptr.numel() - 1
    H, D = q.shape[1], q.shape[2]
    page_size, KH, _ = _kv_geom(kv_cache, kv_layout)
    if sm_scale is None:
        sm_scale = 1.0 / math.sqrt(D)
    if out is None:
        out = torch.empty((q.shape[0], H, D), dtype=q.dtype, device=q.device)
    lib = _lib.load()
    nbytes = lib.md_paged_attn_workspace_bytes(B, n_max, H, KH, D, max_pages_per_req, page_size)
    ws = workspace.get(nbytes)
    check(lib.md_paged_attn(_p(q), _row_stride(q), _p(kv_cache), _p(out), _p(_i32(qo_indptr)), _p(_i32(page_indices)),
                            _p(_i32(page_indptr)), _p(_i32(last_page_len)), B, n_max, H, KH, D, page_size,
                            1 if causal else 0, float(sm_scale), max_pages_per_req,
                            *_kv_args(kv_cache, kv_scales, kv_layout),
                            _p(ws), ws.numel(), _stream()),
          "md_paged_attn")
    return out


# ----------------------------------------------------------------------------- K6
def snapkv_select(q_win, kv_cache, page_indices, page_indptr, ctx_len, window, budget, pool_kernel, draft_cache,
                  draft_page_indices, draft_page_indptr, draft_last_page_len, workspace: AttnWorkspace,
                  return_scores=False, kv_scales=None, kv_layout="NHD"):
    """Attention.gen_draft_kv (Engine/SnapKV/model.py:389-439): writes budget rows per request and kv head
    into draft_cache; returns the selected positions [B, KH, budget-window] int32 (reference order).
    kv_layout describes kv_cache (the source); draft_cache is always bf16 NHD."""
    _gpu(q_win, kv_cache, draft_cache)
    B = page_indptr.numel() - 1
    H, D = q_win.shape[1], q_win.shape[2]
    page_size, KH, _ = _kv_geom(kv_cache, kv_layout)
    if not q_win.is_contiguous():
        q_win = q_win.contiguous()
    idx = torch.empty((B, KH, budget - window), dtype=torch.int32, device=q_win.device)
    lib = _lib.load()
    nbytes = lib.md_snapkv_workspace_bytes(B, H, KH, ctx_len, window)
    if nbytes == 0:
        raise ValueError("md_snapkv_workspace_bytes: bad shape")
    ws = workspace.get(nbytes + 512)
    off = (-ws.data_ptr()) % 256
    check(lib.md_snapkv_select(_p(q_win), _p(kv_cache), _p(_i32(page_indices)), _p(_i32(page_indptr)), B, H, KH, D,
                               page_size, ctx_len, window, budget, pool_kernel, _p(draft_cache),
                               _p(_i32(draft_page_indices)), _p(_i32(draft_page_indptr)), _p(_i32(draft_last_page_len)),
                               _p(idx), *_kv_args(kv_cache, kv_scales, kv_layout), ctypes.c_void_p(ws.data_ptr() + off),
                               nbytes, _stream()), "md_snapkv_select")
    if return_scores:
        soff = lib.md_snapkv_scores_offset(B, H, KH, ctx_len, window)
        n = B * KH * (ctx_len - window)
        sc = ws[off + soff: off + soff + 2 * n].view(torch.bfloat16).view(B, KH, ctx_len - window).clone()
        return idx, sc
    return idx


# ----------------------------------------------------------------------------- K7
def streaming_shift_append(k_new, v_new, kv_cache, n_new, kv_len, sink, pages_per_req):
    """In-place equivalent of KVCache.prefill's eviction branch (Engine/StreamingLLM/model_draft.py:122-134)."""
    _gpu(k_new, v_new, kv_cache)
    B = k_new.shape[0] // n_new
    KH, D = kv_cache.shape[3], kv_cache.shape[4]
    lib = _lib.load()
    check(lib.md_streaming_shift_append(_p(k_new), _p(v_new), _row_stride(k_new), _row_stride(v_new), _p(kv_cache), B,
                                        n_new, kv_len, sink, pages_per_req, KH, D, kv_cache.shape[2], _stream()),
          "md_streaming_shift_append")


def streaming_rotate(kv_cache, rot_cache, B, valid_len, pages_per_req, table: RopeTable):
    """Rotated clone of the draft cache (Engine/StreamingLLM/model_draft.py:112-118,135-143): K rotated by
    slot position, V copied, rows [0, valid_len) of every request.  rot_cache may be kv_cache (in place)."""
    _gpu(kv_cache, rot_cache)
    KH, D = kv_cache.shape[3], kv_cache.shape[4]
    lib = _lib.load()
    check(lib.md_streaming_rotate(_p(kv_cache), _p(rot_cache), B, valid_len, pages_per_req, KH, D, kv_cache.shape[2],
                                  _p(table.table), table.max_pos, _stream()), "md_streaming_rotate")


# ----------------------------------------------------------------------------- K8
MD_W_BF16, MD_W_INT8 = 0, 1
EPI_NONE, EPI_SWIGLU = 0, 1


def linear_supported(M, N, K, swiglu=False):
    """True when md_linear (the weight-streaming skinny GEMM) takes this shape; otherwise use a library GEMM."""
    return bool(_lib.load().md_linear_supported(int(M), int(N), int(K), EPI_SWIGLU if swiglu else EPI_NONE))


class PackedWeight:
    """A linear's weight in the streaming layout of md_linear (include/magicdec_hip.h: w_packed = 1):
    [ceil(N/32)][K/16][64 lanes][8] -- element (t, s, lane, e) = W[32t + lane%32][16s + 8*(lane/32) + e], rows past N
    zero; for swiglu=True tile t = rows 16t..16t+15 of w1 then rows 16t..16t+15 of w3 (weight = [w1; w3], N = 2I).
    Built once at load time (a torch permute: plumbing); bf16 or int8."""

    def __init__(self, weight, swiglu=False):
        N, K = weight.shape
        if K % 16:
            raise ValueError("PackedWeight needs K % 16 == 0")
        self.N, self.K, self.swiglu, self.dtype = N, K, swiglu, weight.dtype
        if swiglu:
            inter = N // 2
            t = (inter + 15) // 16
            w1 = torch.zeros((t * 16, K), dtype=weight.dtype, device=weight.device)
            w3 = torch.zeros((t * 16, K), dtype=weight.dtype, device=weight.device)
            w1[:inter], w3[:inter] = weight[:inter], weight[inter:]
            rows = torch.cat([w1.view(t, 16, K), w3.view(t, 16, K)], dim=1)          # [t, 32, K]
        else:
            t = (N + 31) // 32
            rows = torch.zeros((t * 32, K), dtype=weight.dtype, device=weight.device)
            rows[:N] = weight
            rows = rows.view(t, 32, K)
        # [t, j, s, kh, e] -> [t, s, kh, j, e]
        self.data = rows.view(t, 32, K // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous()

    def unpack(self):
        """The row-major [N, K] tensor this copy was built from (the inverse permute: plumbing).  Used to re-materialise
        a row-major weight that was released after prefill (Transformer.release_rowmajor): the streaming layout is then
        the ONE resident copy of the weight."""
        t, K = self.data.shape[0], self.K
        rows = self.data.permute(0, 3, 1, 2, 4).reshape(t, 32, K)      # [t, s, kh, j, e] -> [t, j, (s, kh, e)]
        if not self.swiglu:
            return rows.view(t * 32, K)[:self.N] if t * 32 != self.N else rows.view(t * 32, K)
        inter = self.N // 2
        out = torch.empty((self.N, K), dtype=self.data.dtype, device=self.data.device)
        out[:inter].copy_(rows[:, :16].reshape(t * 16, K)[:inter])
        out[inter:].copy_(rows[:, 16:].reshape(t * 16, K)[:inter])
        return out


def linear(x, weight, bias=None, scales=None, swiglu=False, workspace: "AttnWorkspace" = None, out=None,
           pro: "DeferredNorm" = None):
    """F.linear(x, weight, bias) for M = x.shape[0] <= 256 rows on the hand-written gfx950 skinny GEMM (md_linear).
    x [M, K] bf16 with unit inner stride (row stride free); weight: a contiguous [N, K] tensor or a PackedWeight
    (streaming layout), bf16 -- or int8 with bf16 per-row `scales` (WeightOnlyInt8Linear semantics:
    bf16(x.w^T) * scales).  swiglu=True: weight = [w1; w3] (N = 2*I rows) and the result is
    silu(x.w1^T) * (x.w3^T) [M, I] with the reference's bf16 rounding points.
    pro (bf16 weights): x is pro.h, the un-normalised hidden state, and is normalised on the fly (md_linear_normed)."""
    packed = isinstance(weight, PackedWeight)
    wt = weight.data if packed else weight
    _gpu(x, wt, bias, scales)
    if x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("linear expects x [M, K] with unit inner stride")
    M, K = x.shape
    if packed:
        if weight.swiglu != swiglu:
            raise ValueError("PackedWeight was packed for a different epilogue")
        N, wk = weight.N, weight.K
    else:
        if weight.dim() != 2 or not weight.is_contiguous():
            raise ValueError("linear expects a contiguous weight [N, K]")
        N, wk = weight.shape
    if wk != K:
        raise ValueError(f"linear: x has K={K}, weight has K={wk}")
    if wt.dtype == torch.int8:
        if scales is None:
            raise ValueError("int8 weights need per-row scales")
        wd = MD_W_INT8
    elif wt.dtype == torch.bfloat16:
        wd = MD_W_BF16
    else:
        raise TypeError(f"linear: weight dtype {wt.dtype} unsupported (bf16 or int8)")
    epi = EPI_SWIGLU if swiglu else EPI_NONE
    n_out = N // 2 if swiglu else N
    if out is None:
        out = torch.empty((M, n_out), dtype=x.dtype, device=x.device)
    lib = _lib.load()
    nbytes = lib.md_linear_workspace_bytes(M, N, K, epi)
    ws = None
    if nbytes:
        if workspace is None:
            raise ValueError("linear: this shape splits K and needs a workspace")
        ws = workspace.get(nbytes + 256)
        off = (-ws.data_ptr()) % 256
    wsp = ctypes.c_void_p(ws.data_ptr() + off) if ws is not None else None
    if pro is not None:
        if wd != MD_W_BF16:
            raise TypeError("linear: the deferred norm needs bf16 weights")
        a = _lib.FusedLinearArgs()             # _set_pro's checks (x is pro.h, ssq [M, K / 32] fp32, bf16 weight [K])
        _set_pro(a, x, pro)
        check(lib.md_linear_normed(_p(x), x.stride(0), _p(pro.ssq), pro.ssq.shape[1], _p(pro.weight), pro.eps, _p(wt),
                                   1 if packed else 0, _p(bias), _p(out), out.stride(0), M, N, K, epi, wsp, nbytes,
                                   _stream()),
              "md_linear_normed")
        return out
    check(lib.md_linear(_p(x), x.stride(0), _p(wt), wd, 1 if packed else 0, _p(scales), _p(bias), _p(out),
                        out.stride(0), M, N, K, epi, wsp, nbytes, _stream()),
          "md_linear")
    return out


def linear_add_rmsnorm_supported(M, N, K):
    """True when md_linear_add_rmsnorm takes this shape (an md_linear shape whose K is split, N <= 8192)."""
    return bool(_lib.load().md_linear_add_rmsnorm_supported(int(M), int(N), int(K)))


def linear_add_rmsnorm(x, weight, resid, norm_weight, eps, bias=None, scales=None, workspace: "AttnWorkspace" = None):
    """(h, y) = (resid + F.linear(x, W, bias), rmsnorm(h) * norm_weight) on md_linear with the residual add and the norm
    fused into its split-K combine launch -- bit-identical to linear() followed by add_rmsnorm()."""
    packed = isinstance(weight, PackedWeight)
    wt = weight.data if packed else weight
    _gpu(x, wt, bias, scales, resid, norm_weight)
    if x.dim() != 2 or x.stride(1) != 1 or resid.dim() != 2 or resid.stride(1) != 1:
        raise ValueError("linear_add_rmsnorm expects 2-D x / resid with unit inner stride")
    M, K = x.shape
    N = weight.N if packed else weight.shape[0]
    if packed and weight.swiglu:
        raise ValueError("linear_add_rmsnorm: the weight was packed for the SwiGLU epilogue")
    if resid.shape != (M, N) or norm_weight.numel() != N:
        raise ValueError("linear_add_rmsnorm: resid must be [M, N], norm_weight [N]")
    wd = MD_W_INT8 if wt.dtype == torch.int8 else MD_W_BF16
    if wd == MD_W_INT8 and scales is None:
        raise ValueError("int8 weights need per-row scales")
    lib = _lib.load()
    nbytes = lib.md_linear_workspace_bytes(M, N, K, EPI_NONE)
    if workspace is None or not nbytes:
        raise ValueError("linear_add_rmsnorm: needs a workspace and a shape whose K is split")
    ws = workspace.get(nbytes + 256)
    off = (-ws.data_ptr()) % 256
    h = torch.empty((M, N), dtype=x.dtype, device=x.device)
    y = torch.empty((M, N), dtype=x.dtype, device=x.device)
    check(lib.md_linear_add_rmsnorm(_p(x), x.stride(0), _p(wt), wd, 1 if packed else 0, _p(scales), _p(bias), _p(resid),
                                    resid.stride(0), _p(norm_weight), float(eps), _p(h), _p(y), M, N, K,
                                    ctypes.c_void_p(ws.data_ptr() + off), nbytes, _stream()), "md_linear_add_rmsnorm")
    return h, y


# ----------------------------------------------------------------------------- K8b, K split over workgroups
def fused_split_supported(M, N, K):
    """md_linear_fused_split takes this shape (M <= 256, K % 128 == 0, N % 32 == 0)."""
    return 1 <= M <= 256 and K >= 128 and K % 128 == 0 and N >= 32 and N % 32 == 0


def _split_ws(lib, M, N, K, workspace):
    nbytes = lib.md_linear_fused_split_workspace_bytes(M, N, K)
    if not nbytes or workspace is None:
        raise ValueError("fused_split_linear: needs a workspace and a supported shape")
    ws = workspace.get(nbytes + 256)
    return ctypes.c_void_p(ws.data_ptr() + (-ws.data_ptr()) % 256), nbytes


def fused_split_linear(x, weight: "PackedWeight", bias=None, workspace: "AttnWorkspace" = None):
    """F.linear(x, W, bias) on the tile kernel with K also split over workgroups + the fixed-order combine launch
    (md_linear_fused_split; csrc/tilegemm.hip)."""
    _gpu(x, weight.data, bias)
    if x.dim() != 2 or x.stride(1) != 1 or weight.swiglu:
        raise ValueError("fused_split_linear expects a 2-D x with unit inner stride and a plain packed weight")
    M, K = x.shape
    N = weight.N
    lib = _lib.load()
    wsp, nbytes = _split_ws(lib, M, N, K, workspace)
    out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    check(lib.md_linear_fused_split(_p(x), x.stride(0), _p(weight.data), _p(bias), _p(out), out.stride(0), M, N, K, wsp,
                                    nbytes, _stream()), "md_linear_fused_split")
    return out


def fused_split_linear_add_rmsnorm(x, weight: "PackedWeight", resid, norm_weight, eps, bias=None,
                                   workspace: "AttnWorkspace" = None):
    """(h, y) = (resid + F.linear(x, W, bias), rmsnorm(h) * norm_weight): the split tile kernel + ONE combine launch that
    adds the slices in order, the residual, and normalises (md_linear_fused_split_add_rmsnorm) -- bit-identical to
    fused_split_linear() followed by add_rmsnorm()."""
    _gpu(x, weight.data, bias, resid, norm_weight)
    if x.dim() != 2 or x.stride(1) != 1 or resid.dim() != 2 or resid.stride(1) != 1 or weight.swiglu:
        raise ValueError("fused_split_linear_add_rmsnorm expects 2-D x / resid with unit inner stride and a plain packed weight")
    M, K = x.shape
    N = weight.N
    if resid.shape != (M, N) or norm_weight.numel() != N:
        raise ValueError("fused_split_linear_add_rmsnorm: resid must be [M, N], norm_weight [N]")
    lib = _lib.load()
    wsp, nbytes = _split_ws(lib, M, N, K, workspace)
    h = torch.empty((M, N), dtype=x.dtype, device=x.device)
    y = torch.empty((M, N), dtype=x.dtype, device=x.device)
    check(lib.md_linear_fused_split_add_rmsnorm(_p(x), x.stride(0), _p(weight.data), _p(bias), _p(resid), resid.stride(0),
                                                _p(norm_weight), float(eps), _p(h), _p(y), M, N, K, wsp, nbytes, _stream()),
          "md_linear_fused_split_add_rmsnorm")
    return h, y


# ----------------------------------------------------------------------------- K8c
def linear_block_supported(M, N, K, swiglu=False):
    """True when md_linear_block (csrc/blockgemm.hip: the block-tile GEMM of the 129..256-row verify linears) takes
    this shape (M <= 256, N % 128 == 0, K % 64 == 0)."""
    return bool(_lib.load().md_linear_block_supported(int(M), int(N), int(K), EPI_SWIGLU if swiglu else EPI_NONE))


def _block_ws(lib, M, N, K, force, workspace):
    nbytes = lib.md_linear_block_workspace_bytes(M, N, K, 1 if force else 0)
    if not nbytes:
        return None, 0
    if workspace is None:
        raise ValueError("linear_block: this shape splits K and needs a workspace")
    ws = workspace.get(nbytes + 256)
    return ctypes.c_void_p(ws.data_ptr() + (-ws.data_ptr()) % 256), nbytes


def _block_args(x, weight, swiglu):
    if not isinstance(weight, PackedWeight) or weight.dtype != torch.bfloat16:
        raise TypeError("linear_block streams a bf16 PackedWeight")
    _gpu(x, weight.data)
    if x.dim() != 2 or x.stride(1) != 1 or x.dtype != torch.bfloat16:
        raise ValueError("linear_block expects x [M, K] bf16 with unit inner stride")
    if weight.K != x.shape[1]:
        raise ValueError(f"linear_block: x has K={x.shape[1]}, weight has K={weight.K}")
    if weight.swiglu != swiglu:
        raise ValueError("PackedWeight was packed for a different epilogue")
    return x.shape[0], weight.N, weight.K


def _block_bias(bias, N):
    """md_linear_block reads the bias as 16-byte vectors of N bf16 values."""
    if bias is None:
        return
    _gpu(bias)
    if bias.dtype != torch.bfloat16 or bias.numel() != N or not bias.is_contiguous() or bias.data_ptr() % 16:
        raise ValueError(f"linear_block: bias must be a contiguous, 16-byte aligned bf16 vector of {N} elements")


def linear_block(x, weight: "PackedWeight", bias=None, swiglu=False, workspace: "AttnWorkspace" = None, out=None):
    """F.linear(x, W, bias) (or silu(x.w1^T) * (x.w3^T) for swiglu=True, weight = [w1; w3]) for 129..256 rows on the
    block-tile GEMM md_linear_block; same contract and rounding points as linear()."""
    M, N, K = _block_args(x, weight, swiglu)
    _block_bias(bias, N)
    n_out = N // 2 if swiglu else N
    if out is None:
        out = torch.empty((M, n_out), dtype=x.dtype, device=x.device)
    else:
        _gpu(out)
        if out.dtype != torch.bfloat16 or out.dim() != 2 or out.shape != (M, n_out) or out.stride(1) != 1:
            raise ValueError("linear_block: out must be bf16 [M, N_out] with unit inner stride")
    lib = _lib.load()
    wsp, nbytes = _block_ws(lib, M, N, K, False, workspace)
    check(lib.md_linear_block(_p(x), x.stride(0), _p(weight.data), _p(bias), _p(out), out.stride(0), M, N, K,
                              EPI_SWIGLU if swiglu else EPI_NONE, wsp, nbytes, _stream()), "md_linear_block")
    return out


def linear_block_add_rmsnorm(x, weight: "PackedWeight", resid, norm_weight, eps, bias=None,
                             workspace: "AttnWorkspace" = None):
    """(h, y) = (resid + F.linear(x, W, bias), rmsnorm(h) * norm_weight): md_linear_block with the residual add and the
    norm in its combine launch -- the combine kernel of linear_add_rmsnorm (same rounding points)."""
    M, N, K = _block_args(x, weight, False)
    _block_bias(bias, N)
    _gpu(resid, norm_weight)
    if resid.dim() != 2 or resid.stride(1) != 1 or resid.shape != (M, N) or norm_weight.numel() != N:
        raise ValueError("linear_block_add_rmsnorm: resid must be [M, N] with unit inner stride, norm_weight [N]")
    lib = _lib.load()
    wsp, nbytes = _block_ws(lib, M, N, K, True, workspace)
    h = torch.empty((M, N), dtype=x.dtype, device=x.device)
    y = torch.empty((M, N), dtype=x.dtype, device=x.device)
    check(lib.md_linear_block_add_rmsnorm(_p(x), x.stride(0), _p(weight.data), _p(bias), _p(resid), resid.stride(0),
                                          _p(norm_weight), float(eps), _p(h), _p(y), M, N, K, wsp, nbytes, _stream()),
          "md_linear_block_add_rmsnorm")
    return h, y


# ----------------------------------------------------------------------------- K8b
FL_NONE, FL_SWIGLU, FL_RESID, FL_ROPE_APPEND = 0, 1, 2, 3


def fused_linear_supported(M, N, K, epilogue=FL_NONE):
    """True when md_linear_fused (csrc/tilegemm.hip) takes this shape."""
    return bool(_lib.load().md_linear_fused_supported(int(M), int(N), int(K), int(epilogue)))


def _fused_args(x, weight: "PackedWeight", bias, epilogue):
    _gpu(x, weight.data, bias)
    if x.dim() != 2 or x.stride(1) != 1 or x.dtype != torch.bfloat16:
        raise ValueError("fused linear expects x [M, K] bf16 with unit inner stride")
    if weight.dtype != torch.bfloat16:
        raise TypeError("fused linear streams bf16 weights")
    M, K = x.shape
    if weight.K != K:
        raise ValueError(f"fused linear: x has K={K}, weight has K={weight.K}")
    if weight.swiglu != (epilogue == FL_SWIGLU):
        raise ValueError("PackedWeight was packed for a different epilogue")
    a = _lib.FusedLinearArgs()
    a.x, a.ldx, a.w_packed, a.bias = x.data_ptr(), x.stride(0), weight.data.data_ptr(), (bias.data_ptr() if bias is not None else None)
    a.M, a.N, a.K, a.epilogue = M, weight.N, K, epilogue
    return a


def _run_fused(a):
    check(_lib.load().md_linear_fused(ctypes.byref(a), _stream()), "md_linear_fused")


class DeferredNorm:
    """An RMSNorm that has NOT been applied yet: `h` [M, K] are the un-normalised hidden states and `ssq` [M, K/32]
    the per-row partial sums of squares the residual epilogue of the producing linear wrote; the consuming fused
    linear normalises on the fly (md_linear_fused, deferred RMSNorm).  `materialize()` is the stand-alone kernel."""

    def __init__(self, h, ssq, weight, eps):
        self.h, self.ssq, self.weight, self.eps = h, ssq, weight, float(eps)

    def materialize(self):
        return rmsnorm(self.h, self.weight, self.eps)


def _set_pro(a, x, pro: "DeferredNorm"):
    if pro.h.data_ptr() != x.data_ptr() or pro.ssq.dtype != torch.float32 or not pro.ssq.is_contiguous():
        raise ValueError("deferred norm: x must be the un-normalised h the partial sums belong to")
    if (pro.ssq.dim() != 2 or pro.ssq.shape[0] != x.shape[0] or pro.ssq.shape[1] * 32 != x.shape[1]
            or pro.weight.numel() != x.shape[1] or pro.weight.dtype != torch.bfloat16):
        # ssq must hold one partial per 32-column tile of the producer (a shorter one would give a silently wrong rstd)
        raise ValueError("deferred norm: ssq [M, K / 32] float32 and a bf16 norm weight [K] expected")
    a.pro_ssq, a.pro_tiles = pro.ssq.data_ptr(), pro.ssq.shape[1]
    a.pro_norm_w, a.pro_eps = pro.weight.data_ptr(), pro.eps


def fused_linear(x, weight: "PackedWeight", bias=None, swiglu=False, resid=None, out=None, want_ssq=False,
                 pro: "DeferredNorm" = None):
    """One launch (md_linear_fused): F.linear(x, W, bias) for M <= 256 rows over a PackedWeight, optionally followed by
    SiLU(h1) * h3 (swiglu=True, weight = [w1; w3]) or by the bf16 residual add `resid + linear` (resid [M, N]).
    want_ssq (with resid): also return the per-row partial sums of squares [M, N/32] of the result (the producer half of
    a deferred RMSNorm).  pro (with swiglu): x is pro.h and is normalised on the fly (the consumer half)."""
    epi = FL_SWIGLU if swiglu else (FL_RESID if resid is not None else FL_NONE)
    a = _fused_args(x, weight, bias, epi)
    n_out = weight.N // 2 if swiglu else weight.N
    if out is None:
        out = torch.empty((x.shape[0], n_out), dtype=x.dtype, device=x.device)
    else:
        _gpu(out)
        if out.dtype != torch.bfloat16 or out.dim() != 2 or out.shape != (x.shape[0], n_out) or out.stride(1) != 1:
            raise ValueError("fused linear: out must be bf16 [M, N_out] with unit inner stride")
    a.out, a.ldo = out.data_ptr(), out.stride(0)
    ssq = None
    if resid is not None:
        _gpu(resid)
        if resid.shape != out.shape or resid.stride(1) != 1 or resid.dtype != torch.bfloat16:
            raise ValueError("fused linear: resid must be bf16 [M, N] with unit inner stride")
        a.resid, a.ldr = resid.data_ptr(), resid.stride(0)
        if want_ssq:
            ssq = torch.empty((x.shape[0], weight.N // 32), dtype=torch.float32, device=x.device)
            a.ssq_out = ssq.data_ptr()
    elif want_ssq:
        raise ValueError("fused linear: want_ssq belongs to the residual epilogue")
    if pro is not None:
        if not swiglu:
            raise ValueError("fused linear: the deferred-norm prologue exists for the w1|w3 and qkv linears")
        _set_pro(a, x, pro)
    _run_fused(a)
    return (out, ssq) if want_ssq else out


def fused_qkv_rope_append(x, weight: "PackedWeight", bias, H, KH, D, rows_per_req, offsets, table: RopeTable, kv_cache,
                          page_indices, page_indptr, last_page_len, kv_cache2=None, page_indices2=None,
                          page_indptr2=None, last_page_len2=None, kv_scales=None, kv_layout="NHD",
                          pro: "DeferredNorm" = None):
    """wqkv + RoPE + paged KV append in one launch (md_linear_fused, MD_FL_ROPE_APPEND): x [M, K] are the normalised
    hidden states of a decode / verify step in which request b owns rows [b * rows_per_req, (b+1) * rows_per_req).
    Returns the rotated q [M, H, D]; rotated k and v go to the paged cache (and to kv_cache2, bf16 NHD, when given).
    Same results as linear -> rope_append."""
    a = _fused_args(x, weight, bias, FL_ROPE_APPEND)
    _gpu(kv_cache, kv_cache2)
    M = x.shape[0]
    if weight.N != (H + 2 * KH) * D:
        raise ValueError("fused qkv: weight rows must be (H + 2 KH) * D")
    if M % rows_per_req:
        raise ValueError("fused qkv: every request must own rows_per_req rows")
    q_out = torch.empty((M, H, D), dtype=x.dtype, device=x.device)
    a.out, a.ldo = q_out.data_ptr(), H * D
    a.H, a.KH, a.D, a.rows_per_req, a.max_pos = H, KH, D, rows_per_req, table.max_pos
    offsets = _i32(offsets)
    pi, pp, lp = _i32(page_indices), _i32(page_indptr), _i32(last_page_len)
    a.offsets, a.cos_sin = offsets.data_ptr(), table.table.data_ptr()
    a.cache, a.page_indices, a.page_indptr, a.last_page_len = kv_cache.data_ptr(), pi.data_ptr(), pp.data_ptr(), lp.data_ptr()
    keep = [offsets, pi, pp, lp]
    if kv_cache2 is not None:
        pi2, pp2, lp2 = _i32(page_indices2), _i32(page_indptr2), _i32(last_page_len2)
        keep += [pi2, pp2, lp2]
        a.cache2, a.page_indices2, a.page_indptr2, a.last_page_len2 = (kv_cache2.data_ptr(), pi2.data_ptr(),
                                                                       pp2.data_ptr(), lp2.data_ptr())
    a.page_size = _kv_geom(kv_cache, kv_layout)[0]
    kvd, ks, vs = _kv_args(kv_cache, kv_scales, kv_layout)
    a.kv_dtype, a.k_scale, a.v_scale = kvd, ks, vs
    if pro is not None:                      # x is the un-normalised h: normalise on the fly (deferred RMSNorm)
        _set_pro(a, x, pro)
    _run_fused(a)
    return q_out


# ----------------------------------------------------------------------------- K9
def rmsnorm(x, weight, eps):
    _gpu(x, weight)
    dim = x.shape[-1]
    x2 = x.reshape(-1, dim)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    y = torch.empty_like(x2)
    lib = _lib.load()
    check(lib.md_rmsnorm(_p(x2), _p(weight), _p(y), x2.shape[0], dim, float(eps), _stream()), "md_rmsnorm")
    return y.view(x.shape)


def add_rmsnorm(x, r, weight, eps):
    """h = x + r ; y = rmsnorm(h).  Returns (h, y)."""
    _gpu(x, r, weight)
    dim = x.shape[-1]
    x2, r2 = x.reshape(-1, dim), r.reshape(-1, dim)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    if not r2.is_contiguous():
        r2 = r2.contiguous()
    h = torch.empty_like(x2)
    y = torch.empty_like(x2)
    lib = _lib.load()
    check(lib.md_add_rmsnorm(_p(x2), _p(r2), _p(weight), _p(h), _p(y), x2.shape[0], dim, float(eps), _stream()),
          "md_add_rmsnorm")
    return h.view(x.shape), y.view(x.shape)


def silu_mul(a, b):
    """bf16(silu(a)) * b for 2-D row-strided views (e.g. the two halves of a fused w1|w3 GEMM output)."""
    _gpu(a, b)
    if a.dim() != 2 or a.stride(1) != 1 or b.stride(1) != 1:
        raise ValueError("silu_mul expects 2-D tensors with unit inner stride")
    rows, dim = a.shape
    y = torch.empty((rows, dim), dtype=a.dtype, device=a.device)
    lib = _lib.load()
    check(lib.md_silu_mul(_p(a), _p(b), a.stride(0), b.stride(0), _p(y), rows, dim, _stream()), "md_silu_mul")
    return y


# ----------------------------------------------------------------------------- K10
def argmax(logits, index_offset=0, return_values=False):
    """Row-wise argmax of bf16 logits [rows, vocab] (lowest index among equal maxima)."""
    _gpu(logits)
    if logits.dim() != 2 or logits.stride(1) != 1:
        raise ValueError("argmax expects [rows, vocab] with unit inner stride")
    rows, vocab = logits.shape
    idx = torch.empty(rows, dtype=torch.int64, device=logits.device)
    vals = torch.empty(rows, dtype=logits.dtype, device=logits.device) if return_values else None
    lib = _lib.load()
    check(lib.md_argmax(_p(logits), logits.stride(0), rows, vocab, int(index_offset), _p(vals), _p(idx), _stream()),
          "md_argmax")
    return (vals, idx) if return_values else idx


def argmax_tp_slots(logits, tp_rank, tp_world, index_offset=0):
    """(vals [rows, tp_world] bf16, idx [rows, tp_world] int64): the row-wise argmax of this rank's vocab shard in column
    tp_rank, zeros elsewhere -- the tensors the reference sum-all-reduces before its merge (Engine/SnapKV/model.py:178-184),
    written by the argmax launch itself."""
    _gpu(logits)
    if logits.dim() != 2 or logits.stride(1) != 1:
        raise ValueError("argmax_tp_slots expects [rows, vocab] with unit inner stride")
    rows, vocab = logits.shape
    vals = torch.empty((rows, tp_world), dtype=logits.dtype, device=logits.device)
    idx = torch.empty((rows, tp_world), dtype=torch.int64, device=logits.device)
    check(_lib.load().md_argmax_tp_slots(_p(logits), logits.stride(0), rows, vocab, int(index_offset), int(tp_rank),
                                         int(tp_world), _p(vals), _p(idx), _stream()), "md_argmax_tp_slots")
    return vals, idx


def tp_argmax_merge(vals, idx):
    """vals [rows, tp] bf16, idx [rows, tp] int64 -> winning global index per row (lowest rank on ties)."""
    _gpu(vals, idx)
    rows, tp = vals.shape
    out = torch.empty(rows, dtype=torch.int64, device=vals.device)
    lib = _lib.load()
    check(lib.md_tp_argmax_merge(_p(vals.contiguous()), _p(idx.contiguous()), rows, tp, _p(out), _stream()),
          "md_tp_argmax_merge")
    return out


# ----------------------------------------------------------------------------- a1
def accept_rollback(tokens_buffer, target_tokens, output, num_nodes, cachelens, last_page_len, draft_cachelens,
                    draft_last_page_len, gamma, draft_rollback, draft_cap, eot_1, eot_2, max_nodes, accept_nums,
                    bonus, double_buffer, cachelens_update, flags):
    """The verify-loop body (tests/SnapKV/longspec_benchmark.py:208-285) as one launch; all tensors are
    updated in place, `flags` = int32[2] (terminal, next_double)."""
    _gpu(tokens_buffer, target_tokens, output, num_nodes, cachelens, last_page_len)
    B = tokens_buffer.shape[0]
    lib = _lib.load()
    check(lib.md_accept_rollback(_p(tokens_buffer), _p(target_tokens), _p(output), output.shape[1], _p(num_nodes),
                                 _p(cachelens), _p(last_page_len), _p(draft_cachelens), _p(draft_last_page_len), B,
                                 gamma, draft_rollback, draft_cap, int(eot_1), int(eot_2), int(max_nodes),
                                 _p(accept_nums), _p(bonus), _p(double_buffer), _p(cachelens_update), _p(flags),
                                 _stream()), "md_accept_rollback")
