"""CPU restatement of the four flashinfer entry points MagicDec's decode path calls.

TEST INFRASTRUCTURE ONLY.  Nothing under ``magicdec_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use it, as the checker / the CPU baseline.

PARITY UNPINNED at this boundary: the arithmetic of these ops lives in the
third-party dependency **flashinfer** (wheel index ``cu121/torch2.4``, no version
pin in the reference's requirements.txt / README.md:41; the call signatures used
match the flashinfer **0.1.6** public API).  flashinfer is neither vendored under
the reference checkout nor installed here, and the reference holds no golden
vectors or tests for it.  This file restates flashinfer's *documented* semantics
 and is cross-checked in tests/ against an independent
dense fp32 ``softmax(QK^T/sqrt(D))V``; everything above this boundary (page-table
bookkeeping, SnapKV select, StreamingLLM eviction, TP sharding, the accept loop)
is pinned against the real reference imported on CPU (oracle/gen_golden.py).

Call sites restated (reference file:line):
  append_paged_kv_cache                 Engine/utils.py:46-54
  BatchPrefillWithPagedKVCacheWrapper   Engine/SnapKV/backend.py:49-55,91-95 (ctor),
      .plan                             :148-159,186-197,218-229,276-287
      .run                              :60-64,73-77,100-104
  rope.apply_rope / apply_llama31_rope  Engine/SnapKV/model.py:140,152
"""
from __future__ import annotations

import math

import numpy as np
import torch


def kv_len_of(indptr, last_page_len, b, page_size):
    npages = int(indptr[b + 1]) - int(indptr[b])
    return (npages - 1) * page_size + int(last_page_len[b]) if npages > 0 else 0


def append_paged_kv_cache(k, v, append_indptr, cache, indices, indptr, last_page_len):
    """Row j of request b goes to position len_b - n_b + j (the page table already
    includes the appended rows: Engine/SnapKV/backend.py:147 bumps last_page_len
    before the model call).  cache: [pages, 2, page_size, KH, D] (NHD)."""
    page_size = cache.shape[2]
    B = indptr.numel() - 1
    for b in range(B):
        a0, a1 = int(append_indptr[b]), int(append_indptr[b + 1])
        n_b = a1 - a0
        if n_b == 0:
            continue
        ln = kv_len_of(indptr, last_page_len, b, page_size)
        p0 = int(indptr[b])
        for j in range(n_b):
            pos = ln - n_b + j
            page = int(indices[p0 + pos // page_size])
            slot = pos % page_size
            cache[page, 0, slot] = k[a0 + j]
            cache[page, 1, slot] = v[a0 + j]


# ---------------------------------------------------------------------------- fp8 (OCP e4m3fn) KV cache
# Not a flashinfer 0.1.6 / MagicDec feature: BASELINE.json configs[4] asks for it as the CDNA4 fp8 path, so the
# definition below IS the specification and the GPU kernels are checked against it.
FP8_MAX = 448.0


def quantize_fp8(x, scale):
    """x [rows, KH, D] bf16, scale [KH] float32 -> e4m3fn bytes: rne(clamp(x * (1/scale), +-448)).
    (float32 multiply by the float32 reciprocal; the clamp makes the cast saturating.)"""
    inv = (1.0 / scale.float()).view(1, -1, 1)
    return (x.float() * inv).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn)


def append_paged_kv_cache_fp8(k, v, append_indptr, cache, indices, indptr, last_page_len, k_scale, v_scale):
    """append_paged_kv_cache into an e4m3fn cache (viewed as uint8 for the row copies)."""
    append_paged_kv_cache(quantize_fp8(k, k_scale).view(torch.uint8), quantize_fp8(v, v_scale).view(torch.uint8),
                          append_indptr, cache.view(torch.uint8), indices, indptr, last_page_len)


def dequantize_cache_fp8(cache, k_scale, v_scale):
    """e4m3fn cache [pages, 2, ps, KH, D] -> float32 cache of exact products byte * scale."""
    out = cache.float()
    out[:, 0] *= k_scale.float().view(1, 1, -1, 1)
    out[:, 1] *= v_scale.float().view(1, 1, -1, 1)
    return out


def gather_request_kv(cache, indices, indptr, last_page_len, b):
    """All K,V rows of request b as [len_b, KH, D] tensors."""
    page_size = cache.shape[2]
    ln = kv_len_of(indptr, last_page_len, b, page_size)
    p0, p1 = int(indptr[b]), int(indptr[b + 1])
    pages = indices[p0:p1].long()
    kv = cache[pages]  # [np, 2, ps, KH, D]
    k = kv[:, 0].reshape(-1, cache.shape[3], cache.shape[4])[:ln]
    v = kv[:, 1].reshape(-1, cache.shape[3], cache.shape[4])[:ln]
    return k, v


# How the probabilities enter the P.V product (a test-side switch, like magicdec_ref.LINEAR_MODE):
#   "fp32": softmax in fp32, P.V with fp32 P (the documented *semantics*: this module's default and the oracle
#           every fixture / lock-step log is recorded with);
#   "bf16": the tensor-core *algorithm* flashinfer's prefill kernels (and csrc/attn.hip) run: p = exp(s - M) in fp32,
#           the row sum l accumulates the UN-rounded p, p is rounded to the KV dtype (bf16) to be the MMA operand,
#           P.V accumulates in fp32 and the result is divided by l.  A second valid implementation of the same op:
#           tests/test_gpu_engine.py replays it beside the HIP engine so that the argmax flips the bf16 P causes are
#           MEASURED instead of asserted.
ATTN_P_MODE = "fp32"


def batch_prefill_paged(q, cache, qo_indptr, indices, indptr, last_page_len, num_qo_heads, num_kv_heads,
                        head_dim, causal=True, sm_scale=None):
    """flashinfer BatchPrefillWithPagedKVCacheWrapper.plan(...)+run(q, cache).

    Query row i of request b (m_b rows) attends kv positions <= len_b - m_b + i
    (causal) ; q head h uses kv head h // (H/KH); softmax(q.k*sm_scale) in fp32,
    P.V accumulated in fp32, output rounded to q.dtype.  No positional encoding,
    no logit cap."""
    H, KH, D = num_qo_heads, num_kv_heads, head_dim
    g = H // KH
    if sm_scale is None:
        sm_scale = 1.0 / math.sqrt(D)
    out = torch.empty_like(q)
    B = indptr.numel() - 1
    for b in range(B):
        q0, q1 = int(qo_indptr[b]), int(qo_indptr[b + 1])
        m_b = q1 - q0
        if m_b == 0:
            continue
        k, v = gather_request_kv(cache, indices, indptr, last_page_len, b)
        ln = k.shape[0]
        qf = q[q0:q1].float().view(m_b, KH, g, D)
        kf = k.float()  # [ln, KH, D]
        vf = v.float()
        s = torch.einsum("mhgd,lhd->hgml", qf, kf) * sm_scale  # [KH,g,m,ln]
        if causal:
            pos = torch.arange(ln).view(1, 1, 1, ln)
            lim = (ln - m_b + torch.arange(m_b)).view(1, 1, m_b, 1)
            s = s.masked_fill(pos > lim, float("-inf"))
        if ATTN_P_MODE == "bf16":
            m = s.amax(dim=-1, keepdim=True)
            m = torch.where(torch.isinf(m), torch.zeros_like(m), m)      # rows with an empty key set
            e = torch.exp(s - m)
            l = e.sum(dim=-1)                                             # [KH,g,m] un-rounded
            o = torch.einsum("hgml,lhd->mhgd", e.to(torch.bfloat16).float(), vf)
            o = o / l.permute(2, 0, 1).unsqueeze(-1).clamp_min(1e-38)
            o = o.reshape(m_b, H, D)
        else:
            p = torch.softmax(s, dim=-1)
            p = torch.nan_to_num(p, nan=0.0)  # rows with an empty key set
            o = torch.einsum("hgml,lhd->mhgd", p, vf).reshape(m_b, H, D)
        out[q0:q1] = o.to(q.dtype)
    return out


def rope_freqs(D, rope_theta, rope_scale, low_freq_factor=None, high_freq_factor=None, old_context_len=None):
    """Per-pair rotation frequency, float64.  Plain: theta^(-2i/D)/rope_scale.
    Llama-3.1 (apply_llama31_rope): smooth = clamp((old_ctx*f/(2pi) - low)/(high-low), 0, 1);
    f' = (1-smooth)*f/rope_scale + smooth*f."""
    i = np.arange(D // 2, dtype=np.float64)
    f = np.power(float(rope_theta), -2.0 * i / float(D))
    if low_freq_factor is not None and high_freq_factor is not None:
        smooth = (float(old_context_len) * f / (2.0 * np.pi) - float(low_freq_factor)) / (
            float(high_freq_factor) - float(low_freq_factor))
        smooth = np.clip(smooth, 0.0, 1.0)
        f = (1.0 - smooth) * f / float(rope_scale) + smooth * f
    else:
        f = f / float(rope_scale)
    return f


def rope_table(max_pos, D, rope_theta, rope_scale, low_freq_factor=None, high_freq_factor=None,
               old_context_len=None):
    """float32 [max_pos, D/2, 2] (cos, sin); angles and trig in float64, one rounding."""
    f = rope_freqs(D, rope_theta, rope_scale, low_freq_factor, high_freq_factor, old_context_len)
    ang = np.arange(max_pos, dtype=np.float64)[:, None] * f[None, :]
    tab = np.stack([np.cos(ang), np.sin(ang)], axis=-1).astype(np.float32)
    return torch.from_numpy(tab)


def _rope_rows(x, pos, table):
    """x: [n, heads, D] (bf16/fp16), pos: [n] int64; interleaved pairs (x[2i], x[2i+1]).
    fp32 arithmetic with separately rounded products (numpy semantics)."""
    cs = table[pos]  # [n, D/2, 2]
    cos = cs[..., 0].unsqueeze(1)
    sin = cs[..., 1].unsqueeze(1)
    xf = x.float()
    xe, xo = xf[..., 0::2], xf[..., 1::2]
    ye = xe * cos - xo * sin
    yo = xo * cos + xe * sin
    y = torch.stack([ye, yo], dim=-1).reshape(x.shape)
    return y.to(x.dtype)


def apply_rope(q, k, indptr, offsets, table):
    """flashinfer.rope.apply_rope / apply_llama31_rope with interleave=True: token j
    of request b has position offsets[b]+j.  Returns new tensors."""
    n = q.shape[0]
    pos = torch.zeros(n, dtype=torch.long)
    B = indptr.numel() - 1
    for b in range(B):
        a0, a1 = int(indptr[b]), int(indptr[b + 1])
        pos[a0:a1] = int(offsets[b]) + torch.arange(a1 - a0)
    return _rope_rows(q, pos, table), _rope_rows(k, pos, table)


# ---------------------------------------------------------------------------
# A module object that quacks like `flashinfer` for importing the reference on
# CPU (oracle/ref_import.py installs it into sys.modules before the import).
# ---------------------------------------------------------------------------
class BatchPrefillWithPagedKVCacheWrapper:
    def __init__(self, workspace, kv_layout="NHD", use_cuda_graph=False, qo_indptr_buf=None,
                 paged_kv_indptr_buf=None, paged_kv_indices_buf=None, paged_kv_last_page_len_buf=None):
        assert kv_layout == "NHD"
        self._plan = None

    def plan(self, qo_indptr, paged_kv_indptr, paged_kv_indices, paged_kv_last_page_len, num_qo_heads,
             num_kv_heads, head_dim, page_size, q_data_type=None, causal=True, **kw):
        self._plan = dict(qo_indptr=qo_indptr.clone(), indptr=paged_kv_indptr.clone(),
                          indices=paged_kv_indices.clone(), last=paged_kv_last_page_len.clone(),
                          H=num_qo_heads, KH=num_kv_heads, D=head_dim, causal=causal)

    def run(self, q, kv_cache):
        p = self._plan
        return batch_prefill_paged(q, kv_cache, p["qo_indptr"], p["indices"], p["indptr"], p["last"], p["H"],
                                   p["KH"], p["D"], causal=p["causal"])


class _RopeNS:
    _tables = {}

    @classmethod
    def _table(cls, D, theta, scale, low, high, old):
        key = (D, theta, scale, low, high, old)
        if key not in cls._tables:
            cls._tables[key] = rope_table(1 << 17, D, theta, scale, low, high, old)
        return cls._tables[key]

    @classmethod
    def apply_rope(cls, q, k, indptr, offsets, interleave=False, rope_scale=1.0, rope_theta=1e4):
        assert interleave
        return apply_rope(q, k, indptr, offsets, cls._table(q.shape[-1], rope_theta, rope_scale, None, None, None))

    @classmethod
    def apply_llama31_rope(cls, q, k, indptr, offsets, interleave=False, rope_scale=8.0, rope_theta=5e5,
                           low_freq_factor=1.0, high_freq_factor=4.0, old_context_len=8192):
        assert interleave
        return apply_rope(q, k, indptr, offsets,
                          cls._table(q.shape[-1], rope_theta, rope_scale, low_freq_factor, high_freq_factor,
                                     old_context_len))


rope = _RopeNS
