"""GPU parity tests of every C-ABI entry point against the oracle / golden vectors (run with -m gpu).

Bars (stated per test): bit-exact for integer / byte / index work (KV append, RoPE, StreamingLLM eviction,
accept loop, argmax, top-k given equal scores); bf16-rounding-level tolerance for floating-point kernels
(attention, norms, SiLU)."""
import math
import zlib

import numpy as np
import pytest
import torch

from oracle import flashinfer_ref as fr
from oracle import magicdec_ref as mr
from tests import golden_cfg as gc
from tests.conftest import parity_report
from tests.parity_util import check_attention, dense_attention_f64

pytestmark = pytest.mark.gpu

DEV = "cuda"
BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    from magicdec_amd import ops as _ops
    _ops._lib.load()      # fail loudly if the HIP library is missing
    return _ops


def bits(t):
    return t.contiguous().view(torch.int16)


def case_seed(name):
    """Fixed per-case seed (crc32 of the case name; independent of PYTHONHASHSEED)."""
    return zlib.crc32(name.encode()) % 1000


def make_paged(B, lens, KH, D, seed, page_size=128, scatter=False, extra_pages=2):
    g = torch.Generator().manual_seed(seed)
    max_pages = max(1, max((l + page_size - 1) // page_size for l in lens))
    tot = B * max_pages + extra_pages
    cache = torch.randn(tot, 2, page_size, KH, D, generator=g).to(BF)
    perm = torch.randperm(tot, generator=g) if scatter else torch.arange(tot)
    indices, indptr, last = [], [0], []
    for b in range(B):
        npg = (lens[b] + page_size - 1) // page_size
        indices += [int(perm[b * max_pages + i]) for i in range(npg)]
        indptr.append(indptr[-1] + npg)
        last.append(lens[b] - (npg - 1) * page_size if npg else 0)
    return (cache, torch.tensor(indices + [0], dtype=torch.int32), torch.tensor(indptr, dtype=torch.int32),
            torch.tensor(last, dtype=torch.int32), max_pages)


# ----------------------------------------------------------------------------------------- attention
ATTN_CASES = [
    ("verify-8b-shape", 2, 4, 8, 2, 128, [300, 257], True, False),
    ("verify-tile-edge-32", 1, 4, 4, 1, 128, [32], True, False),
    ("verify-tile-edge-33", 1, 4, 4, 1, 128, [33], True, False),
    ("verify-short", 1, 4, 4, 1, 128, [20], True, False),
    ("verify-ragged-scattered-pages", 3, 4, 32, 8, 128, [1000, 129, 640], True, True),
    ("verify-split-kv", 2, 4, 8, 2, 128, [8069, 7000], True, False),
    ("draft-1row-d64", 4, 1, 32, 8, 64, [260, 258, 300, 257], True, False),
    ("draft-2row-d64", 4, 2, 8, 2, 64, [260, 258, 300, 257], True, False),
    ("g8-two-mtiles", 2, 4, 16, 2, 128, [500, 300], True, False),
    ("g5-padded-mtile", 2, 4, 10, 2, 128, [500, 300], True, False),
    ("mha-g1", 2, 1, 12, 12, 64, [129, 200], True, False),
    ("prefill-chunk-128", 2, 128, 8, 2, 128, [384, 384], True, False),
    ("prefill-last-chunk-32", 2, 32, 8, 2, 128, [160, 160], True, False),
    ("prefill-d64", 2, 128, 8, 2, 64, [256, 256], True, False),
    ("non-causal", 2, 4, 8, 2, 128, [300, 257], False, False),
    ("empty-request", 2, 4, 8, 2, 128, [0, 200], True, False),
]


@pytest.mark.parametrize("name,B,n,H,KH,D,lens,causal,scatter", ATTN_CASES, ids=[c[0] for c in ATTN_CASES])
def test_paged_attention_vs_oracle(ops, name, B, n, H, KH, D, lens, causal, scatter):
    """fp32-softmax attention, bf16 in/out, against a float64 dense reference on the same inputs: every element within
    the forward-error bound (u_P + u_O) * sum_i p_i |v_i| of the bf16-P algorithm (tests/parity_util.py); the achieved
    error (in units of the bound and in bf16 ulps, next to the oracle's) goes to the parity report."""
    cache, indices, indptr, last, max_pages = make_paged(B, lens, KH, D, seed=case_seed(name), scatter=scatter)
    g = torch.Generator().manual_seed(1)
    q = torch.randn(B * n, H, D, generator=g).to(BF)
    qo = torch.arange(B + 1, dtype=torch.int32) * n
    oracle = fr.batch_prefill_paged(q, cache, qo, indices, indptr, last, H, KH, D, causal=causal)
    ref64, bnd = dense_attention_f64(q, cache, qo, indices, indptr, last, H, KH, D, causal=causal)
    ws = ops.AttnWorkspace(DEV)
    out = ops.paged_attention(q.to(DEV), cache.to(DEV), qo.to(DEV), indices.to(DEV), indptr.to(DEV), last.to(DEV), n,
                              max_pages, ws, causal=causal)
    check_attention(name, out, oracle, ref64, bnd)


@pytest.mark.parametrize("name,H,KH,D,ns,lens,fp8", [
    ("verify-ragged-rows", 8, 2, 128, [3, 4, 1], [300, 257, 64], False),
    ("prefill-ragged-rows", 8, 2, 128, [100, 128, 0, 17], [384, 300, 50, 17], False),
    ("prefill-ragged-rows-8waves", 16, 2, 128, [128, 37], [640, 200], False),
    ("prefill-ragged-rows-d64", 32, 8, 64, [128, 5], [256, 130], False),
    ("prefill-ragged-rows-fp8", 8, 2, 128, [100, 128, 17], [384, 300, 17], True),
])
def test_paged_attention_ragged_query_counts(ops, name, H, KH, D, ns, lens, fp8):
    """qo_indptr with a different number of query rows per request (flashinfer allows it; the Engine happens to pass
    equal counts): rows past a request's count must neither be read nor written, requests with 0 rows are skipped."""
    B = len(ns)
    cache, indices, indptr, last, max_pages = make_paged(B, lens, KH, D, seed=11, scatter=True)
    g = torch.Generator().manual_seed(2)
    tot = sum(ns)
    q = torch.randn(tot, H, D, generator=g).to(BF)
    qo = torch.tensor([0] + list(np.cumsum(ns)), dtype=torch.int32)
    scales = None
    ref_cache, dev_cache = cache, cache.to(DEV)
    if fp8:
        ks = 0.02 * (1 + torch.arange(KH, dtype=torch.float32))
        vs = 0.015 * (1 + torch.arange(KH, dtype=torch.float32))
        P, _, ps, _, _ = cache.shape
        c8 = torch.empty(cache.shape, dtype=torch.float8_e4m3fn)
        c8[:, 0] = fr.quantize_fp8(cache[:, 0].reshape(-1, KH, D), ks).view(P, ps, KH, D)
        c8[:, 1] = fr.quantize_fp8(cache[:, 1].reshape(-1, KH, D), vs).view(P, ps, KH, D)
        ref_cache, dev_cache, scales = fr.dequantize_cache_fp8(c8, ks, vs), c8.to(DEV), (ks.to(DEV), vs.to(DEV))
    oracle = fr.batch_prefill_paged(q, ref_cache, qo, indices, indptr, last, H, KH, D, causal=True)
    ref64, bnd = dense_attention_f64(q, ref_cache, qo, indices, indptr, last, H, KH, D, causal=True)
    ws = ops.AttnWorkspace(DEV)
    out = torch.full((tot + 2, H, D), 7.0, dtype=BF, device=DEV)          # 2 guard rows
    ops.paged_attention(q.to(DEV), dev_cache, qo.to(DEV), indices.to(DEV), indptr.to(DEV), last.to(DEV), max(ns),
                        max_pages, ws, causal=True, out=out[:tot], kv_scales=scales)
    assert (out[tot:].float() == 7.0).all(), "wrote past the last query row"
    check_attention(name, out[:tot], oracle, ref64, bnd)


PREFILL_VARIANT_SHAPES = [
    # name, H, KH, D, rows per request, context lengths, page size
    ("d128-many-tiles", 8, 2, 128, [128, 128], [2100, 1500], 128),
    ("d128-g1-uneven-waves", 4, 4, 128, [300, 270], [700, 300], 128),      # waves of a workgroup end 4 tiles apart; idle waves
    ("d128-short", 8, 2, 128, [128, 64], [128, 70], 128),                  # 2 tiles / 1 tile + prologue-only paths
    ("d64-many-tiles", 32, 8, 64, [128, 100], [1300, 900], 128),
    ("d128-page64", 8, 2, 128, [128, 128], [1000, 640], 64),
]
PREFILL_VARIANTS = [0, 32, 64, 128, 129]   # md_debug_set_prefill_mfma32: 16x16 kernel | 32x32 kernel: keys per tile | first V pairing


@pytest.mark.parametrize("knob", PREFILL_VARIANTS)
@pytest.mark.parametrize("name,H,KH,D,ns,lens,page_size", PREFILL_VARIANT_SHAPES, ids=[c[0] for c in PREFILL_VARIANT_SHAPES])
def test_prefill_kernel_variants_vs_oracle(ops, name, H, KH, D, ns, lens, page_size, knob):
    """Every prefill kernel the dispatcher can pick (16x16x32 shared-tile; 32x32x16 at 32 / 64 / 128 keys per tile and
    with either V sub-tile pairing) on shapes that run many tiles, end the waves of one workgroup on different tiles and
    leave whole waves without rows -- same bound as test_paged_attention_vs_oracle.  A variant the shape does not admit (keys per tile not dividing the page, D) falls
    back inside the library; the result must still be right."""
    B = len(ns)
    cache, indices, indptr, last, max_pages = make_paged(B, lens, KH, D, seed=case_seed(name), page_size=page_size,
                                                         scatter=True)
    g = torch.Generator().manual_seed(3)
    tot = sum(ns)
    q = torch.randn(tot, H, D, generator=g).to(BF)
    qo = torch.tensor([0] + list(np.cumsum(ns)), dtype=torch.int32)
    oracle = fr.batch_prefill_paged(q, cache, qo, indices, indptr, last, H, KH, D, causal=True)
    ref64, bnd = dense_attention_f64(q, cache, qo, indices, indptr, last, H, KH, D, causal=True)
    ws = ops.AttnWorkspace(DEV)
    lib = ops._lib.load()
    out = torch.full((tot + 2, H, D), 7.0, dtype=BF, device=DEV)
    lib.md_debug_set_prefill_mfma32(knob)
    try:
        ops.paged_attention(q.to(DEV), cache.to(DEV), qo.to(DEV), indices.to(DEV), indptr.to(DEV), last.to(DEV),
                            max(ns), max_pages, ws, causal=True, out=out[:tot])
        torch.cuda.synchronize()
    finally:
        lib.md_debug_set_prefill_mfma32(-1)
    assert (out[tot:].float() == 7.0).all(), "wrote past the last query row"
    check_attention(f"{name}-knob{knob}", out[:tot], oracle, ref64, bnd)


@pytest.mark.parametrize("page_size", [32, 64, 96])
@pytest.mark.parametrize("D,n,layout,fp8", [(128, 128, "NHD", False), (64, 128, "NHD", False), (128, 4, "NHD", False),
                                            (128, 128, "HND", False), (64, 128, "HND", False), (128, 4, "HND", False),
                                            (128, 128, "NHD", True), (128, 4, "HND", True)])
def test_paged_attention_page_sizes(ops, page_size, D, n, layout, fp8):
    """The C ABI admits any page size that is a multiple of 32 (the reference runs 128): a shared K/V tile of the prefill
    kernels must never cross a page, whatever tile size the dispatcher prefers -- in both page layouts and for bf16 and
    fp8 pages."""
    H, KH, B = 8, 2, 2
    lens = [5 * page_size + 17, 3 * page_size]
    cache, indices, indptr, last, max_pages = make_paged(B, lens, KH, D, seed=page_size + D + n, page_size=page_size,
                                                         scatter=True)
    g = torch.Generator().manual_seed(4)
    q = torch.randn(B * n, H, D, generator=g).to(BF)
    qo = torch.arange(B + 1, dtype=torch.int32) * n
    scales = None
    ref_cache, dev_cache = cache, cache
    if fp8:
        ks = 0.02 * (1 + torch.arange(KH, dtype=torch.float32))
        vs = 0.015 * (1 + torch.arange(KH, dtype=torch.float32))
        P = cache.shape[0]
        c8 = torch.empty(cache.shape, dtype=torch.float8_e4m3fn)
        c8[:, 0] = fr.quantize_fp8(cache[:, 0].reshape(-1, KH, D), ks).view(P, page_size, KH, D)
        c8[:, 1] = fr.quantize_fp8(cache[:, 1].reshape(-1, KH, D), vs).view(P, page_size, KH, D)
        ref_cache, dev_cache, scales = fr.dequantize_cache_fp8(c8, ks, vs), c8, (ks.to(DEV), vs.to(DEV))
    if layout == "HND":
        dev_cache = dev_cache.permute(0, 1, 3, 2, 4).contiguous()
    oracle = fr.batch_prefill_paged(q, ref_cache, qo, indices, indptr, last, H, KH, D, causal=True)
    ref64, bnd = dense_attention_f64(q, ref_cache, qo, indices, indptr, last, H, KH, D, causal=True)
    ws = ops.AttnWorkspace(DEV)
    out = ops.paged_attention(q.to(DEV), dev_cache.to(DEV), qo.to(DEV), indices.to(DEV), indptr.to(DEV), last.to(DEV), n,
                              max_pages, ws, causal=True, kv_scales=scales, kv_layout=layout)
    check_attention(f"page{page_size}-d{D}-n{n}-{layout}{'-fp8' if fp8 else ''}", out, oracle, ref64, bnd)


def test_paged_attention_ignores_garbage_beyond_length(ops):
    """Rows past a request's length may hold NaN/Inf (stale pages): they must not leak into the output."""
    B, n, H, KH, D = 2, 4, 8, 2, 128
    cache, indices, indptr, last, max_pages = make_paged(B, [200, 130], KH, D, seed=3)
    dirty = cache.clone()
    for b, ln in enumerate([200, 130]):
        pg = int(indices[int(indptr[b]) + ln // 128])
        dirty[pg, :, ln % 128:] = float("nan")
    q = torch.randn(B * n, H, D, generator=torch.Generator().manual_seed(7)).to(BF)
    qo = torch.arange(B + 1, dtype=torch.int32) * n
    ws = ops.AttnWorkspace(DEV)
    a = ops.paged_attention(q.to(DEV), cache.to(DEV), qo.to(DEV), indices.to(DEV), indptr.to(DEV), last.to(DEV), n,
                            max_pages, ws)
    b_ = ops.paged_attention(q.to(DEV), dirty.to(DEV), qo.to(DEV), indices.to(DEV), indptr.to(DEV), last.to(DEV), n,
                             max_pages, ws)
    assert torch.equal(bits(a.cpu()), bits(b_.cpu()))


def test_verify_attention_full_size_properties(ops):
    """BASELINE size (one layer of Llama-3.1-8B, B=64, S=16K): size-independent properties --
    (1) softmax weights sum to one: with V == 1 everywhere the output is exactly 1;
    (2) linearity in V: attn(V1+V2) == attn(V1)+attn(V2) up to bf16 rounding;
    (3) split-KV invariance: the same rows computed with a short page table prefix match the oracle."""
    B, n, H, KH, D, S = 64, 4, 32, 8, 128, 16032 + 4
    mp = (S + 127) // 128
    g = torch.Generator(device=DEV).manual_seed(0)
    cache = torch.randn(B * mp, 2, 128, KH, D, device=DEV, generator=g, dtype=torch.float32).to(BF)
    q = torch.randn(B * n, H, D, device=DEV, generator=g, dtype=torch.float32).to(BF)
    indices = torch.arange(B * mp, dtype=torch.int32, device=DEV)
    indptr = torch.arange(B + 1, dtype=torch.int32, device=DEV) * mp
    last = torch.full((B,), S - (mp - 1) * 128, dtype=torch.int32, device=DEV)
    qo = torch.arange(B + 1, dtype=torch.int32, device=DEV) * n
    ws = ops.AttnWorkspace(DEV)
    ones = cache.clone()
    ones[:, 1] = 1.0
    o1 = ops.paged_attention(q, ones, qo, indices, indptr, last, n, mp, ws)
    e1 = (o1.float() - 1.0).abs().max().item()
    parity_report(f"[attn] full-size B=64 S=16K: V==1 -> max |o-1| = {e1:.3e} (bound (u_P+u_O)*1 = {2 ** -7:.3e})")
    assert e1 <= 1.05 * 2 ** -7
    v2 = cache.clone()
    v2[:, 1] = (cache[:, 1].float() * 0.5).to(BF)
    oa = ops.paged_attention(q, cache, qo, indices, indptr, last, n, mp, ws).float()
    ob = ops.paged_attention(q, v2, qo, indices, indptr, last, n, mp, ws).float()
    # halving V is an exact bf16 scaling: every product, partial sum and rounding scales with it -> bit-exact
    assert torch.equal(oa * 0.5, ob), (oa * 0.5 - ob).abs().max().item()
    # request 5 alone against the CPU oracle
    b = 5
    sub = cache[b * mp:(b + 1) * mp].cpu()
    args = (q[b * n:(b + 1) * n].cpu(), sub, torch.tensor([0, n], dtype=torch.int32),
            torch.arange(mp, dtype=torch.int32), torch.tensor([0, mp], dtype=torch.int32), last[b:b + 1].cpu(), H, KH, D)
    oracle = fr.batch_prefill_paged(*args)
    ref64, bnd = dense_attention_f64(*args)
    check_attention("verify-full-size B=64 S=16K (req 5)", oa[b * n:(b + 1) * n].to(BF), oracle, ref64, bnd)


# ----------------------------------------------------------------------------------------- rope / append
@pytest.mark.parametrize("llama31", [False, True])
def test_rope_bit_exact(ops, llama31):
    B, n, H, KH, D = 3, 4, 8, 2, 128
    kw = dict(low_freq_factor=1.0, high_freq_factor=4.0, old_context_len=8192) if llama31 else {}
    tab_ref = fr.rope_table(4096, D, 500000.0, 8.0, **kw)
    tab = ops.RopeTable(4096, D, 500000.0, 8.0, kw.get("low_freq_factor"), kw.get("high_freq_factor"),
                        kw.get("old_context_len"), device=DEV)
    assert torch.equal(tab.table.cpu(), tab_ref), "host table must equal the oracle's float64->float32 table"
    g = torch.Generator().manual_seed(2)
    qkv = torch.randn(B * n, (H + 2 * KH) * D, generator=g).to(BF)       # strided views like the wqkv output
    q = qkv[:, :H * D].unflatten(1, (H, D))
    k = qkv[:, H * D:(H + KH) * D].unflatten(1, (KH, D))
    indptr = torch.arange(B + 1, dtype=torch.int32) * n
    offsets = torch.tensor([0, 1000, 4000], dtype=torch.int32)
    rq, rk = fr.apply_rope(q, k, indptr, offsets, tab_ref)
    dqkv = qkv.to(DEV)
    dq = dqkv[:, :H * D].unflatten(1, (H, D))
    dk = dqkv[:, H * D:(H + KH) * D].unflatten(1, (KH, D))
    oq, ok = ops.rope(dq, dk, indptr.to(DEV), offsets.to(DEV), tab)
    assert torch.equal(bits(oq.cpu()), bits(rq)) and torch.equal(bits(ok.cpu()), bits(rk))


def test_append_and_fused_rope_append_bit_exact(ops):
    B, n, H, KH, D = 3, 4, 8, 2, 64
    lens = [200, 131, 4]        # lengths AFTER the append (page table already includes the new rows)
    cache, indices, indptr, last, _ = make_paged(B, lens, KH, D, seed=9, scatter=True)
    tab_ref = fr.rope_table(2048, D, 10000.0, 1.0)
    tab = ops.RopeTable(2048, D, 10000.0, 1.0, device=DEV)
    g = torch.Generator().manual_seed(4)
    qkv = torch.randn(B * n, (H + 2 * KH) * D, generator=g).to(BF)
    q = qkv[:, :H * D].unflatten(1, (H, D))
    k = qkv[:, H * D:(H + KH) * D].unflatten(1, (KH, D))
    v = qkv[:, (H + KH) * D:].unflatten(1, (KH, D))
    ip = torch.arange(B + 1, dtype=torch.int32) * n
    offsets = torch.tensor([l - n for l in lens], dtype=torch.int32)
    # oracle: rope then append
    rq, rk = fr.apply_rope(q, k, ip, offsets, tab_ref)
    ref_cache = cache.clone()
    fr.append_paged_kv_cache(rk, v, ip, ref_cache, indices, indptr, last)
    d = lambda t: t.to(DEV)
    dqkv = d(qkv)
    dq = dqkv[:, :H * D].unflatten(1, (H, D))
    dk = dqkv[:, H * D:(H + KH) * D].unflatten(1, (KH, D))
    dv = dqkv[:, (H + KH) * D:].unflatten(1, (KH, D))
    # (a) separate ops
    c1 = d(cache)
    oq, ok = ops.rope(dq, dk, d(ip), d(offsets), tab)
    ops.update_kv(ok, dv, d(ip), c1, d(indices), d(indptr), d(last))
    assert torch.equal(bits(c1.cpu()), bits(ref_cache))
    # (b) fused, writing two caches
    c2, c3 = d(cache), d(cache)
    oq2 = ops.rope_append(dq, dk, dv, d(ip), d(offsets), tab, c2, d(indices), d(indptr), d(last), c3, d(indices),
                          d(indptr), d(last))
    assert torch.equal(bits(oq2.cpu()), bits(rq))
    assert torch.equal(bits(c2.cpu()), bits(ref_cache)) and torch.equal(bits(c3.cpu()), bits(ref_cache))


def check_append_overflow(ops, layout, fp8=False):
    """One page mapped, lengths say 130 rows and 4 are appended (rows 126..129): the two rows beyond the page are
    dropped and counted (md_page_overflow_count), rows 126/127 land, no other page is touched; attention over the same
    table clamps the length to the mapped page instead of reading a page that is not there."""
    KH, D, H, n = 2, 64, 8, 4
    d = lambda t: t.to(DEV)
    g = torch.Generator().manual_seed(11)
    k = d(torch.randn(n, KH, D, generator=g).to(BF))
    v = d(torch.randn(n, KH, D, generator=g).to(BF))
    q = d(torch.randn(n, H, D, generator=g).to(BF))
    dt = torch.float8_e4m3fn if fp8 else BF
    scales = (d(torch.tensor([0.05, 0.04])), d(torch.tensor([0.03, 0.06]))) if fp8 else None
    shape = (3, 2, KH, 128, D) if layout == "HND" else (3, 2, 128, KH, D)
    rows = (lambda c, half, sl: c[1, half, :, sl]) if layout == "HND" else (lambda c, half, sl: c[1, half, sl])
    ip = d(torch.tensor([0, n], dtype=torch.int32))
    tabs = (d(torch.tensor([1], dtype=torch.int32)), d(torch.tensor([0, 1], dtype=torch.int32)),
            d(torch.tensor([130], dtype=torch.int32)))
    tab = ops.RopeTable(2048, D, 10000.0, 1.0, device=DEV)
    ops.page_overflow_count(reset=True)
    for fused in (False, True):
        one = torch.zeros(shape, dtype=dt, device=DEV)
        if fused:
            ops.rope_append(q, k, v, ip, d(torch.tensor([126], dtype=torch.int32)), tab, one, *tabs, kv_scales=scales,
                            kv_layout=layout)
        else:
            ops.update_kv(k, v, ip, one, *tabs, kv_scales=scales, kv_layout=layout)
        torch.cuda.synchronize()
        assert ops.page_overflow_count(reset=True) == 2, (layout, fused)
        f = one.float()
        assert f[0].abs().sum().item() == 0 and f[2].abs().sum().item() == 0
        for half in (0, 1):
            assert rows(f, half, slice(0, 126)).abs().sum().item() == 0
            assert (rows(f, half, slice(126, 128)).abs().sum(-1) > 0).all()
    assert ops.page_overflow_count(reset=False) == 0
    ws = ops.AttnWorkspace(DEV)
    qo = d(torch.tensor([0, n], dtype=torch.int32))
    o130 = ops.paged_attention(q, one, qo, *tabs, n, 1, ws, kv_scales=scales, kv_layout=layout)
    o128 = ops.paged_attention(q, one, qo, tabs[0], tabs[1], d(torch.tensor([128], dtype=torch.int32)), n, 1, ws,
                               kv_scales=scales, kv_layout=layout)
    assert not torch.isnan(o130.float()).any()
    assert torch.equal(bits(o130.cpu()), bits(o128.cpu()))


def test_append_beyond_mapped_pages_is_dropped_and_counted(ops):
    check_append_overflow(ops, "NHD")


# ----------------------------------------------------------------------------------------- small fused ops
def _ulp_close(a, b, ulps=1):
    """bf16 tensors equal up to `ulps` units in the last place."""
    ia, ib = bits(a).int(), bits(b).int()
    # map sign-magnitude to a monotonic integer line
    ia = torch.where(ia < 0, -(ia & 0x7fff), ia)
    ib = torch.where(ib < 0, -(ib & 0x7fff), ib)
    return (ia - ib).abs().max().item() <= ulps


@pytest.mark.parametrize("dim", [512, 2048, 4096, 8192])
def test_rmsnorm_and_add_rmsnorm(ops, dim):
    """<= 1 bf16 ulp vs the oracle (fp32 reduction order differs); the residual sum h is bit-exact."""
    g = torch.Generator().manual_seed(dim)
    x = torch.randn(37, dim, generator=g).to(BF)
    r = torch.randn(37, dim, generator=g).to(BF)
    w = (1 + 0.1 * torch.randn(dim, generator=g)).to(BF)
    ref = mr.rmsnorm(x, w, 1e-5)
    y = ops.rmsnorm(x.to(DEV), w.to(DEV), 1e-5).cpu()
    assert _ulp_close(y, ref)
    h_ref = x + r
    h, y2 = ops.add_rmsnorm(x.to(DEV), r.to(DEV), w.to(DEV), 1e-5)
    assert torch.equal(bits(h.cpu()), bits(h_ref))
    assert _ulp_close(y2.cpu(), mr.rmsnorm(h_ref, w, 1e-5))


def test_silu_mul(ops):
    g = torch.Generator().manual_seed(5)
    ab = (torch.randn(33, 2 * 1024, generator=g) * 3).to(BF)
    a, b = ab[:, :1024], ab[:, 1024:]
    ref = torch.nn.functional.silu(a) * b
    dab = ab.to(DEV)
    y = ops.silu_mul(dab[:, :1024], dab[:, 1024:]).cpu()
    assert _ulp_close(y, ref)


@pytest.mark.parametrize("vocab", [16032, 128256, 32003])      # 256-thread, 1024-thread, unaligned-row kernels
def test_argmax_lowest_index_and_tp_merge(ops, vocab):
    g = torch.Generator().manual_seed(6)
    logits = torch.randn(9, vocab, generator=g).to(BF)
    logits[3, 100] = logits[3].max()
    logits[3, 7000] = logits[3, 100]                   # exact tie -> lowest index
    logits[4, vocab - 1] = logits[4].max() + 1         # maximum in the last element (tail loop)
    logits[5] = 0                                      # all equal -> index 0
    logits[6, vocab - 3] = logits[6].max()             # tie between an early wave's element and the tail
    ref = torch.tensor([int(torch.nonzero(row == row.max())[0]) for row in logits.float()])
    vals, idx = ops.argmax(logits.to(DEV), index_offset=16032 * 2, return_values=True)
    assert torch.equal(idx.cpu(), ref + 16032 * 2)
    assert torch.equal(bits(vals.cpu()), bits(logits.float().max(dim=-1).values.to(BF)))
    tpv = torch.randn(9, 8, generator=g).to(BF)
    tpv[2, 1] = tpv[2].max()
    tpv[2, 6] = tpv[2, 1]
    tpi = torch.randint(0, 100000, (9, 8), generator=g)
    out = ops.tp_argmax_merge(tpv.to(DEV), tpi.to(DEV)).cpu()
    assert torch.equal(out, mr.tp_argmax_merge(tpv, tpi))
    # round 5: the one-hot-slot form (what the reference builds with zeros + an index assignment before its all-reduces)
    for rank, world in ((0, 1), (2, 3), (7, 8)):
        sv, si = ops.argmax_tp_slots(logits.to(DEV), rank, world, index_offset=rank * vocab)
        want_v = torch.zeros(9, world, dtype=BF)
        want_i = torch.zeros(9, world, dtype=torch.long)
        want_v[:, rank] = logits.float().max(dim=-1).values.to(BF)
        want_i[:, rank] = ref + rank * vocab
        assert torch.equal(bits(sv.cpu()), bits(want_v)) and torch.equal(si.cpu(), want_i), (rank, world)


# ----------------------------------------------------------------------------------------- accept loop
VARIANTS = {"longspec": (lambda g: g, lambda g: g, True, "draft_"),
            "selfspec_snapkv": (lambda g: g + 1, lambda g: g + 1, False, "engine_draft_"),
            "selfspec_stream": (lambda g: g, lambda g: g, True, "engine_draft_")}


@pytest.mark.parametrize("fixture", ["accept_loop.json", "accept_loop_fuzz.json"])
@pytest.mark.parametrize("variant", list(VARIANTS))
def test_accept_rollback_kernel_matches_reference_loop_body(ops, variant, fixture):
    """Bit-exact against vectors produced by exec'ing the reference's own loop body (gen_golden.py:accept_loop;
    accept_loop_fuzz: unstructured cases, gamma up to 6, up to 130 rows = more than one wavefront of requests)."""
    dr, cap, dbl, pre = VARIANTS[variant]
    for c in gc.load_json(fixture)[variant]:
        i, o = c["inp"], c["out"]
        G, B = i["gamma"], i["B"]
        d = lambda x, dt: torch.tensor(x, dtype=dt, device=DEV)
        tb, tt = d(i["tokens_buffer"], torch.long), d(i["target_tokens"], torch.long)
        output = torch.zeros(B, i["out_cols"], dtype=torch.long, device=DEV)
        nn_ = d(i["num_nodes"], torch.long)
        cl, lp = d(i["cachelens"], torch.int32), d(i["last_page_len"], torch.int32)
        dcl, dlp = d(i[pre + "cachelens"], torch.int32), d(i[pre + "last_page_len"], torch.int32)
        an, bo = torch.zeros(B, dtype=torch.long, device=DEV), torch.zeros(B, dtype=torch.long, device=DEV)
        db, cu = torch.zeros(B, 2, dtype=torch.long, device=DEV), torch.zeros(B, dtype=torch.long, device=DEV)
        fl = torch.zeros(2, dtype=torch.int32, device=DEV)
        ops.accept_rollback(tb, tt, output, nn_, cl, lp, dcl, dlp, G, dr(G), cap(G), i["eot_1"], i["eot_2"],
                            i["prefix"] + 80, an, bo, db if dbl else None, cu if dbl else None, fl)
        assert bool(fl[0]) == o["terminal"]
        assert an.tolist() == o["accept_nums"] and bo.tolist() == o["bonus"]
        assert tb.tolist() == o["tokens_buffer"]
        assert cl.tolist() == o["cachelens"] and lp.tolist() == o["last_page_len"]
        assert dcl.tolist() == o[pre + "cachelens"] and dlp.tolist() == o[pre + "last_page_len"]
        assert nn_.tolist() == o["num_nodes"]
        out_np = output.cpu().numpy()
        nz = np.nonzero(out_np)
        assert [[int(a), int(b)] for a, b in zip(*nz)] == o["output_nz"] and out_np[nz].tolist() == o["output_vals"]
        assert bool(fl[1]) == o["next_double"]
        if o["next_double"]:
            assert db.tolist() == o["double_buffer"] and cu.tolist() == o["cachelens_update"]


# ----------------------------------------------------------------------------------------- StreamingLLM eviction
def test_streaming_shift_and_rotate_match_reference_cache_bytes(ops, golden_dir):
    """The in-place shift + rotate reproduces the reference KVCache.prefill's cache bytes chunk by chunk."""
    z = np.load(f"{golden_dir}/stream_prefill.npz")
    B, KH, D, budget, ppr = [int(x) for x in z["meta"]]
    tab = ops.RopeTable(1024, D, 10000.0, 1.0, device=DEV)
    cache = torch.zeros(B * ppr, 2, 128, KH, D, dtype=BF, device=DEV)
    rot = torch.empty_like(cache)
    for step in range(int(z["nsteps"][0])):
        ctx, n, is_last, npr, last = [int(x) for x in z[f"info{step}"]]
        k = gc.from_bits(z[f"k{step}"]).to(DEV)
        v = gc.from_bits(z[f"v{step}"]).to(DEV)
        if ctx + n <= budget:
            indices = torch.cat([torch.arange(i * ppr, i * ppr + npr, dtype=torch.int32) for i in range(B)]).to(DEV)
            indptr = (torch.arange(B + 1) * npr).to(torch.int32).to(DEV)
            ops.update_kv(k, v, (torch.arange(B + 1) * n).to(torch.int32).to(DEV), cache, indices, indptr,
                          torch.full((B,), last, dtype=torch.int32, device=DEV))
            valid = ctx + n
        else:
            ops.streaming_shift_append(k, v, cache, n, budget, 16, ppr)
            valid = budget
        overflow_last = (ctx + n > budget) and is_last
        dst = cache if overflow_last else rot
        if not overflow_last:
            rot.copy_(cache)     # rows >= valid of the clone are the cache's (the reference clones everything)
        ops.streaming_rotate(cache, dst, B, valid, ppr, tab)
        assert torch.equal(bits(cache.cpu()), bits(gc.from_bits(z[f"cache{step}"]))), f"cache step {step}"
        assert torch.equal(bits(dst.cpu()), bits(gc.from_bits(z[f"rot{step}"]))), f"rot step {step}"


def test_streaming_shift_and_rotate_at_budget_513_match_reference_digests(ops):
    """The same at BASELINE configs[3]'s draft budget (513 rows = 5 pages per request, two kv heads): the eviction moves
    rows across page boundaries.  Expected values: SHA-256 digests of the REAL reference's cache / rotated cache after
    every chunk (tests/golden/stream_prefill_b513.json; inputs regenerated from its seed)."""
    import hashlib
    j = gc.load_json("stream_prefill_b513.json")
    m = j["meta"]
    B, KH, D, budget, ppr = m["B"], m["KH"], m["D"], m["budget"], m["ppr"]
    tab = ops.RopeTable(m["rope_positions"], D, 10000.0, 1.0, device=DEV)
    sha = lambda t: hashlib.sha256(t.cpu().contiguous().view(torch.int16).numpy().tobytes()).hexdigest()
    g = torch.Generator().manual_seed(m["seed"])
    cache = torch.zeros(B * ppr, 2, 128, KH, D, dtype=BF, device=DEV)
    rot = torch.empty_like(cache)
    for i, st in enumerate(j["steps"]):
        ctx, n, is_last, npr, last = st["ctx"], st["n"], st["is_last"], st["npr"], st["last"]
        k = torch.randn(B * n, KH, D, generator=g).to(BF).to(DEV)
        v = torch.randn(B * n, KH, D, generator=g).to(BF).to(DEV)
        if ctx + n <= budget:
            indices = torch.cat([torch.arange(r * ppr, r * ppr + npr, dtype=torch.int32) for r in range(B)]).to(DEV)
            indptr = (torch.arange(B + 1) * npr).to(torch.int32).to(DEV)
            ops.update_kv(k, v, (torch.arange(B + 1) * n).to(torch.int32).to(DEV), cache, indices, indptr,
                          torch.full((B,), last, dtype=torch.int32, device=DEV))
            valid = ctx + n
        else:
            ops.streaming_shift_append(k, v, cache, n, budget, 16, ppr)
            valid = budget
        overflow_last = (ctx + n > budget) and bool(is_last)
        dst = cache if overflow_last else rot
        if not overflow_last:
            rot.copy_(cache)
        ops.streaming_rotate(cache, dst, B, valid, ppr, tab)
        assert sha(cache) == st["cache_sha256"], f"cache after chunk {i}"
        assert sha(dst) == st["rot_sha256"], f"rotated cache after chunk {i}"


# ----------------------------------------------------------------------------------------- SnapKV select
def _snapkv_alt_oracle(q, k, v, g, W, budget):
    """The oracle with correctly rounded (float64-accumulated) QK^T scores: a second valid implementation of the
    reference's arithmetic with another summation order.  Returns (scores [B,KH,S-W] bf16, idx [B,KH,topk])."""
    old = mr.LINEAR_MODE
    mr.LINEAR_MODE = "fp64"
    try:
        sc, idx = [], []
        for b in range(q.shape[0] // W):
            i, _, _, s_ = mr.snapkv_select(q[b * W:(b + 1) * W], k[b], v[b], g, W, budget)
            sc.append(s_)
            idx.append(i)
    finally:
        mr.LINEAR_MODE = old
    return torch.stack(sc), torch.stack(idx)


@pytest.mark.parametrize("tag", ["g4", "g5", "g8", "g4d128", "g4s3104",
                                 "g4w16"])
def test_snapkv_select_vs_reference_fixture(ops, tag, golden_dir):
    """Against the reference's own gen_draft_kv output (fixture).  The index work is exact GIVEN the scores (stable
    descending top-k, lowest index among equals; gathered rows bit-equal).  The bf16 scores themselves depend on the
    fp32 summation order inside QK^T (oneDNN's on the reference's CPU run, the MFMA's here): a product that lands on
    a bf16 rounding boundary flips one ulp and the flip propagates through softmax / pooling.  That noise is
    MEASURED, not assumed: the oracle re-run with correctly rounded (float64) QK^T is a second valid implementation,
    and the HIP kernel must sit no further from the reference than twice that yardstick --
      * fraction of pooled scores differing from the reference:  hip <= 2 * alt + 0.2 %,  never more than 2 ulp;
      * selected-index set per (request, kv head): |hip ^ ref| <= 2 * max|alt ^ ref| + 2, and every differing
        position's reference score lies within 2 ulp of the reference's selection threshold.
    All counts go to the parity report."""
    # g4s3104: a context of three 1024-column score chunks (per-chunk softmax statistics, combined), budget 257;
    # g4w16: --window_size 16 instead of the default 32
    z = np.load(f"{golden_dir}/snapkv_select_long.npz" if tag in ("g4s3104", "g4w16") else f"{golden_dir}/snapkv_select.npz")
    g, KH, D, S, budget, B, W = [int(x) for x in z[f"{tag}_meta"]]
    H = g * KH
    q = gc.from_bits(z[f"{tag}_q"])
    k = gc.from_bits(z[f"{tag}_k"])
    v = gc.from_bits(z[f"{tag}_v"])
    ref_scores = gc.from_bits(z[f"{tag}_scores"])
    ref_idx = torch.from_numpy(z[f"{tag}_idx"])
    npg = (S + 127) // 128
    cache = torch.zeros(B * npg, 2, 128, KH, D, dtype=BF)
    for b in range(B):
        kk = torch.zeros(npg * 128, KH, D, dtype=BF)
        vv = torch.zeros(npg * 128, KH, D, dtype=BF)
        kk[:S], vv[:S] = k[b], v[b]
        cache[b * npg:(b + 1) * npg, 0] = kk.view(npg, 128, KH, D)
        cache[b * npg:(b + 1) * npg, 1] = vv.view(npg, 128, KH, D)
    dppr = budget // 128 + 1
    dcache = torch.zeros(B * dppr, 2, 128, KH, D, dtype=BF, device=DEV)
    ws = ops.AttnWorkspace(DEV)
    idx, sc = ops.snapkv_select(q.to(DEV), cache.to(DEV), torch.arange(B * npg, dtype=torch.int32, device=DEV),
                                (torch.arange(B + 1, dtype=torch.int32) * npg).to(DEV), S, W, budget, 5, dcache,
                                torch.arange(B * dppr, dtype=torch.int32, device=DEV),
                                (torch.arange(B + 1, dtype=torch.int32) * dppr).to(DEV),
                                torch.ones(B, dtype=torch.int32, device=DEV), ws, return_scores=True)
    idx, sc = idx.cpu().long(), sc.cpu()
    topk = budget - W
    alt_sc, alt_idx = _snapkv_alt_oracle(q, k, v, g, W, budget)
    mism_hip = 1.0 - (bits(sc) == bits(ref_scores)).float().mean().item()
    mism_alt = 1.0 - (bits(alt_sc) == bits(ref_scores)).float().mean().item()
    assert _ulp_close(sc, ref_scores, ulps=2)
    assert mism_hip <= 2 * mism_alt + 2e-3, (mism_hip, mism_alt)
    dk = dcache.cpu()
    nd_hip, nd_alt = [], []
    for b in range(B):
        for h in range(KH):
            s = sc[b, h].float()
            mine = idx[b, h]
            assert torch.equal(mine, torch.sort(s, descending=True, stable=True).indices[:topk]), "order / tie-break"
            theirs = set(ref_idx[b, h].tolist())
            diff = set(mine.tolist()) ^ theirs
            nd_hip.append(len(diff))
            nd_alt.append(len(set(alt_idx[b, h].tolist()) ^ theirs))
            thr = ref_scores[b, h].float()[ref_idx[b, h]].min()
            for p in diff:
                assert abs(ref_scores[b, h, p].float() - thr) <= 2 * thr * 2 ** -8 + 1e-30, (b, h, p)
            # gathered rows: draft slots [0,topk) = K/V at our indices, then the last W positions
            rows_k = dk[b * dppr:(b + 1) * dppr, 0].reshape(-1, KH, D)[:budget, h]
            rows_v = dk[b * dppr:(b + 1) * dppr, 1].reshape(-1, KH, D)[:budget, h]
            assert torch.equal(bits(rows_k[:topk]), bits(k[b][mine, h]))
            assert torch.equal(bits(rows_v[:topk]), bits(v[b][mine, h]))
            assert torch.equal(bits(rows_k[topk:]), bits(k[b][S - W:, h]))
            assert torch.equal(bits(rows_v[topk:]), bits(v[b][S - W:, h]))
    parity_report(f"[snapkv] {tag:7s} scores != reference: hip {100 * mism_hip:.3f}%  fp64-oracle {100 * mism_alt:.3f}%"
                  f"  | index-set symmetric difference per (b,kvh) of top-{topk}: hip max {max(nd_hip)} "
                  f"sum {sum(nd_hip)}  fp64-oracle max {max(nd_alt)} sum {sum(nd_alt)}  ({B * KH} sets)")
    assert max(nd_hip) <= 2 * max(nd_alt) + 2, (nd_hip, nd_alt)


@pytest.mark.parametrize("case", ["ints2", "ints4", "ints8", "tiny", "mixed"])
def test_snapkv_select_every_score_magnitude_bit_exact(ops, case):
    """The select's softmax over the WHOLE range of unscaled scores (the table of exponentials used
    to end at |s| < 16).  q and k hold small integers (x a power of two), so every q.k is an integer computed exactly
    in ANY summation order: the kernel's bf16 scores equal the oracle's bit for bit, and the rest of the pipeline --
    correctly rounded bf16(exp(s - M) / Z), group sums, accumulation, pooling -- must then be BIT-EXACT against the
    float64 oracle, with no summation-order yardstick in between:
      ints2: |entries| <= 2 -> scores ~ N(0, 16^2), a third beyond the old table's 16, all inside the new one;
      ints4: scores ~ N(0, 53^2) (a trained checkpoint's unscaled range), the largest beyond 256;
      ints8: scores ~ N(0, 190^2): a good part beyond the table's 512 -> table and exp_neg() paths mixed in a tile;
      tiny:  entries in {-1, 0, 1} * 2^-10 -> |scores| < 2^-12 and many exact zeros: below the table;
      mixed: ints2 keys with every 7th key row x 16 and every 5th x 2^-12 (all regimes inside one row's softmax)."""
    B, KH, g, D, S, W, budget = 2, 2, 4, 64, 640, 32, 129
    H = KH * g
    gen = torch.Generator().manual_seed({"ints2": 1, "ints4": 2, "ints8": 3, "tiny": 4, "mixed": 5}[case])
    amp = {"ints2": 2, "ints4": 4, "ints8": 8, "tiny": 1, "mixed": 2}[case]
    q = torch.randint(-amp, amp + 1, (B * W, H, D), generator=gen).float()
    k = torch.randint(-amp, amp + 1, (B, S, KH, D), generator=gen).float()
    if case == "tiny":
        q, k = q * 2.0 ** -5, k * 2.0 ** -15
    if case == "mixed":
        k[:, ::7] *= 16.0
        k[:, ::5] *= 2.0 ** -12
    v = torch.randn(B, S, KH, D, generator=gen)
    q, k, v = q.to(BF), k.to(BF), v.to(BF)
    npg = (S + 127) // 128
    cache = torch.zeros(B * npg, 2, 128, KH, D, dtype=BF)
    for b in range(B):
        cache[b * npg:(b + 1) * npg, 0] = k[b].reshape(npg, 128, KH, D)
        cache[b * npg:(b + 1) * npg, 1] = v[b].reshape(npg, 128, KH, D)
    dppr = budget // 128 + 1
    dcache = torch.zeros(B * dppr, 2, 128, KH, D, dtype=BF, device=DEV)
    ws = ops.AttnWorkspace(DEV)
    idx, sc = ops.snapkv_select(q.to(DEV), cache.to(DEV), torch.arange(B * npg, dtype=torch.int32, device=DEV),
                                (torch.arange(B + 1, dtype=torch.int32) * npg).to(DEV), S, W, budget, 5, dcache,
                                torch.arange(B * dppr, dtype=torch.int32, device=DEV),
                                (torch.arange(B + 1, dtype=torch.int32) * dppr).to(DEV),
                                torch.ones(B, dtype=torch.int32, device=DEV), ws, return_scores=True)
    idx, sc = idx.cpu().long(), sc.cpu()
    alt_sc, alt_idx = _snapkv_alt_oracle(q, k, v, g, W, budget)
    raw = torch.einsum("whd,shd->hws", q[:W].float().view(W, KH, g, D)[:, :, 0], k[0].float())
    neq = int((bits(sc) != bits(alt_sc)).sum())
    parity_report(f"[snapkv] magnitudes/{case:6s} unscaled scores: std {raw.std().item():.3g}, max |s| {raw.abs().max().item():.3g}, "
                  f"{100 * (raw.abs() >= 16).float().mean().item():.1f} % beyond 16, {100 * (raw.abs() >= 512).float().mean().item():.1f} % "
                  f"beyond 512, {100 * (raw.abs() < 2.0 ** -12).float().mean().item():.1f} % below 2^-12 | pooled scores != float64 "
                  f"oracle: {neq} of {sc.numel()}")
    assert not torch.isnan(sc.float()).any()
    assert neq == 0, (case, neq)
    topk = budget - W
    for b in range(B):
        for h in range(KH):
            assert torch.equal(idx[b, h], torch.sort(sc[b, h].float(), descending=True, stable=True).indices[:topk])
            assert torch.equal(idx[b, h], alt_idx[b, h])       # equal scores + the same stable tie rule


def test_snapkv_select_full_size_properties(ops):
    """SnapKV select at the BASELINE context length (S=16032, window 32, budget 257; 1B-draft head geometry): the
    oracle is too slow at this size, so the size-independent properties are asserted instead -- indices unique and
    inside [0, S-W); exactly the stable descending top-k of the kernel's own pooled scores (order and tie-break);
    draft-cache rows bit-equal to the source rows at those indices followed by the last W rows; run-to-run
    deterministic; stale bytes beyond S never selected."""
    B, KH, g, D, S, W, budget = 4, 2, 4, 64, 16032, 32, 257
    H = KH * g
    npg = (S + 127) // 128
    gen = torch.Generator(device=DEV).manual_seed(5)
    cache = torch.randn(B * npg, 2, 128, KH, D, device=DEV, generator=gen, dtype=torch.float32).to(BF)
    # poison the slots past S in every request's last page: they must never be read as candidates
    tail = S - (npg - 1) * 128
    for b in range(B):
        cache[(b + 1) * npg - 1, :, tail:] = float("nan")
    q = (torch.randn(B * W, H, D, device=DEV, generator=gen, dtype=torch.float32) * 0.3).to(BF)
    dppr = budget // 128 + 1
    indices = torch.arange(B * npg, dtype=torch.int32, device=DEV)
    indptr = (torch.arange(B + 1, dtype=torch.int32) * npg).to(DEV)
    dind = torch.arange(B * dppr, dtype=torch.int32, device=DEV)
    dptr = (torch.arange(B + 1, dtype=torch.int32) * dppr).to(DEV)
    dlast = torch.ones(B, dtype=torch.int32, device=DEV)
    ws = ops.AttnWorkspace(DEV)
    runs = []
    for _ in range(2):
        dcache = torch.zeros(B * dppr, 2, 128, KH, D, dtype=BF, device=DEV)
        idx, sc = ops.snapkv_select(q, cache, indices, indptr, S, W, budget, 5, dcache, dind, dptr, dlast, ws,
                                    return_scores=True)
        runs.append((idx.cpu().long(), sc.cpu(), dcache.cpu()))
    (idx, sc, dk), (idx2, sc2, dk2) = runs
    assert torch.equal(idx, idx2) and torch.equal(bits(sc), bits(sc2)) and torch.equal(bits(dk), bits(dk2))
    assert not torch.isnan(sc.float()).any()
    topk = budget - W
    src = cache.cpu()
    for b in range(B):
        k_all = src[b * npg:(b + 1) * npg, 0].reshape(-1, KH, D)
        v_all = src[b * npg:(b + 1) * npg, 1].reshape(-1, KH, D)
        for h in range(KH):
            mine = idx[b, h]
            assert mine.min() >= 0 and mine.max() < S - W and len(set(mine.tolist())) == topk
            want = torch.sort(sc[b, h].float(), descending=True, stable=True).indices[:topk]
            assert torch.equal(mine, want), "not the stable descending top-k of the pooled scores"
            rows_k = dk[b * dppr:(b + 1) * dppr, 0].reshape(-1, KH, D)[:budget, h]
            rows_v = dk[b * dppr:(b + 1) * dppr, 1].reshape(-1, KH, D)[:budget, h]
            assert torch.equal(bits(rows_k[:topk]), bits(k_all[mine, h]))
            assert torch.equal(bits(rows_v[:topk]), bits(v_all[mine, h]))
            assert torch.equal(bits(rows_k[topk:]), bits(k_all[S - W:S, h]))
            assert torch.equal(bits(rows_v[topk:]), bits(v_all[S - W:S, h]))
