"""Randomised parity sweep of the hot-path kernels through the C ABI (run with -m gpu).

The hand-picked cases of test_gpu_ops.py / test_gpu_gemm.py / test_gpu_fused.py cover the shapes the Engine passes and the
edges we thought of; this file draws shapes nobody thought of.  Every case is derived from a fixed seed (the sweep is
deterministic and a failure names its seed); MAGICDEC_FUZZ_CASES=<n> widens every sweep to n cases (default below: the whole
file runs in well under a minute; one wide run: profiles/r06_fuzz_11000_cases.txt).

* paged attention: request count, ragged query-row counts (0 rows included), head grouping g in {1,2,3,4,5,7,8}, D, context
  lengths from 0 to a few thousand rows, page size in {32, 64, 128}, scattered page tables, NHD / HND pages, bf16 / fp8
  pages, causal or not -- gate of test_gpu_ops.py::test_paged_attention_vs_oracle (float64 dense reference, bf16-P bound);
* RoPE + paged append (separate ops and the fused launch, one or two caches, both layouts): BIT-EXACT against the oracle;
* the four GEMM families on random (M, N, K) inside their *_supported ranges, strided x, optional bias: the float64 gate of
  test_gpu_gemm.py; the residual epilogue of the tile kernel and the split combine BIT-EXACT against the unfused sequence;
* argmax with planted ties: BIT-EXACT, lowest index;
* SnapKV select on integer-valued inputs (scores exact in any summation order): scores, indices, gathered rows BIT-EXACT;
* StreamingLLM eviction + re-rotation chunk by chunk, and the fused accept / rollback kernel on random batches: BIT-EXACT
  against the oracle functions that tests/golden pins to the real reference.
"""
import os
import random

import pytest
import torch

from oracle import flashinfer_ref as fr
from tests.conftest import parity_report
from tests.parity_util import check_attention, dense_attention_f64
from tests.test_gpu_ops import _ulp_close, bits, make_paged

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16
N_CASES = int(os.environ.get("MAGICDEC_FUZZ_CASES", "0"))


def cases(default):
    return list(range(N_CASES or default))


@pytest.fixture(scope="module")
def ops():
    from magicdec_amd import ops as _ops
    _ops._lib.load()      # fail loudly if the HIP library is missing
    return _ops


def d(t):
    return t.to(DEV) if t is not None else None


# ----------------------------------------------------------------------------------------- attention
def draw_attention_case(seed):
    r = random.Random(9000 + seed)
    D = r.choice([64, 128, 128])
    g = r.choice([1, 2, 3, 4, 4, 5, 7, 8])
    KH = r.choice([1, 2, 2, 4, 8])
    H = g * KH
    B = r.randint(1, 5)
    page_size = r.choice([32, 64, 128, 128])
    mode = r.choice(["decode", "decode", "verify", "verify", "ragged", "prefill"])
    ns, lens = [], []
    for _ in range(B):
        if mode == "decode":
            n = 1
        elif mode == "verify":
            n = r.choice([2, 3, 4, 5, 8])
        elif mode == "ragged":
            n = r.choice([0, 1, 2, 4, 7, 33, 100, 128])
        else:
            n = r.choice([32, 100, 128, 130, 257])
        top = r.choice([40, 300, 1200, 3000]) if mode != "prefill" else r.choice([300, 900])
        ln = n + r.randint(0, top) if r.random() > 0.1 else n      # sometimes the request is exactly its own rows
        ns.append(n)
        lens.append(ln)
    if mode in ("decode", "verify"):
        ns = [ns[0]] * B                                           # the Engine's equal counts; ragged mode covers the rest
        lens = [max(l, ns[0]) for l in lens]
    if mode == "decode" and r.random() < 0.3:
        lens[r.randrange(B)] = 8000 + r.randint(0, 500)            # long enough for the split-KV path
    fp8 = D == 128 and r.random() < 0.3
    layout = r.choice(["NHD", "HND"])
    causal = mode != "decode" or r.random() < 0.7
    if sum(ns) == 0:
        ns[0], lens[0] = 1, max(lens[0], 1)
    return dict(D=D, H=H, KH=KH, B=B, page_size=page_size, ns=ns, lens=lens, fp8=fp8, layout=layout,
                scatter=r.random() < 0.6, causal=causal, mode=mode)


@pytest.mark.parametrize("seed", cases(28))
def test_fuzz_paged_attention(ops, seed):
    c = draw_attention_case(seed)
    D, H, KH, B, ps = c["D"], c["H"], c["KH"], c["B"], c["page_size"]
    cache, indices, indptr, last, max_pages = make_paged(B, c["lens"], KH, D, seed=seed, page_size=ps,
                                                         scatter=c["scatter"])
    tot = sum(c["ns"])
    g = torch.Generator().manual_seed(100 + seed)
    q = torch.randn(max(tot, 1), H, D, generator=g).to(BF)[:tot]
    qo = torch.tensor([0] + [sum(c["ns"][:i + 1]) for i in range(B)], dtype=torch.int32)
    scales, ref_cache, dev_cache = None, cache, cache
    if c["fp8"]:
        ks = 0.02 * (1 + torch.arange(KH, dtype=torch.float32))
        vs = 0.015 * (1 + torch.arange(KH, dtype=torch.float32))
        P = cache.shape[0]
        c8 = torch.empty(cache.shape, dtype=torch.float8_e4m3fn)
        c8[:, 0] = fr.quantize_fp8(cache[:, 0].reshape(-1, KH, D), ks).view(P, ps, KH, D)
        c8[:, 1] = fr.quantize_fp8(cache[:, 1].reshape(-1, KH, D), vs).view(P, ps, KH, D)
        ref_cache, dev_cache, scales = fr.dequantize_cache_fp8(c8, ks, vs), c8, (ks.to(DEV), vs.to(DEV))
    if c["layout"] == "HND":
        dev_cache = dev_cache.permute(0, 1, 3, 2, 4).contiguous()
    args = (q, ref_cache, qo, indices, indptr, last, H, KH, D)
    oracle = fr.batch_prefill_paged(*args, causal=c["causal"])
    ref64, bnd = dense_attention_f64(*args, causal=c["causal"])
    ws = ops.AttnWorkspace(DEV)
    out = torch.full((tot + 2, H, D), 7.0, dtype=BF, device=DEV)          # 2 guard rows
    ops.paged_attention(d(q), d(dev_cache), d(qo), d(indices), d(indptr), d(last), max(max(c["ns"]), 1), max_pages, ws,
                        causal=c["causal"], out=out[:tot], kv_scales=scales, kv_layout=c["layout"])
    assert (out[tot:].float() == 7.0).all(), f"seed {seed}: wrote past the last query row ({c})"
    if tot:
        check_attention(f"fuzz-{seed} {c['mode']} B{B} g{H // KH} KH{KH} D{D} page{ps} {c['layout']}"
                        f"{' fp8' if c['fp8'] else ''}{'' if c['causal'] else ' non-causal'} n={c['ns']} L={c['lens']}",
                        out[:tot], oracle, ref64, bnd)


# ----------------------------------------------------------------------------------------- rope + append
@pytest.mark.parametrize("seed", cases(16))
def test_fuzz_rope_append_bit_exact(ops, seed):
    r = random.Random(7000 + seed)
    D = r.choice([64, 128])
    KH = r.choice([1, 2, 4, 8])
    H = KH * r.choice([1, 2, 4, 5, 8])
    B = r.randint(1, 6)
    n = r.choice([1, 1, 2, 4, 5, 128])
    ps = r.choice([32, 64, 128, 128])
    lens = [n + r.randint(0, 700) for _ in range(B)]              # lengths AFTER the append
    layout = r.choice(["NHD", "HND"])
    cache, indices, indptr, last, _ = make_paged(B, lens, KH, D, seed=seed, page_size=ps, scatter=r.random() < 0.7)
    theta = r.choice([10000.0, 500000.0])
    kw = dict(low_freq_factor=1.0, high_freq_factor=4.0, old_context_len=8192) if r.random() < 0.5 else {}
    tab_ref = fr.rope_table(2048, D, theta, 8.0 if kw else 1.0, **kw)
    tab = ops.RopeTable(2048, D, theta, 8.0 if kw else 1.0, kw.get("low_freq_factor"), kw.get("high_freq_factor"),
                        kw.get("old_context_len"), device=DEV)
    g = torch.Generator().manual_seed(300 + seed)
    qkv = torch.randn(B * n, (H + 2 * KH) * D, generator=g).to(BF)       # strided views like the wqkv output
    q = qkv[:, :H * D].unflatten(1, (H, D))
    k = qkv[:, H * D:(H + KH) * D].unflatten(1, (KH, D))
    v = qkv[:, (H + KH) * D:].unflatten(1, (KH, D))
    ip = torch.arange(B + 1, dtype=torch.int32) * n
    offsets = torch.tensor([l - n for l in lens], dtype=torch.int32)
    rq, rk = fr.apply_rope(q, k, ip, offsets, tab_ref)
    ref_cache = cache.clone()
    fr.append_paged_kv_cache(rk, v, ip, ref_cache, indices, indptr, last)
    to_layout = (lambda c: c.permute(0, 1, 3, 2, 4).contiguous()) if layout == "HND" else (lambda c: c)
    want = bits(to_layout(ref_cache))
    dqkv = d(qkv)
    dq = dqkv[:, :H * D].unflatten(1, (H, D))
    dk = dqkv[:, H * D:(H + KH) * D].unflatten(1, (KH, D))
    dv = dqkv[:, (H + KH) * D:].unflatten(1, (KH, D))
    tabs = (d(indices), d(indptr), d(last))
    tag = f"seed {seed}: B{B} n{n} H{H} KH{KH} D{D} page{ps} {layout}"
    c1 = d(to_layout(cache))
    oq, ok_ = ops.rope(dq, dk, d(ip), d(offsets), tab)
    ops.update_kv(ok_, dv, d(ip), c1, *tabs, kv_layout=layout)
    assert torch.equal(bits(oq.cpu()), bits(rq)) and torch.equal(bits(ok_.cpu()), bits(rk)), tag
    assert torch.equal(bits(c1.cpu()), want), tag
    c2 = d(to_layout(cache))
    two = r.random() < 0.5 and layout == "NHD"
    if two:
        c3 = d(cache)
        oq2 = ops.rope_append(dq, dk, dv, d(ip), d(offsets), tab, c2, *tabs, c3, *tabs, kv_layout=layout)
        assert torch.equal(bits(c3.cpu()), bits(ref_cache)), tag
    else:
        oq2 = ops.rope_append(dq, dk, dv, d(ip), d(offsets), tab, c2, *tabs, kv_layout=layout)
    assert torch.equal(bits(oq2.cpu()), bits(rq)) and torch.equal(bits(c2.cpu()), want), tag
    torch.cuda.synchronize()
    assert ops.page_overflow_count(reset=True) == 0, tag


# ----------------------------------------------------------------------------------------- the GEMM families
def _gemm_inputs(M, N, K, bias, seed):
    g = torch.Generator().manual_seed(seed)
    xfull = torch.randn(M, K + 64, generator=g).to(BF)             # row stride != K
    w = (torch.randn(N, K, generator=g) * 0.05).to(BF)
    b = torch.randn(N, generator=g).to(BF) if bias else None
    x = xfull[:, :K]
    ref = x.double() @ w.double().t() + (b.double() if bias else 0)
    mag = x.double().abs() @ w.double().abs().t() + (b.double().abs() if bias else 0)
    tol = 2.0 ** -8 * ref.abs() + 2 * K * 2.0 ** -24 * mag
    return xfull, w, b, ref, tol


def _draw_mnk(r, m_hi=256, n_mult=32, k_mult=128, n_hi=3072, k_hi=4096):
    M = r.choice([1, 2, 7, 31, 32, 33, 64, 65, 100, 128, 129, 200, 255, 256])
    M = min(M, m_hi)
    N = n_mult * r.randint(1, n_hi // n_mult)
    K = k_mult * r.randint(1, k_hi // k_mult)
    return M, N, K


@pytest.mark.parametrize("seed", cases(20))
def test_fuzz_gemm_families_vs_exact(ops, seed):
    """One random shape per seed through every family that supports it."""
    r = random.Random(5000 + seed)
    M, N, K = _draw_mnk(r)
    bias = r.random() < 0.5
    xfull, w, b, ref, tol = _gemm_inputs(M, N, K, bias, seed)
    ws = ops.AttnWorkspace(DEV)
    pw = ops.PackedWeight(d(w))
    x = d(xfull)[:, :K]
    ran = []

    def gate(name, y):
        assert y.shape == (M, N) and y.dtype == BF
        err = (y.cpu().double() - ref).abs()
        worst = float((err / tol).max())
        ran.append(f"{name} {worst:.3f}")
        assert bool((err <= tol).all()), f"seed {seed}: {name} M{M} N{N} K{K} bias={bias}: err/tol {worst:.3f}"

    if ops.linear_supported(M, N, K):
        gate("md_linear", ops.linear(x, pw, d(b), workspace=ws))
        gate("md_linear(row-major)", ops.linear(x, d(w), d(b), workspace=ws))
    if ops.fused_linear_supported(M, N, K):
        gate("md_linear_fused", ops.fused_linear(x, pw, d(b)))
    if ops.fused_split_supported(M, N, K):
        gate("md_linear_fused_split", ops.fused_split_linear(x, pw, d(b), workspace=ws))
    if N % 128 == 0 and ops.linear_block_supported(M, N, K):
        gate("md_linear_block", ops.linear_block(x, pw, d(b), workspace=ws))
    assert ran, f"seed {seed}: no family took M{M} N{N} K{K}"
    parity_report(f"[fuzz-gemm] seed {seed:2d} M={M:3d} N={N:5d} K={K:5d} bias={int(bias)}  max err/tol: " + "; ".join(ran))


@pytest.mark.parametrize("seed", cases(12))
def test_fuzz_residual_epilogues_bit_exact_vs_unfused_sequence(ops, seed):
    """resid + linear in the tile kernel's epilogue, and slices + bias + residual add + RMSNorm in the split combine: the same
    bits as the product followed by the stand-alone kernels (the reference's h = x + wo(...); rmsnorm(h) * w)."""
    r = random.Random(6000 + seed)
    M, N, K = _draw_mnk(r, n_hi=2048, k_hi=8192)
    bias = r.random() < 0.5
    xfull, w, b, _, _ = _gemm_inputs(M, N, K, bias, 50 + seed)
    g = torch.Generator().manual_seed(70 + seed)
    resid = torch.randn(M, N + 32, generator=g).to(BF)[:, :N]
    nw = (1 + 0.1 * torch.randn(N, generator=g)).to(BF)
    ws = ops.AttnWorkspace(DEV)
    pw = ops.PackedWeight(d(w))
    x = d(xfull)[:, :K]
    tag = f"seed {seed}: M{M} N{N} K{K} bias={bias}"
    if ops.fused_linear_supported(M, N, K, ops.FL_RESID):
        y = ops.fused_linear(x, pw, d(b))
        h = ops.fused_linear(x, pw, d(b), resid=d(resid).contiguous())
        assert torch.equal(bits(h.cpu()), bits((d(resid).contiguous() + y).cpu())), tag
    if ops.fused_split_supported(M, N, K):
        y = ops.fused_split_linear(x, pw, d(b), workspace=ws)
        rc = d(resid).contiguous()
        h_ref, n_ref = ops.add_rmsnorm(rc.clone(), y, d(nw), 1e-5)
        h, nrm = ops.fused_split_linear_add_rmsnorm(x, pw, rc.clone(), d(nw), 1e-5, d(b), ws)
        assert torch.equal(bits(h.cpu()), bits(h_ref.cpu())) and torch.equal(bits(nrm.cpu()), bits(n_ref.cpu())), tag


# ----------------------------------------------------------------------------------------- argmax
@pytest.mark.parametrize("seed", cases(10))
def test_fuzz_argmax_lowest_index_among_ties(ops, seed):
    r = random.Random(4000 + seed)
    rows = r.choice([1, 3, 64, 200, 256])
    vocab = r.choice([1000, 16032, 32003, 50257, 128256, 151936])
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(rows, vocab + 8, generator=g).to(BF)[:, :vocab]        # row stride != vocab
    top = float(torch.tensor(logits.float().max().item() + 1.0).to(BF))      # a bf16 value above every logit
    for row in range(rows):                                        # plant 1..4 equal maxima per row
        for _ in range(r.randint(1, 4)):
            logits[row, r.randrange(vocab)] = top
    want = torch.stack([torch.nonzero(logits[i].float() == top)[0, 0] for i in range(rows)])
    got = ops.argmax(d(logits)).cpu().view(-1)
    assert torch.equal(got, want), f"seed {seed}: rows {rows} vocab {vocab}"


# ----------------------------------------------------------------------------------------- SnapKV select
@pytest.mark.parametrize("seed", cases(10))
def test_fuzz_snapkv_select_bit_exact_on_exact_scores(ops, seed):
    """Random geometry (requests, kv heads, group size, head dim, context length off every tile boundary, window 16 / 32,
    budget, scattered source pages) on inputs whose q.k are small integers times a power of two -- exact in ANY summation
    order (the construction of test_gpu_ops.py::test_snapkv_select_every_score_magnitude_bit_exact) -- so pooled scores,
    selected indices (stable descending, lowest index among ties) and the gathered draft-cache rows equal the
    float64-linear oracle's bit for bit, up to the one thing left: the fp32 summation order of the softmax denominator
    (<= 2 pooled scores per request one ulp apart; the indices are then checked against the kernel's own scores)."""
    from oracle import magicdec_ref as mr
    r = random.Random(3000 + seed)
    W = r.choice([16, 32, 32])
    g = r.choice([4, 5, 8] if W == 32 else [2, 3, 4, 8])
    KH = r.choice([1, 2, 4])
    D = r.choice([64, 128])
    B = r.randint(1, 3)
    S = r.randint(W + 300, 2600)
    topk = min(r.choice([17, 97, 225, 480]), S - W)
    budget = topk + W
    H = KH * g
    amp, qs, ks = r.choice([(2, 1.0, 1.0), (4, 1.0, 1.0), (1, 2.0 ** -5, 2.0 ** -15), (2, 1.0, 1.0)])
    gen = torch.Generator().manual_seed(40 + seed)
    q = (torch.randint(-amp, amp + 1, (B * W, H, D), generator=gen).float() * qs).to(BF)
    k = torch.randint(-amp, amp + 1, (B, S, KH, D), generator=gen).float() * ks
    if r.random() < 0.4:                                           # every regime of the exponential inside one softmax row
        k[:, ::7] *= 16.0
        k[:, ::5] *= 2.0 ** -12
    k = k.to(BF)
    v = torch.randn(B, S, KH, D, generator=gen).to(BF)
    npg = (S + 127) // 128
    perm = torch.randperm(B * npg + 3, generator=gen)
    cache = torch.full((B * npg + 3, 2, 128, KH, D), float("nan")).to(BF)     # stale rows past S must never be candidates
    indices = []
    for b in range(B):
        kk = torch.full((npg * 128, KH, D), float("nan")).to(BF)
        vv = kk.clone()
        kk[:S], vv[:S] = k[b], v[b]
        pages = perm[b * npg:(b + 1) * npg]
        cache[pages, 0] = kk.view(npg, 128, KH, D)
        cache[pages, 1] = vv.view(npg, 128, KH, D)
        indices += pages.tolist()
    # the draft page table describes the cache AFTER the select (Engine/SnapKV/backend.py: budget rows per request)
    dppr = (budget + 127) // 128
    dcache = torch.zeros(B * dppr, 2, 128, KH, D, dtype=BF, device=DEV)
    ws = ops.AttnWorkspace(DEV)
    idx, sc = ops.snapkv_select(d(q), d(cache), torch.tensor(indices, dtype=torch.int32, device=DEV),
                                (torch.arange(B + 1, dtype=torch.int32) * npg).to(DEV), S, W, budget, 5, dcache,
                                torch.arange(B * dppr, dtype=torch.int32, device=DEV),
                                (torch.arange(B + 1, dtype=torch.int32) * dppr).to(DEV),
                                torch.full((B,), budget - (dppr - 1) * 128, dtype=torch.int32, device=DEV), ws,
                                return_scores=True)
    idx, sc, dk = idx.cpu().long(), sc.cpu(), dcache.cpu()
    tag = f"seed {seed}: B{B} KH{KH} g{g} D{D} S{S} W{W} budget{budget} amp{amp}"
    old = mr.LINEAR_MODE
    mr.LINEAR_MODE = "fp64"
    try:
        for b in range(B):
            want_idx, nk, nv, want_sc = mr.snapkv_select(q[b * W:(b + 1) * W], k[b], v[b], g, W, budget)
            neq = int((bits(sc[b]) != bits(want_sc)).sum())
            # the q.k scores are exact; what is left is the fp32 summation order of the softmax denominator over the
            # context (the kernel combines per-1024-column statistics, torch's CPU softmax has its own order -- as has the
            # reference's CUDA softmax): a probability on a bf16 rounding boundary may land on either side.  Seen once in
            # 1 150 wide-run cases: ONE of 5 100 pooled scores one ulp apart (seed 151), indices unaffected
            assert neq <= 2 and _ulp_close(sc[b], want_sc, ulps=1), tag + f": {neq} pooled scores of request {b} differ"
            rows = dk[b * dppr:(b + 1) * dppr]
            got_k, got_v = rows[:, 0].reshape(-1, KH, D)[:budget], rows[:, 1].reshape(-1, KH, D)[:budget]
            for h in range(KH):
                mine = idx[b, h]
                assert torch.equal(mine, torch.sort(sc[b, h].float(), descending=True, stable=True).indices[:topk]), tag
                assert torch.equal(bits(got_k[:topk, h]), bits(k[b][mine, h])), tag
                assert torch.equal(bits(got_v[:topk, h]), bits(v[b][mine, h])), tag
            if neq == 0:
                assert torch.equal(idx[b], want_idx), tag + f": indices of request {b}"
                assert torch.equal(bits(got_k), bits(nk)) and torch.equal(bits(got_v), bits(nv)), tag
            else:
                parity_report(f"[fuzz-snapkv] {tag}: request {b}: {neq} of {want_sc.numel()} pooled scores 1 ulp from the "
                              f"oracle's (softmax denominator summation order); indices equal: {torch.equal(idx[b], want_idx)}")
            assert torch.equal(bits(got_k[topk:]), bits(k[b][S - W:])) and torch.equal(bits(got_v[topk:]), bits(v[b][S - W:])), tag
    finally:
        mr.LINEAR_MODE = old


# ----------------------------------------------------------------------------------------- StreamingLLM eviction
@pytest.mark.parametrize("seed", cases(8))
def test_fuzz_streaming_shift_and_rotate_bit_exact(ops, seed):
    """KVCache.prefill of the StreamingLLM draft (sink 16 + window) chunk by chunk, for random budgets (1..5 pages, the last
    one partly filled), kv heads, head dims, batch sizes and prefix lengths: cache bytes and rotated-cache bytes after every
    chunk equal the oracle's (which is pinned to the real reference at budgets 129 and 513)."""
    from oracle import magicdec_ref as mr
    r = random.Random(2000 + seed)
    B, KH, D = r.randint(1, 3), r.choice([1, 2, 4]), r.choice([64, 128])
    budget = 128 * r.randint(1, 4) + r.randint(1, 127)
    ppr = budget // 128 + 1
    seq_len = r.randint(60, budget + 128 * r.randint(1, 5) + r.randint(0, 127))
    table = fr.rope_table(1024, D, 10000.0, 1.0)
    tab = ops.RopeTable(1024, D, 10000.0, 1.0, device=DEV)
    rope = lambda q, k, indptr, offsets: fr.apply_rope(q, k, indptr, offsets, table)
    g = torch.Generator().manual_seed(900 + seed)
    ref = torch.zeros(B * ppr, 2, 128, KH, D, dtype=BF)
    cache = torch.zeros(B * ppr, 2, 128, KH, D, dtype=BF, device=DEV)
    rot = torch.empty_like(cache)
    ctx, npr, evictions = 0, 0, 0
    for c0 in range(0, seq_len, 128):
        n = min(128, seq_len - c0)
        is_last = n != 128
        if ctx + n <= budget:                      # pre_encode (StreamingLLM/backend_draft.py:155-192): one more page
            npr, last = npr + 1, n
        else:
            npr, last = ppr, budget % 128
            evictions += 1
        k = torch.randn(B * n, KH, D, generator=g).to(BF)
        v = torch.randn(B * n, KH, D, generator=g).to(BF)
        t = dict(indices=torch.cat([torch.arange(b * ppr, b * ppr + npr, dtype=torch.int32) for b in range(B)]),
                 indptr=(torch.arange(B + 1) * npr).to(torch.int32), last=torch.full((B,), last, dtype=torch.int32))
        want_rot = mr.streaming_prefill_kv(ref, k, v, B, ctx, n, budget, t, rope, is_last)
        if ctx + n <= budget:
            ops.update_kv(d(k), d(v), d((torch.arange(B + 1) * n).to(torch.int32)), cache, d(t["indices"]), d(t["indptr"]),
                          d(t["last"]))
            valid = ctx + n
        else:
            ops.streaming_shift_append(d(k), d(v), cache, n, budget, 16, ppr)
            valid = budget
        overflow_last = (ctx + n > budget) and is_last
        dst = cache if overflow_last else rot
        if not overflow_last:
            rot.copy_(cache)
        ops.streaming_rotate(cache, dst, B, valid, ppr, tab)
        tag = f"seed {seed}: B{B} KH{KH} D{D} budget{budget} prefix{seq_len}, chunk at {c0} (n={n})"
        assert torch.equal(bits(cache.cpu()), bits(ref)), tag + ": cache"
        assert torch.equal(bits(dst.cpu()), bits(want_rot)), tag + ": rotated cache"
        ctx = min(ctx + n, budget)
    parity_report(f"[fuzz-stream] seed {seed} B{B} KH{KH} D{D} budget {budget} prefix {seq_len}: {evictions} evicting chunks, "
                  f"cache and rotated cache bit-equal to the oracle after every chunk")


# ----------------------------------------------------------------------------------------- accept / rollback
@pytest.mark.parametrize("variant", ["longspec", "selfspec_snapkv", "selfspec_stream"])
@pytest.mark.parametrize("seed", cases(8))
def test_fuzz_accept_rollback_vs_oracle(ops, seed, variant):
    """The fused accept / rollback kernel against the oracle's restatement of the three verify-loop bodies (pinned to the
    reference's own loop by tests/golden/accept_loop*.json) on random batches: gamma 1..7, 1..300 requests (several
    wavefronts), random match patterns, end-of-text tokens among drafts and bonus tokens, requests next to the length cap."""
    from oracle import magicdec_ref as mr
    r = random.Random(1000 + 31 * seed + len(variant))
    G = r.randint(1, 7)
    B = r.choice([1, 2, 5, 64, 65, 130, 300])
    dr, cap, dbl = {"longspec": (G, G, True), "selfspec_snapkv": (G + 1, G + 1, False),
                    "selfspec_stream": (G, G, True)}[variant]
    eot_1, eot_2 = 7, 11
    prefix = r.randint(20, 60)
    max_nodes = prefix + 80
    g = torch.Generator().manual_seed(77 + seed)
    p_match = r.choice([0.3, 0.8, 0.97, 1.0])
    tb = torch.randint(20, 1000, (B, G + 1), generator=g)
    tt = torch.randint(20, 1000, (B, G + 1), generator=g)
    m = torch.rand(B, G, generator=g) < p_match
    tt[:, :G] = torch.where(m, tb[:, 1:], tt[:, :G])
    if r.random() < 0.5:                                           # end-of-text tokens among the drafts / the targets
        tb[torch.rand(B, G + 1, generator=g) < 0.03] = eot_1
        tt[torch.rand(B, G + 1, generator=g) < 0.03] = eot_2
    grown = torch.randint(0, 70, (B,), generator=g)               # tokens generated so far
    if r.random() < 0.3:
        grown[r.randrange(B)] = 79 - r.randint(0, G)               # next to the cap: num_nodes reaches max_nodes
    num_nodes = prefix + grown
    cachelens = (num_nodes - 1 + G + 1).to(torch.int32)            # the verify pass has appended gamma + 1 rows
    lp = ((cachelens - 1) % 128 + 1).to(torch.int32)
    dcl = torch.randint(40, 300, (B,), generator=g).to(torch.int32)
    dlp = ((dcl - 1) % 128 + 1).to(torch.int32)
    out_cols = prefix + 80 + G + 2
    ref = dict(tb=tb.clone(), out=torch.zeros(B, out_cols, dtype=torch.long), nn=num_nodes.clone(), cl=cachelens.clone(),
               lp=lp.clone(), dcl=dcl.clone(), dlp=dlp.clone())
    res = mr.accept_step(ref["tb"], tt.clone(), ref["out"], ref["nn"], ref["cl"], ref["lp"], ref["dcl"], ref["dlp"], G, dr,
                         cap, eot_1, eot_2, max_nodes, dbl)
    dev = dict(tb=d(tb), out=torch.zeros(B, out_cols, dtype=torch.long, device=DEV), nn=d(num_nodes), cl=d(cachelens),
               lp=d(lp), dcl=d(dcl), dlp=d(dlp))
    an, bo = torch.zeros(B, dtype=torch.long, device=DEV), torch.zeros(B, dtype=torch.long, device=DEV)
    db, cu = torch.zeros(B, 2, dtype=torch.long, device=DEV), torch.zeros(B, dtype=torch.long, device=DEV)
    fl = torch.zeros(2, dtype=torch.int32, device=DEV)
    ops.accept_rollback(dev["tb"], d(tt), dev["out"], dev["nn"], dev["cl"], dev["lp"], dev["dcl"], dev["dlp"], G, dr, cap,
                        eot_1, eot_2, max_nodes, an, bo, db if dbl else None, cu if dbl else None, fl)
    tag = f"seed {seed} {variant}: B{B} gamma{G} p_match {p_match}"
    assert bool(fl[0]) == res["terminal"], tag
    assert torch.equal(an.cpu(), res["accept_nums"]) and torch.equal(bo.cpu(), res["bonus"]), tag
    for key in ("tb", "out", "nn", "cl", "lp", "dcl", "dlp"):
        assert torch.equal(dev[key].cpu(), ref[key]), tag + f": {key}"
    assert bool(fl[1]) == res["next_double"], tag
    if res["next_double"]:
        assert torch.equal(db.cpu(), res["double_buffer"]) and torch.equal(cu.cpu(), res["cachelens_update"]), tag


# ----------------------------------------------------------------------------------------- the tile kernel's fused launches
@pytest.mark.parametrize("seed", cases(12))
def test_fuzz_fused_qkv_rope_append_and_tile_shapes(ops, seed):
    """Random geometry through the checks of test_gpu_fused.py: wqkv + RoPE + paged append in one launch BIT-EXACT against the
    plain product followed by md_rope_append (bf16 / fp8 pages, NHD / HND, one or two caches, scattered tables, bias); the
    SwiGLU epilogue BIT-EXACT against product + md_silu_mul; 2 x 2 tiles BIT-IDENTICAL to 1 x 1 in every epilogue incl. the
    deferred-RMSNorm prologue."""
    from tests import test_gpu_fused as tf
    r = random.Random(11000 + seed)
    D = r.choice([64, 128])
    KH = r.choice([1, 2, 4, 8])
    H = KH * r.choice([1, 2, 4, 5, 8])
    n = r.choice([1, 1, 2, 4, 5])
    B = r.randint(1, min(64, 256 // n))
    K = 128 * r.randint(1, 24)
    lens = [n + r.randint(0, 400) for _ in range(B)]
    layout = r.choice(["NHD", "HND"])
    fp8 = r.random() < 0.3
    two = r.random() < 0.3 and not fp8
    if ops.fused_linear_supported(B * n, (H + 2 * KH) * D, K, ops.FL_ROPE_APPEND):
        tf.test_fused_qkv_rope_append_bit_exact(ops, f"fuzz{seed}", B, n, H, KH, D, K, lens, layout, fp8, two,
                                                r.random() < 0.6, r.random() < 0.5)
    M, inter, K2 = r.choice([1, 5, 32, 64, 100, 128, 200, 256]), 32 * r.randint(1, 48), 128 * r.randint(1, 16)
    if ops.fused_linear_supported(M, 2 * inter, K2, ops.FL_SWIGLU):
        g = torch.Generator().manual_seed(seed)
        x = d(torch.randn(M, K2, generator=g).to(BF))
        w13 = d((torch.randn(2 * inter, K2, generator=g) * 0.08).to(BF))
        y = ops.fused_linear(x, ops.PackedWeight(w13, swiglu=True), swiglu=True)
        h = ops.fused_linear(x, ops.PackedWeight(w13))
        assert torch.equal(bits(y), bits(ops.silu_mul(h[:, :inter], h[:, inter:]))), f"seed {seed}: swiglu M{M} I{inter} K{K2}"
    M, N, K3 = r.choice([1, 33, 64, 100, 128, 256]), 64 * r.randint(1, 40), 128 * r.randint(1, 24)
    if ops.fused_linear_supported(M, N, K3):
        tf.test_fused_2x2_tiles_bit_identical_to_1x1(ops, M, N, K3)
