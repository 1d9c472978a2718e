"""GPU parity of md_linear_fused (csrc/tilegemm.hip: a decode-step linear + the op that consumes it, one launch) --
run with -m gpu.

* the product itself: against the float64 matmul of the same bf16 operands, the gate of tests/test_gpu_gemm.py
  (|err| <= u |exact| + 2K 2^-24 sum|x||w|);
* every fused epilogue: BIT-EXACT against the unfused kernel sequence fed with this kernel's own plain output
  (MD_FL_NONE) -- a per-element dot product is accumulated identically whatever the epilogue, so the residual add, the
  SiLU * mul and RoPE + paged append must reproduce md_silu_mul / md_rope_append (themselves pinned to the oracle in
  tests/test_gpu_ops.py) to the last bit, cache bytes and dropped-row counter included.
"""
import pytest
import torch
import torch.nn.functional as F

from tests.conftest import parity_report
from tests.parity_util import bf16_ulp
from tests.test_gpu_ops import bits, make_paged

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16
F8 = torch.float8_e4m3fn


@pytest.fixture(scope="module")
def ops():
    from magicdec_amd import ops as _ops
    _ops._lib.load()
    return _ops


def d(t):
    return t.to(DEV) if t is not None else None


# (M, N, K): TP shards of the 1B / 8B linears (N = 768, K = 512 / 1792), short tails (K/8 = 64, 224 deep slices),
# ragged M (last M tile partly empty), one- and many-tile grids
SHAPES = [(1, 32, 128), (7, 96, 256), (32, 2048, 1024), (33, 160, 384), (64, 768, 2048), (64, 3072, 2048),
          (64, 2048, 512), (100, 1024, 512), (128, 768, 4096), (256, 768, 4096), (256, 4096, 512), (200, 4096, 1792),
          (64, 2048, 8192), (64, 4096, 1792), (96, 3584, 3584)]    # 16-wave path with a 7-step tail; 224-deep slices


@pytest.mark.parametrize("M,N,K", SHAPES, ids=[f"M{m}-N{n}-K{k}" for m, n, k in SHAPES])
@pytest.mark.parametrize("bias", [False, True], ids=["nobias", "bias"])
def test_fused_linear_vs_exact(ops, M, N, K, bias):
    g = torch.Generator().manual_seed(M * 131 + N * 7 + K)
    xfull = torch.randn(M, K + 64, generator=g).to(BF)
    x = xfull[:, :K]                                     # row stride != K
    w = (torch.randn(N, K, generator=g) * 0.05).to(BF)
    b = torch.randn(N, generator=g).to(BF) if bias else None
    ref = x.double() @ w.double().t() + (b.double() if bias else 0)
    mag = x.double().abs() @ w.double().abs().t() + (b.double().abs() if bias else 0)
    assert ops.fused_linear_supported(M, N, K)
    y = ops.fused_linear(d(xfull)[:, :K], ops.PackedWeight(d(w)), d(b))
    assert y.shape == (M, N) and y.dtype == BF
    err = (y.cpu().double() - ref).abs()
    tol = 2.0 ** -8 * ref.abs() + 2 * K * 2.0 ** -24 * mag
    ulp = bf16_ulp(ref)
    parity_report(f"[fused-gemm] M={M:3d} N={N:5d} K={K:5d} bias={int(bias)}  max err/tol {float((err / tol).max()):.3f}  "
                  f"max err {float((err / ulp).max()):.2f} ulp  != correctly rounded: "
                  f"{100 * float((y.cpu() != ref.to(BF)).double().mean()):.3f}%")
    assert bool((err <= tol).all())
    # deterministic: a second launch gives the same bits
    y2 = ops.fused_linear(d(xfull)[:, :K], ops.PackedWeight(d(w)), d(b))
    assert torch.equal(bits(y), bits(y2))


@pytest.mark.parametrize("M,N,K", [(4, 64, 256), (64, 2048, 512), (100, 1024, 1792), (256, 4096, 1024)])
def test_fused_residual_epilogue_bit_exact(ops, M, N, K):
    """out = bf16(resid + bf16(x W^T)): the reference's `h = x + attention(...)` (Engine/SnapKV/model.py:260-278)."""
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(BF)
    w = (torch.randn(N, K, generator=g) * 0.05).to(BF)
    r = torch.randn(M, N, generator=g).to(BF)
    pw = ops.PackedWeight(d(w))
    o = ops.fused_linear(d(x), pw)
    h = ops.fused_linear(d(x), pw, resid=d(r))
    assert torch.equal(bits(h.cpu()), bits(r + o.cpu()))            # torch's bf16 add = fp32 add, one rounding


@pytest.mark.parametrize("M,I,K", [(4, 64, 256), (64, 1024, 512), (100, 176, 384), (256, 2048, 1024), (64, 512, 2048)])
def test_fused_swiglu_epilogue_bit_exact(ops, M, I, K):
    """silu(x w1^T) * (x w3^T) == md_silu_mul on this kernel's own plain w1|w3 product, bit for bit."""
    g = torch.Generator().manual_seed(M + I + K)
    x = torch.randn(M, K, generator=g).to(BF)
    w13 = (torch.randn(2 * I, K, generator=g) * 0.08).to(BF)
    y = ops.fused_linear(d(x), ops.PackedWeight(d(w13), swiglu=True), swiglu=True)
    assert y.shape == (M, I)
    if (2 * I) % 32 == 0:
        h = ops.fused_linear(d(x), ops.PackedWeight(d(w13)))
        want = ops.silu_mul(h[:, :I], h[:, I:])
        assert torch.equal(bits(y), bits(want))
    # and against the correctly rounded sequence (the gate of test_gpu_gemm.py::test_linear_swiglu_epilogue)
    hx = (x.double() @ w13.double().t()).to(BF)
    ref = F.silu(hx[:, :I]) * hx[:, I:]
    eq = float((y.cpu() == ref).double().mean())
    parity_report(f"[fused-gemm] swiglu M={M} I={I} K={K}: bit-equal to the correctly rounded sequence {100 * eq:.3f}%")
    assert eq >= 0.998


QKV_CASES = [
    # name, B, rows_per_req, H, KH, D, K, lens (AFTER the append), layout, fp8, second cache, scatter, qkv bias
    ("draft-1b-shape", 4, 1, 8, 2, 64, 512, [300, 257, 129, 5], "NHD", False, False, False, False),
    ("verify-4rows-d128", 3, 4, 8, 2, 128, 1024, [260, 131, 4], "NHD", False, False, True, False),
    ("verify-4rows-d128-hnd", 3, 4, 8, 2, 128, 1024, [260, 131, 4], "HND", False, False, True, False),
    ("two-token-step", 5, 2, 4, 4, 64, 256, [130, 129, 128, 2, 77], "NHD", False, False, True, True),
    ("selfspec-verify-two-caches", 2, 4, 8, 2, 128, 512, [300, 200], "HND", False, True, False, False),
    ("fp8-pages-hnd", 3, 4, 10, 2, 128, 640, [260, 131, 9], "HND", True, False, True, True),
    ("fp8-pages-nhd", 2, 1, 4, 1, 64, 256, [129, 64], "NHD", True, False, False, False),
    ("tp8-shard-one-kv-head", 64, 4, 4, 1, 128, 4096, [16036 % 640 + 130] * 64, "HND", False, False, False, False),
    ("two-m-tiles-small-k", 16, 4, 4, 1, 128, 256, [166] * 16, "NHD", False, False, False, False),
    ("eight-m-tiles-k512", 64, 4, 4, 1, 128, 512, [166] * 64, "NHD", False, False, False, False),
    ("draft-1b-tp1-b64", 64, 1, 32, 8, 64, 2048, [258] * 64, "NHD", False, False, False, False),
]


@pytest.mark.parametrize("name,B,n,H,KH,D,K,lens,layout,fp8,two,scatter,bias", QKV_CASES, ids=[c[0] for c in QKV_CASES])
def test_fused_qkv_rope_append_bit_exact(ops, name, B, n, H, KH, D, K, lens, layout, fp8, two, scatter, bias):
    """wqkv + RoPE + paged append in one launch == this kernel's plain wqkv product followed by md_rope_append:
    rotated q, every byte of the cache(s), for bf16 / fp8 pages, NHD / HND, one or two caches, scattered pages."""
    g = torch.Generator().manual_seed(len(name) * 7 + B)
    M, N = B * n, (H + 2 * KH) * D
    x = torch.randn(M, K, generator=g).to(BF)
    w = (torch.randn(N, K, generator=g) * 0.05).to(BF)
    bvec = torch.randn(N, generator=g).to(BF) if bias else None
    cache, indices, indptr, last, _ = make_paged(B, lens, KH, D, seed=11, scatter=scatter)
    cache2, indices2, indptr2, last2, _ = make_paged(B, [l + 3 for l in lens], KH, D, seed=12)
    offsets = torch.tensor([l - n + 1000 * (b % 2) for b, l in enumerate(lens)], dtype=torch.int32)   # RoPE positions
    tab = ops.RopeTable(4096, D, 500000.0, 8.0, 1, 4, 8192, device=DEV)
    ip = torch.arange(B + 1, dtype=torch.int32) * n
    scales = None
    if layout == "HND":
        cache = cache.permute(0, 1, 3, 2, 4).contiguous()
    if fp8:
        scales = (d(torch.rand(KH, generator=g) * 0.02 + 0.01), d(torch.rand(KH, generator=g) * 0.02 + 0.01))
        cache = (cache.float() * 20).to(F8)
    pw = ops.PackedWeight(d(w))
    # unfused: plain product (same kernel) -> rope_append
    qkv = ops.fused_linear(d(x), pw, d(bvec))
    q = qkv[:, :H * D].unflatten(1, (H, D))
    k = qkv[:, H * D:(H + KH) * D].unflatten(1, (KH, D))
    v = qkv[:, (H + KH) * D:].unflatten(1, (KH, D))
    c1a, c2a = d(cache.clone()), d(cache2.clone()) if two else None
    ops.page_overflow_count(reset=True)
    want_q = ops.rope_append(q, k, v, d(ip), d(offsets), tab, c1a, d(indices), d(indptr), d(last), c2a,
                             d(indices2) if two else None, d(indptr2) if two else None, d(last2) if two else None,
                             n_max=n, kv_scales=scales, kv_layout=layout)
    # fused
    c1b, c2b = d(cache.clone()), d(cache2.clone()) if two else None
    got_q = ops.fused_qkv_rope_append(d(x), pw, d(bvec), H, KH, D, n, d(offsets), tab, c1b, d(indices), d(indptr), d(last),
                                      c2b, d(indices2) if two else None, d(indptr2) if two else None,
                                      d(last2) if two else None, kv_scales=scales, kv_layout=layout)
    torch.cuda.synchronize()
    raw = (lambda t: t.contiguous().view(torch.uint8)) if fp8 else bits
    if not torch.equal(bits(got_q), bits(want_q)):
        neq = torch.nonzero(bits(got_q.view(M, -1)) != bits(want_q.view(M, -1))).cpu()
        rows, cols = sorted(set(neq[:, 0].tolist())), sorted(set(neq[:, 1].tolist()))
        r0, c0 = int(neq[0, 0]), int(neq[0, 1])
        raise AssertionError(("q differs", len(neq), "rows", rows[:40], len(rows), "cols", cols[:40], len(cols), "first",
                              (r0, c0), float(got_q.view(M, -1)[r0, c0]), float(want_q.view(M, -1)[r0, c0]),
                              "plain", float(qkv[r0, c0])))
    assert torch.equal(raw(c1b), raw(c1a))
    assert not torch.equal(raw(c1b), raw(d(cache)))                  # something was appended
    if two:
        assert torch.equal(bits(c2b), bits(c2a))
        assert not torch.equal(bits(c2b), bits(d(cache2)))


def test_fused_qkv_append_beyond_mapped_pages_is_dropped_and_counted(ops):
    """An over-long last page (the reference's page tables do not grow in decode): the rows that fall beyond the
    request's mapped pages are dropped and counted once per row, exactly as md_rope_append does."""
    B, n, H, KH, D, K = 2, 4, 4, 2, 64, 256
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B * n, K, generator=g).to(BF)
    w = (torch.randn((H + 2 * KH) * D, K, generator=g) * 0.05).to(BF)
    cache, indices, indptr, last, _ = make_paged(B, [128, 100], KH, D, seed=5)
    last = torch.tensor([130, 100], dtype=torch.int32)                # request 0: 2 of its 4 rows beyond its one page
    offsets = torch.tensor([126, 96], dtype=torch.int32)
    tab = ops.RopeTable(1024, D, 10000.0, 1.0, None, None, None, device=DEV)
    ip = torch.arange(B + 1, dtype=torch.int32) * n
    pw = ops.PackedWeight(d(w))
    qkv = ops.fused_linear(d(x), pw)
    q = qkv[:, :H * D].unflatten(1, (H, D))
    k = qkv[:, H * D:(H + KH) * D].unflatten(1, (KH, D))
    v = qkv[:, (H + KH) * D:].unflatten(1, (KH, D))
    ca = d(cache.clone())
    ops.page_overflow_count(reset=True)
    ops.rope_append(q, k, v, d(ip), d(offsets), tab, ca, d(indices), d(indptr), d(last), n_max=n)
    n_unfused = ops.page_overflow_count(reset=True)
    cb = d(cache.clone())
    ops.fused_qkv_rope_append(d(x), pw, None, H, KH, D, n, d(offsets), tab, cb, d(indices), d(indptr), d(last))
    n_fused = ops.page_overflow_count(reset=True)
    assert n_unfused == n_fused == 2
    assert torch.equal(bits(cb), bits(ca))


def _close_bf16(got, want, what, min_equal=0.99):
    """Deferred-norm results vs the stand-alone norm: the only arithmetic difference is the ORDER of the fp32 sum of
    squares, i.e. rstd in its last bit -- a normalised element then lands on the other side of a bf16 rounding boundary
    with probability ~1e-5, and such a 1-ulp input change reaches an output element rarely: almost all outputs are
    bit-equal and none is further than 2 bf16 ulps away."""
    g, w = got.float().cpu(), want.float().cpu()
    eq = float((bits(got.cpu()) == bits(want.cpu())).double().mean())
    ulp = bf16_ulp(w.double()).float()
    worst = float(((g - w).abs() / ulp).max())
    parity_report(f"[fused-gemm] deferred norm, {what}: bit-equal to norm-then-linear {100 * eq:.3f}%, worst {worst:.2f} ulp")
    assert eq >= min_equal and worst <= 2.0, (what, eq, worst)


@pytest.mark.parametrize("M,dim,I", [(64, 2048, 1024), (64, 512, 256), (100, 1024, 512), (8, 256, 128)])
def test_deferred_rmsnorm_resid_then_swiglu(ops, M, dim, I):
    """wo + residual (writing the partial sums of squares) -> w1|w3 with the norm applied on the fly, against
    wo + residual -> md_rmsnorm -> w1|w3."""
    g = torch.Generator().manual_seed(M + dim + I)
    att = torch.randn(M, dim, generator=g).to(BF)
    wo = (torch.randn(dim, dim, generator=g) * 0.03).to(BF)
    x = torch.randn(M, dim, generator=g).to(BF)
    nw = (1 + 0.1 * torch.randn(dim, generator=g)).to(BF)
    w13 = (torch.randn(2 * I, dim, generator=g) * 0.05).to(BF)
    pwo, pw13 = ops.PackedWeight(d(wo)), ops.PackedWeight(d(w13), swiglu=True)
    h, ssq = ops.fused_linear(d(att), pwo, resid=d(x), want_ssq=True)
    h_plain = ops.fused_linear(d(att), pwo, resid=d(x))
    assert torch.equal(bits(h), bits(h_plain))                          # asking for ssq does not change h
    want_ssq = h.float().cpu().square().view(M, dim // 32, 32).sum(-1)
    assert ssq.shape == (M, dim // 32)
    assert torch.allclose(ssq.cpu(), want_ssq, rtol=2e-6, atol=0)
    y = ops.rmsnorm(h, d(nw), 1e-5)
    want = ops.fused_linear(y, pw13, swiglu=True)
    got = ops.fused_linear(h, pw13, swiglu=True, pro=ops.DeferredNorm(h, ssq, d(nw), 1e-5))
    _close_bf16(got, want, f"w1|w3 M={M} dim={dim} I={I}")


@pytest.mark.parametrize("B,n,H,KH,D,dim", [(64, 1, 32, 8, 64, 2048), (3, 4, 8, 2, 128, 1024), (16, 2, 4, 4, 64, 256)])
def test_deferred_rmsnorm_resid_then_qkv_rope_append(ops, B, n, H, KH, D, dim):
    """w2 + residual (partial sums) -> next layer's wqkv + RoPE + append with the norm on the fly, against the
    stand-alone norm in between: rotated q and the appended cache rows."""
    g = torch.Generator().manual_seed(B * 7 + n + dim)
    M, N = B * n, (H + 2 * KH) * D
    act = torch.randn(M, 512, generator=g).to(BF)
    w2 = (torch.randn(dim, 512, generator=g) * 0.05).to(BF)
    x = torch.randn(M, dim, generator=g).to(BF)
    nw = (1 + 0.1 * torch.randn(dim, generator=g)).to(BF)
    wqkv = (torch.randn(N, dim, generator=g) * 0.05).to(BF)
    lens = [130 + 17 * (b % 5) for b in range(B)]
    cache, indices, indptr, last, _ = make_paged(B, lens, KH, D, seed=21)
    offsets = torch.tensor([l - n for l in lens], dtype=torch.int32)
    tab = ops.RopeTable(2048, D, 500000.0, 8.0, 1, 4, 8192, device=DEV)
    pw2, pq = ops.PackedWeight(d(w2)), ops.PackedWeight(d(wqkv))
    h, ssq = ops.fused_linear(d(act), pw2, resid=d(x), want_ssq=True)
    y = ops.rmsnorm(h, d(nw), 1e-5)
    ca, cb = d(cache.clone()), d(cache.clone())
    want_q = ops.fused_qkv_rope_append(y, pq, None, H, KH, D, n, d(offsets), tab, ca, d(indices), d(indptr), d(last))
    got_q = ops.fused_qkv_rope_append(h, pq, None, H, KH, D, n, d(offsets), tab, cb, d(indices), d(indptr), d(last),
                                      pro=ops.DeferredNorm(h, ssq, d(nw), 1e-5))
    _close_bf16(got_q, want_q, f"q_rot B={B} n={n} D={D}")
    _close_bf16(cb, ca, f"cache B={B} n={n} D={D}")
    assert not torch.equal(bits(cb), bits(d(cache)))


@pytest.mark.parametrize("M,N,K", [(64, 2048, 2048), (33, 1024, 512), (100, 512, 1792), (256, 4096, 1024), (1, 64, 128),
                                   (64, 16384, 2048)])
def test_fused_2x2_tiles_bit_identical_to_1x1(ops, M, N, K):
    """Round 4: a workgroup may own 2 x 2 MFMA tiles (64 rows x 64 columns: half the per-CU ingest of the wide 64-row
    products).  The K slices, their summation order and the epilogues are those of the 1 x 1 form, so every output --
    plain, residual (+ the partial sums of squares), SwiGLU -- must be BIT-IDENTICAL between the two decompositions
    (md_debug_set_fused_nw 11 / 22 force them).  Round 6: the deferred-RMSNorm SwiGLU form has a 2 x 2 instantiation
    of its own (four-deep W ring, partial sums of squares requested in front of the first loads, a bare s_barrier): the
    row scales are added in the 1 x 1 form's order and both roundings of y = bf16(bf16(h * rstd) * w) are RNE, so
    "swiglu+norm" must be bit-identical across the knob too."""
    lib = ops._lib.load()
    g = torch.Generator().manual_seed(M * 7 + N + K)
    x = d(torch.randn(M, K, generator=g).to(BF))
    w = d((torch.randn(N, K, generator=g) * 0.05).to(BF))
    b = d(torch.randn(N, generator=g).to(BF))
    resid = d(torch.randn(M, N, generator=g).to(BF))
    nw = d((1 + 0.1 * torch.randn(K, generator=g)).to(BF))
    ssq_in = d(torch.rand(M, K // 32, generator=g) * 32.0)
    pw, pw13 = ops.PackedWeight(w), ops.PackedWeight(w, swiglu=True)
    outs = {}
    try:
        for knob in (11, 22):
            lib.md_debug_set_fused_nw(knob)
            plain = ops.fused_linear(x, pw, b)
            h, ssq = ops.fused_linear(x, pw, b, resid=resid, want_ssq=True)
            sw = ops.fused_linear(x, pw13, swiglu=True)
            swn = ops.fused_linear(x, pw13, swiglu=True, pro=ops.DeferredNorm(x, ssq_in, nw, 1e-5))
            outs[knob] = (plain, h, ssq, sw, swn)
    finally:
        lib.md_debug_set_fused_nw(0)
    for a, c, what in zip(outs[11], outs[22], ("plain", "resid", "ssq", "swiglu", "swiglu+norm")):
        assert torch.equal(a.view(torch.int32 if a.dtype == torch.float32 else torch.int16),
                           c.view(torch.int32 if c.dtype == torch.float32 else torch.int16)), what
    assert not torch.isnan(outs[22][4].float()).any()
