"""GPU parity of md_linear (the weight-streaming skinny GEMM of the decode / verify linears) -- run with -m gpu.

Reference = float64 matmul of the same bf16 operands, rounded once (the correctly rounded result).  Gate per element:
    |hip - exact| <= u * |exact| + gamma_K * sum_k |x_k w_k|,   u = 2^-8 (one bf16 rounding), gamma_K = 2 * K * 2^-24
(the fp32 accumulation bound; the achieved accumulation error is ~1e-7 relative and is reported).  The oracle's own
torch-CPU F.linear is measured the same way and reported next to it.  Fused epilogues (SwiGLU, int8 scales, bias)
follow the reference's rounding points and are gated in bf16 ulps.
"""
import pytest
import torch
import torch.nn.functional as F

from tests.conftest import parity_report
from tests.parity_util import bf16_ulp

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    from magicdec_amd import ops as _ops
    _ops._lib.load()
    return _ops


def _neighbours(t):
    """bf16 tensor -> (t - 1 ulp, t, t + 1 ulp)."""
    i = t.contiguous().view(torch.int16).int()
    mag = i & 0x7fff
    up = torch.where(i >= 0, i + 1, torch.where(mag == 0, torch.ones_like(i), i - 1))      # towards +inf
    dn = torch.where(i > 0, i - 1, torch.where(mag == 0, torch.full_like(i, -32767), i + 1))
    f = lambda v: v.to(torch.int16).view(torch.bfloat16)
    return f(dn), t, f(up)


def _matches_some_neighbour(y, fn, h1, h3=None, silu_ulp=False):
    """y equals fn evaluated on the correctly rounded GEMM output(s) or on a 1-ulp neighbour of them: what a different
    fp32 summation order can legitimately produce at a rounding boundary (elementwise)."""
    ok = torch.zeros(y.shape, dtype=torch.bool)
    for a in _neighbours(h1):
        for b in (_neighbours(h3) if h3 is not None else (None,)):
            cand = fn(a, b)
            if silu_ulp:
                # + SiLU's own <= 1 ulp (expf implementation), which the bf16 multiply by h3 can turn into 2 ulps of
                # the product: accept a bit distance <= 2 from the candidate
                ci, yi = cand.contiguous().view(torch.int16).int(), y.contiguous().view(torch.int16).int()
                ok |= ((ci - yi).abs() <= 2) & ((ci ^ yi) >= 0)
                ok |= (cand.float().abs() < 1e-30) & (y.float().abs() < 1e-30)
            else:
                ok |= (cand == y)
    return ok


def _exact(x, w, b=None):
    y = x.double() @ w.double().t()
    return y + b.double() if b is not None else y


SHAPES = [(1, 128, 128), (7, 96, 256), (32, 2048, 1024), (33, 132, 384), (64, 3072, 2048), (64, 100, 3456),
          (100, 1024, 512), (128, 6144, 1024), (200, 516, 640), (256, 4096, 1024), (256, 36, 128)]


@pytest.mark.parametrize("M,N,K", SHAPES, ids=[f"M{m}-N{n}-K{k}" for m, n, k in SHAPES])
@pytest.mark.parametrize("bias", [False, True], ids=["nobias", "bias"])
@pytest.mark.parametrize("packed", [False, True], ids=["rowmajor", "packed"])
def test_linear_vs_exact(ops, M, N, K, bias, packed):
    g = torch.Generator().manual_seed(M * 131 + N * 7 + K)
    xfull = torch.randn(M, K + 64, generator=g).to(BF)
    x = xfull[:, :K]                                     # row stride != K: a slice of a wider activation tensor
    w = (torch.randn(N, K, generator=g) * 0.05).to(BF)
    b = torch.randn(N, generator=g).to(BF) if bias else None
    ref = _exact(x, w, b)
    mag = x.double().abs() @ w.double().abs().t() + (b.double().abs() if bias else 0)
    ws = ops.AttnWorkspace(DEV)
    assert ops.linear_supported(M, N, K)
    wd = ops.PackedWeight(w.to(DEV)) if packed else w.to(DEV)
    y = ops.linear(xfull.to(DEV)[:, :K], wd, b.to(DEV) if bias else None, workspace=ws)
    assert y.shape == (M, N) and y.dtype == BF
    err = (y.cpu().double() - ref).abs()
    tol = 2.0 ** -8 * ref.abs() + 2 * K * 2.0 ** -24 * mag
    ora = (F.linear(x, w, b).double() - ref).abs()
    ulp = bf16_ulp(ref)
    parity_report(f"[gemm] M={M:3d} N={N:5d} K={K:5d} bias={int(bias)} packed={int(packed)}  max err/tol {float((err / tol).max()):.3f}  "
                  f"max err {float((err / ulp).max()):.2f} ulp  oracle(torch CPU) {float((ora / ulp).max()):.2f} ulp  "
                  f"!= correctly rounded: hip {100 * float((y.cpu() != ref.to(BF)).double().mean()):.3f}% "
                  f"oracle {100 * float((F.linear(x, w, b) != ref.to(BF)).double().mean()):.3f}%")
    assert bool((err <= tol).all())


@pytest.mark.parametrize("packed", [False, True], ids=["rowmajor", "packed"])
@pytest.mark.parametrize("M,I,K", [(4, 64, 256), (64, 1024, 512), (100, 176, 384), (256, 2048, 1024)])
def test_linear_swiglu_epilogue(ops, M, I, K, packed):
    """silu(x.w1^T) * (x.w3^T) with the reference's rounding points (h1, h3 -> bf16; silu -> bf16; product -> bf16,
    Engine/SnapKV/model.py:451-455): >= 99.9 % bit-equal to that sequence evaluated on the correctly rounded h1 / h3,
    and every other element equals it evaluated on a 1-ulp neighbour of h1 / h3 (an fp32 sum on a rounding boundary),
    up to the 1 ulp the SiLU itself is gated at (expf implementation; tests/test_gpu_ops.py::test_silu_mul) -- or, where
    h1 / h3 nearly cancel to zero, lies within the propagated fp32 accumulation bound."""
    g = torch.Generator().manual_seed(M + I + K)
    x = torch.randn(M, K, generator=g).to(BF)
    w13 = (torch.randn(2 * I, K, generator=g) * 0.08).to(BF)
    h = _exact(x, w13).to(BF)
    ref = F.silu(h[:, :I]) * h[:, I:]
    ws = ops.AttnWorkspace(DEV)
    wd = ops.PackedWeight(w13.to(DEV), swiglu=True) if packed else w13.to(DEV)
    y = ops.linear(x.to(DEV), wd, swiglu=True, workspace=ws).cpu()
    assert y.shape == (M, I)
    eq = float((y == ref).double().mean())
    ok = _matches_some_neighbour(y, lambda a, b: F.silu(a) * b, h[:, :I], h[:, I:], silu_ulp=True)
    parity_report(f"[gemm] swiglu M={M} I={I} K={K} packed={int(packed)}: bit-equal to the correctly rounded sequence "
                  f"{100 * eq:.3f}%; the rest explained by a 1-ulp neighbour of h1/h3: {bool(ok.all())}")
    # elements whose h1 or h3 nearly cancels (|h| ~ 1e-6 against terms of size 1): there a bf16 "1-ulp neighbour" is
    # far smaller than the legitimate fp32 accumulation error, so they are gated with the propagated absolute bound
    # |dy| <= |h3| * 1.1 * tol(h1) + |silu(h1)| * tol(h3) + 2^-7 |ref|,  tol(h) = 2K 2^-24 sum|x||w| + 2^-8 |h|
    mag = x.double().abs() @ w13.double().abs().t()
    hd = _exact(x, w13)
    tolh = 2 * K * 2.0 ** -24 * mag + 2.0 ** -8 * hd.abs()
    bound = (hd[:, I:].abs() * 1.1 * tolh[:, :I] + F.silu(hd[:, :I]).abs() * tolh[:, I:] + 2.0 ** -7 * ref.double().abs())
    ok |= (y.double() - ref.double()).abs() <= bound
    if not bool(ok.all()):
        bad = torch.nonzero(~ok)[:8].tolist()
        hh = ops.linear(x.to(DEV), w13.to(DEV), workspace=ws).cpu()          # the same kernel's plain GEMM output
        detail = [(r, c, "h1 exact/hip", h[r, c].item(), hh[r, c].item(), "h3 exact/hip", h[r, I + c].item(),
                   hh[r, I + c].item(), "y", y[r, c].item(), "ref", ref[r, c].item()) for r, c in bad]
        raise AssertionError((int((~ok).sum()), detail))
    assert eq >= 0.999


@pytest.mark.parametrize("M,N,K,swiglu", [(8, 256, 256, False), (64, 1024, 1024, False), (200, 128, 384, False),
                                          (64, 512, 512, True)])
def test_linear_int8_weight_only(ops, M, N, K, swiglu):
    """WeightOnlyInt8Linear.forward (Engine/quantize.py:84-86): F.linear(x, w_int8.to(bf16)) * scales -- the GEMM output
    is rounded to bf16, then multiplied by the bf16 per-channel scale in bf16.  Weights convert exactly; gate: >= 99.9 %
    bit-equal to that sequence on the correctly rounded GEMM output, the rest explained by its 1-ulp neighbours."""
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(BF)
    wq = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8)
    sc = (torch.rand(N, generator=g) * 0.01 + 0.001).to(BF)
    g0 = _exact(x, wq.to(BF)).to(BF)
    h = g0 * sc
    I = N // 2
    ref = F.silu(h[:, :I]) * h[:, I:] if swiglu else h
    ws = ops.AttnWorkspace(DEV)
    wd = ops.PackedWeight(wq.to(DEV), swiglu=swiglu)
    for tag, wdev in (("rowmajor", wq.to(DEV)), ("packed", wd)):
        y = ops.linear(x.to(DEV), wdev, scales=sc.to(DEV), swiglu=swiglu, workspace=ws).cpu()
        eq = float((y == ref).double().mean())
        if swiglu:
            ok = _matches_some_neighbour(y, lambda a, b: F.silu(a * sc[:I]) * (b * sc[I:]), g0[:, :I], g0[:, I:],
                                         silu_ulp=True)
        else:
            ok = _matches_some_neighbour(y, lambda a, b: a * sc, g0)
        parity_report(f"[gemm] int8 M={M} N={N} K={K} swiglu={int(swiglu)} {tag}: bit-equal {100 * eq:.3f}%; rest "
                      f"explained by 1-ulp neighbours of the GEMM output: {bool(ok.all())}")
        assert eq >= 0.999 and bool(ok.all())


def test_linear_full_size_and_determinism(ops):
    """Headline shapes (8B wqkv at the verify M = 256 and the 1B w2 at M = 64) against a float64 GEMM on the GPU;
    two runs are bit-identical (fixed-order split-K combine)."""
    for M, N, K in ((256, 6144, 4096), (64, 2048, 8192)):
        g = torch.Generator(device=DEV).manual_seed(1)
        x = torch.randn(M, K, device=DEV, generator=g, dtype=torch.float32).to(BF)
        w = (torch.randn(N, K, device=DEV, generator=g, dtype=torch.float32) * 0.02).to(BF)
        ws = ops.AttnWorkspace(DEV)
        y1 = ops.linear(x, w, workspace=ws)
        y2 = ops.linear(x, w, workspace=ws)
        y3 = ops.linear(x, ops.PackedWeight(w), workspace=ws)
        assert torch.equal(y1, y2) and torch.equal(y1, y3), "row-major and streaming layouts must give the same bits"
        ref = x.double() @ w.double().t()
        mag = x.double().abs() @ w.double().abs().t()
        err = (y1.double() - ref).abs()
        tol = 2.0 ** -8 * ref.abs() + 2 * K * 2.0 ** -24 * mag
        lib = (F.linear(x, w).double() - ref).abs()
        parity_report(f"[gemm] full size M={M} N={N} K={K}: max err/tol {float((err / tol).max()):.3f}; != correctly "
                      f"rounded: hip {100 * float((y1 != ref.to(BF)).double().mean()):.4f}%  hipBLASLt "
                      f"{100 * float((F.linear(x, w) != ref.to(BF)).double().mean()):.4f}%  (max err hipBLASLt/tol "
                      f"{float((lib / tol).max()):.3f})")
        assert bool((err <= tol).all())


def test_linear_rejects_bad_arguments(ops):
    x = torch.zeros(4, 100, dtype=BF, device=DEV)
    w = torch.zeros(64, 100, dtype=BF, device=DEV)
    assert not ops.linear_supported(4, 64, 100) and not ops.linear_supported(300, 64, 128)
    with pytest.raises(ops.MagicDecHipError):
        ops.linear(x, w)
    with pytest.raises(ValueError):
        ops.linear(torch.zeros(4, 128, dtype=BF, device=DEV), torch.zeros(64, 128, dtype=torch.int8, device=DEV))


@pytest.mark.parametrize("M,N,K,int8", [(64, 2048, 8192, False), (256, 4096, 14336, False), (32, 1024, 2048, False),
                                        (100, 512, 1024, False), (64, 1024, 2048, True)])
@pytest.mark.parametrize("packed", [False, True], ids=["rowmajor", "packed"])
def test_linear_add_rmsnorm_equals_linear_then_add_rmsnorm(ops, M, N, K, int8, packed):
    """md_linear_add_rmsnorm (the split-K combine launch also adds the residual and normalises) must reproduce
    md_linear -> md_add_rmsnorm bit for bit: h and y."""
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(BF)
    r = torch.randn(M, N, generator=g).to(BF)
    nw = (1 + 0.1 * torch.randn(N, generator=g)).to(BF)
    ws = ops.AttnWorkspace(DEV)
    if int8:
        w = torch.randint(-127, 128, (N, K), generator=g, dtype=torch.int8)
        sc = (torch.rand(N, generator=g) * 0.01 + 0.001).to(BF)
        wd, scales = w.to(DEV), sc.to(DEV)
        if packed:
            wd = ops.PackedWeight(wd)
    else:
        w = (torch.randn(N, K, generator=g) * 0.05).to(BF)
        wd, scales = (ops.PackedWeight(w.to(DEV)) if packed else w.to(DEV)), None
    if not ops.linear_add_rmsnorm_supported(M, N, K):
        pytest.skip("K is not split for this shape: nothing to fuse into")
    o = ops.linear(x.to(DEV), wd, scales=scales, workspace=ws)
    h_want, y_want = ops.add_rmsnorm(r.to(DEV), o, nw.to(DEV), 1e-5)
    h, y = ops.linear_add_rmsnorm(x.to(DEV), wd, r.to(DEV), nw.to(DEV), 1e-5, scales=scales, workspace=ws)
    assert torch.equal(h.view(torch.int16), h_want.view(torch.int16))
    assert torch.equal(y.view(torch.int16), y_want.view(torch.int16))


@pytest.mark.parametrize("M,N,K,swiglu,packed", [
    (32, 28672, 4096, True, True),       # cfg2: the 8B w1|w3 of a draft pass (K split over workgroups)
    (64, 16384, 2048, True, True),       # the 1B w1|w3 of a draft step
    (1, 512, 256, False, False), (33, 1024, 1792, True, True), (100, 2048, 2048, False, True),
    (128, 4096, 1024, True, False), (256, 1024, 4096, False, True), (64, 128256, 2048, False, True),
])
def test_linear_normed_equals_norm_then_linear(ops, M, N, K, swiglu, packed):
    """Round 4: md_linear_normed -- the deferred RMSNorm on the weight-streaming kernel.  x is the un-normalised h with the
    partial sums of squares the residual epilogue of md_linear_fused wrote; against md_rmsnorm followed by md_linear on
    the same kernel the only arithmetic difference is the ORDER of the fp32 sum of squares (rstd in its last bit), so
    nearly every output is bit-equal and none is further than 2 bf16 ulps away; the producer's real ssq is used (a
    residual-epilogue launch), rows >= M and the ragged last slab included."""
    g = torch.Generator().manual_seed(M * 3 + N + K)
    prod_w = ops.PackedWeight((torch.randn(K, 256, generator=g) * 0.05).to(BF).to(DEV))
    act = (torch.randn(M, 256, generator=g)).to(BF).to(DEV)
    resid = torch.randn(M, K, generator=g).to(BF).to(DEV)
    h, ssq = ops.fused_linear(act, prod_w, resid=resid, want_ssq=True)          # h [M, K], ssq [M, K / 32]
    nw = (1 + 0.1 * torch.randn(K, generator=g)).to(BF).to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.05).to(BF).to(DEV)
    wt = ops.PackedWeight(w, swiglu=swiglu) if packed else w
    ws = ops.AttnWorkspace(DEV)
    want = ops.linear(ops.rmsnorm(h, nw, 1e-5), wt, swiglu=swiglu, workspace=ws)
    got = ops.linear(h, wt, swiglu=swiglu, workspace=ws, pro=ops.DeferredNorm(h, ssq, nw, 1e-5))
    assert got.shape == want.shape and not torch.isnan(got.float()).any()
    gi, wi = got.cpu().view(torch.int16), want.cpu().view(torch.int16)
    eq = float((gi == wi).double().mean())
    ulp = bf16_ulp(want.cpu().double()).float()
    worst = float(((got.cpu().float() - want.cpu().float()).abs() / ulp).max())
    parity_report(f"[gemm] md_linear_normed M={M} N={N} K={K} swiglu={swiglu} packed={packed}: bit-equal to "
                  f"norm-then-linear {100 * eq:.3f}%, worst {worst:.2f} ulp")
    assert eq >= 0.99 and worst <= 2.0
    # same launch again: the same bits
    again = ops.linear(h, wt, swiglu=swiglu, workspace=ws, pro=ops.DeferredNorm(h, ssq, nw, 1e-5))
    assert torch.equal(again.view(torch.int16), got.view(torch.int16))


def test_linear_normed_rejects_a_mismatched_norm(ops):
    h = torch.randn(8, 256, device=DEV).to(BF)
    w = torch.randn(64, 256, device=DEV).to(BF)
    nw = torch.ones(256, device=DEV, dtype=BF)
    ws = ops.AttnWorkspace(DEV)
    with pytest.raises(ValueError):      # ssq must cover K / 32 tiles
        ops.linear(h, w, workspace=ws, pro=ops.DeferredNorm(h, torch.zeros(8, 4, device=DEV), nw, 1e-5))
    with pytest.raises(ValueError):      # x must be the h the sums belong to
        ops.linear(h.clone(), w, workspace=ws, pro=ops.DeferredNorm(h, torch.zeros(8, 8, device=DEV), nw, 1e-5))
    with pytest.raises(TypeError):       # bf16 weights only
        ops.linear(h, torch.zeros(64, 256, dtype=torch.int8, device=DEV), scales=torch.ones(64, device=DEV, dtype=BF),
                   workspace=ws, pro=ops.DeferredNorm(h, torch.zeros(8, 8, device=DEV), nw, 1e-5))


@pytest.mark.parametrize("M,N,K,swiglu,packed,normed", [
    (32, 28672, 4096, True, True, False),     # the 8B w1|w3 of a cfg2 draft pass: the balance rule picks 7 waves
    (32, 28672, 4096, True, True, True),      # ... with the deferred RMSNorm on its activation path
    (64, 6144, 4096, False, True, False),     # 8B wqkv (192 wave tiles, 8 K slices): the rule picks 6 waves
    (33, 1024, 1792, True, False, False), (100, 2048, 2048, False, True, True), (128, 3584, 4096, True, True, False),
    (17, 4096, 512, False, False, False),
])
def test_linear_waves_per_workgroup_do_not_change_the_bits(ops, M, N, K, swiglu, packed, normed):
    """Round 5: md_linear with 4, 6 or 7 wavefronts per workgroup (md_debug_set_gemm_waves; 0 = the balance rule of
    csrc/gemm.hip:plan_of, which takes 6 / 7 waves where four leave some CUs with twice the bytes of others).  A wave's
    tile, K range and slab order do not depend on its neighbours in the workgroup, and the K split is the same: every
    form must give the SAME bits, for plain / SwiGLU / deferred-norm products, ragged row counts and partial last
    workgroups (ntiles % 6, % 7 != 0)."""
    import ctypes
    lib = ops._lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(BF).to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.05).to(BF).to(DEV)
    wt = ops.PackedWeight(w, swiglu=swiglu) if packed else w
    ws = ops.AttnWorkspace(DEV)
    pro = None
    if normed:
        ssq = (x.float().view(M, K // 32, 32) ** 2).sum(-1).contiguous()
        pro = ops.DeferredNorm(x, ssq, (1 + 0.1 * torch.randn(K, generator=g)).to(BF).to(DEV), 1e-5)
    outs = {}
    try:
        for nw in (4, 6, 7, 0):
            lib.md_debug_set_gemm_waves(ctypes.c_int(nw))
            outs[nw] = ops.linear(x, wt, swiglu=swiglu, workspace=ws, pro=pro).clone()
    finally:
        lib.md_debug_set_gemm_waves(ctypes.c_int(0))
    assert not torch.isnan(outs[4].float()).any()
    for nw in (6, 7, 0):
        assert torch.equal(outs[nw].view(torch.int16), outs[4].view(torch.int16)), f"{nw} waves per workgroup != 4"
