"""GPU parity of md_linear_block (csrc/blockgemm.hip: the block-tile GEMM of the 129..256-row verify linears) -- run
with -m gpu.  Same gates as tests/test_gpu_gemm.py: the product against the float64 matmul of the same bf16 operands
(|err| <= u |exact| + 2 K 2^-24 sum|x||w|), the SwiGLU epilogue against the reference's rounding sequence on the
correctly rounded h1 / h3 (or a 1-ulp neighbour), and the residual + RMSNorm combine bit-exact against the unfused
sequence fed with this kernel's own plain output."""
import pytest
import torch
import torch.nn.functional as F

from tests.conftest import parity_report
from tests.parity_util import bf16_ulp
from tests.test_gpu_gemm import _exact, _matches_some_neighbour

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    from magicdec_amd import ops as _ops
    _ops._lib.load()
    return _ops


# (M, N, K): un-split (>= 160 column tiles), split (few tiles), ragged M, one / two / many stages per slice, TP shards
SHAPES = [(256, 128, 64), (256, 128, 128), (256, 256, 192), (256, 1024, 4096), (200, 512, 1024), (129, 384, 832),
          (256, 20480, 256), (256, 768, 4096), (256, 4096, 512), (64, 256, 448), (1, 128, 64), (256, 6144, 1024)]


@pytest.mark.parametrize("M,N,K", SHAPES, ids=[f"M{m}-N{n}-K{k}" for m, n, k in SHAPES])
@pytest.mark.parametrize("bias", [False, True], ids=["nobias", "bias"])
def test_block_linear_vs_exact(ops, M, N, K, bias):
    g = torch.Generator().manual_seed(M * 131 + N * 7 + K)
    xfull = torch.randn(M, K + 64, generator=g).to(BF)
    x = xfull[:, :K]                                     # row stride != K
    w = (torch.randn(N, K, generator=g) * 0.05).to(BF)
    b = torch.randn(N, generator=g).to(BF) if bias else None
    ref = _exact(x, w, b)
    mag = x.double().abs() @ w.double().abs().t() + (b.double().abs() if bias else 0)
    ws = ops.AttnWorkspace(DEV)
    assert ops.linear_block_supported(M, N, K)
    y = ops.linear_block(xfull.to(DEV)[:, :K], ops.PackedWeight(w.to(DEV)), b.to(DEV) if bias else None, workspace=ws)
    assert y.shape == (M, N) and y.dtype == BF
    err = (y.cpu().double() - ref).abs()
    tol = 2.0 ** -8 * ref.abs() + 2 * K * 2.0 ** -24 * mag
    ulp = bf16_ulp(ref)
    parity_report(f"[blockgemm] M={M:3d} N={N:5d} K={K:5d} bias={int(bias)}  max err/tol {float((err / tol).max()):.3f}  "
                  f"max err {float((err / ulp).max()):.2f} ulp  != correctly rounded: "
                  f"{100 * float((y.cpu() != ref.to(BF)).double().mean()):.3f}%")
    assert bool((err <= tol).all())


@pytest.mark.parametrize("M,I,K", [(256, 64, 64), (256, 1024, 512), (130, 192, 384), (256, 10240, 256), (256, 1792, 4096)])
def test_block_linear_swiglu_epilogue(ops, M, I, K):
    """silu(x.w1^T) * (x.w3^T) with the reference's rounding points (Engine/SnapKV/model.py:451-455), in the kernel
    (un-split) and in the combine launch (split K)."""
    g = torch.Generator().manual_seed(M + I + K)
    x = torch.randn(M, K, generator=g).to(BF)
    w13 = (torch.randn(2 * I, K, generator=g) * 0.08).to(BF)
    h = _exact(x, w13).to(BF)
    ref = F.silu(h[:, :I]) * h[:, I:]
    ws = ops.AttnWorkspace(DEV)
    y = ops.linear_block(x.to(DEV), ops.PackedWeight(w13.to(DEV), swiglu=True), swiglu=True, workspace=ws).cpu()
    assert y.shape == (M, I)
    eq = float((y == ref).double().mean())
    ok = _matches_some_neighbour(y, lambda a, b: F.silu(a) * b, h[:, :I], h[:, I:], silu_ulp=True)
    mag = x.double().abs() @ w13.double().abs().t()
    hd = _exact(x, w13)
    tolh = 2 * K * 2.0 ** -24 * mag + 2.0 ** -8 * hd.abs()
    bound = (hd[:, I:].abs() * 1.1 * tolh[:, :I] + F.silu(hd[:, :I]).abs() * tolh[:, I:] + 2.0 ** -7 * ref.double().abs())
    ok |= (y.double() - ref.double()).abs() <= bound
    parity_report(f"[blockgemm] swiglu M={M} I={I} K={K}: bit-equal to the correctly rounded sequence {100 * eq:.3f}%; "
                  f"the rest explained by a 1-ulp neighbour of h1/h3: {bool(ok.all())}")
    assert bool(ok.all()) and eq >= 0.999


@pytest.mark.parametrize("M,N,K", [(256, 4096, 1792), (256, 512, 256), (192, 1024, 4096), (256, 2048, 64)])
@pytest.mark.parametrize("bias", [False, True], ids=["nobias", "bias"])
def test_block_linear_add_rmsnorm_matches_the_unfused_sequence(ops, M, N, K, bias):
    """(h, y) of md_linear_block_add_rmsnorm == md_add_rmsnorm(resid, md_linear_block(x)) bit for bit: the combine
    launch is md_linear's (sum in slice order, + bias, one rounding, bf16 add, RMSNorm with the reference's rounding
    points, Engine/SnapKV/model.py:260-278,464-469)."""
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(BF).to(DEV)
    w = ops.PackedWeight((torch.randn(N, K, generator=g) * 0.05).to(BF).to(DEV))
    b = torch.randn(N, generator=g).to(BF).to(DEV) if bias else None
    resid = torch.randn(M, N, generator=g).to(BF).to(DEV)
    nw = (1 + 0.1 * torch.randn(N, generator=g)).to(BF).to(DEV)
    ws = ops.AttnWorkspace(DEV)
    h, y = ops.linear_block_add_rmsnorm(x, w, resid, nw, 1e-5, b, ws)
    o = ops.linear_block(x, w, b, workspace=ws)
    h2, y2 = ops.add_rmsnorm(resid, o, nw, 1e-5)
    parity_report(f"[blockgemm] add_rmsnorm M={M} N={N} K={K} bias={int(bias)}: h equal {bool(torch.equal(h, h2))}, "
                  f"y equal {bool(torch.equal(y, y2))}")
    assert torch.equal(h, h2) and torch.equal(y, y2)


def test_block_linear_is_deterministic_and_ignores_stale_lds(ops):
    """Two calls give the same bits; rows >= M never reach the output (they are zero-filled through the descriptor
    bound, and a NaN-filled neighbour allocation must not leak in)."""
    g = torch.Generator().manual_seed(5)
    M, N, K = 200, 1024, 1024
    big = torch.full((M + 64, K), float("nan"), dtype=BF, device=DEV)
    big[:M] = torch.randn(M, K, generator=g).to(BF).to(DEV)
    w = ops.PackedWeight((torch.randn(N, K, generator=g) * 0.05).to(BF).to(DEV))
    ws = ops.AttnWorkspace(DEV)
    y1 = ops.linear_block(big[:M], w, workspace=ws)
    y2 = ops.linear_block(big[:M], w, workspace=ws)
    assert torch.equal(y1, y2) and bool(torch.isfinite(y1.float()).all())
