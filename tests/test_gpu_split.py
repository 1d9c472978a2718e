"""GPU parity of md_linear_fused_split (csrc/tilegemm.hip, FL_PARTIAL: the tile kernel with the K range also split over
workgroups + the fixed-order combine launch) -- run with -m gpu.

* the product: against the float64 matmul of the same bf16 operands under the gate of tests/test_gpu_gemm.py /
  test_gpu_fused.py (|err| <= u |exact| + 2K 2^-24 sum|x||w|), every tile shape (1x1, 1x2, 2x1, 2x2), ragged M, forced
  split counts, deterministic across launches;
* the fused combine (slices + bias + residual add + RMSNorm in ONE launch): BIT-EXACT against md_linear_fused_split
  followed by md_add_rmsnorm -- the reference's h = x + w2(...), rmsnorm(h) * w (Engine/SnapKV/model.py:260-278,464-469).
"""
import ctypes

import pytest
import torch

from tests.conftest import parity_report
from tests.parity_util import bf16_ulp
from tests.test_gpu_ops import bits

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    from magicdec_amd import ops as _ops
    _ops._lib.load()
    return _ops


def d(t):
    return t.to(DEV) if t is not None else None


# (M, N, K, forced S): the 1B w2 at one / two rows per request, its TP4 shard, the 8B wo / w2 at 32 rows (cfg2), ragged M,
# a 32-column product (1 x 1 tiles), an odd tile count (NT = 1), S forced to 1 (no split: one plane) and to 16
SHAPES = [(64, 2048, 8192, 0), (128, 2048, 8192, 0), (64, 2048, 2048, 0), (32, 4096, 4096, 0), (32, 4096, 14336, 0),
          (1, 32, 128, 0), (7, 96, 1024, 0), (33, 160, 2048, 0), (100, 1056, 1024, 0), (256, 512, 4096, 0),
          (64, 2048, 8192, 1), (64, 2048, 8192, 16), (40, 2048, 8192, 4)]


@pytest.mark.parametrize("M,N,K,S", SHAPES, ids=[f"M{m}-N{n}-K{k}-S{s}" for m, n, k, s in SHAPES])
@pytest.mark.parametrize("bias", [False, True], ids=["nobias", "bias"])
def test_split_linear_vs_exact(ops, M, N, K, S, bias):
    lib = ops._lib.load()
    g = torch.Generator().manual_seed(M * 131 + N * 7 + K + S)
    xfull = torch.randn(M, K + 64, generator=g).to(BF)
    x = xfull[:, :K]                                     # row stride != K
    w = (torch.randn(N, K, generator=g) * 0.05).to(BF)
    b = torch.randn(N, generator=g).to(BF) if bias else None
    ref = x.double() @ w.double().t() + (b.double() if bias else 0)
    mag = x.double().abs() @ w.double().abs().t() + (b.double().abs() if bias else 0)
    ws = ops.AttnWorkspace(DEV)
    pw = ops.PackedWeight(d(w))
    lib.md_debug_set_fused_split(ctypes.c_int(S))
    try:
        y = ops.fused_split_linear(d(xfull)[:, :K], pw, d(b), workspace=ws)
        y2 = ops.fused_split_linear(d(xfull)[:, :K], pw, d(b), workspace=ws)
    finally:
        lib.md_debug_set_fused_split(ctypes.c_int(0))
    assert y.shape == (M, N) and y.dtype == BF
    err = (y.cpu().double() - ref).abs()
    tol = 2.0 ** -8 * ref.abs() + 2 * K * 2.0 ** -24 * mag
    ulp = bf16_ulp(ref)
    parity_report(f"[split-gemm] M={M:3d} N={N:5d} K={K:5d} S={S or 'auto'} bias={int(bias)}  max err/tol "
                  f"{float((err / tol).max()):.3f}  max err {float((err / ulp).max()):.2f} ulp  != correctly rounded: "
                  f"{100 * float((y.cpu() != ref.to(BF)).double().mean()):.3f}%")
    assert bool((err <= tol).all())
    assert torch.equal(bits(y), bits(y2))               # deterministic: slices are added in slice order


@pytest.mark.parametrize("M,N,K", [(64, 2048, 8192), (128, 2048, 8192), (32, 4096, 14336), (5, 512, 1024), (200, 1024, 2048)])
@pytest.mark.parametrize("bias", [False, True], ids=["nobias", "bias"])
def test_split_add_rmsnorm_bit_exact_vs_unfused_sequence(ops, M, N, K, bias):
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(BF)
    w = (torch.randn(N, K, generator=g) * 0.05).to(BF)
    b = torch.randn(N, generator=g).to(BF) if bias else None
    r = torch.randn(M, N + 32, generator=g).to(BF)
    nw = (1 + 0.1 * torch.randn(N, generator=g)).to(BF)
    ws = ops.AttnWorkspace(DEV)
    pw = ops.PackedWeight(d(w))
    rd = d(r)[:, :N]                                     # residual with a row stride != N
    o = ops.fused_split_linear(d(x), pw, d(b), workspace=ws)
    h_want, y_want = ops.add_rmsnorm(rd.contiguous(), o, d(nw), 1e-5)
    h, y = ops.fused_split_linear_add_rmsnorm(d(x), pw, rd, d(nw), 1e-5, bias=d(b), workspace=ws)
    assert torch.equal(bits(h), bits(h_want)) and torch.equal(bits(y), bits(y_want))
    assert not bool(torch.isnan(y.float()).any())


def test_split_rejects_bad_arguments(ops):
    ws = ops.AttnWorkspace(DEV)
    x = torch.zeros(4, 256, device=DEV, dtype=BF)
    w13 = ops.PackedWeight(torch.zeros(64, 256, device=DEV, dtype=BF), swiglu=True)
    with pytest.raises(ValueError):
        ops.fused_split_linear(x, w13, workspace=ws)                       # packed for SwiGLU
    pw = ops.PackedWeight(torch.zeros(64, 256, device=DEV, dtype=BF))
    with pytest.raises(ValueError):
        ops.fused_split_linear(x, pw, workspace=None)                      # no workspace
    lib = ops._lib.load()
    assert lib.md_linear_fused_split_workspace_bytes(4, 64, 200) == 0      # K % 128 != 0
    rc = lib.md_linear_fused_split(ctypes.c_void_p(x.data_ptr()), 256, ctypes.c_void_p(pw.data.data_ptr()), None,
                                   ctypes.c_void_p(x.data_ptr()), 64, 4, 64, 256, None, 0, None)
    assert rc != 0 and b"workspace" in lib.md_last_error_string()
