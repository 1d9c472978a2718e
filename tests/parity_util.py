"""Measured-error gates shared by the GPU parity tests.

Attention (bf16 in/out, fp32 softmax): the gate is the a-priori forward-error bound of the algorithm every
tensor-core implementation of this op uses (flashinfer included): P is rounded to the input dtype for the P.V
matrix product, the output is rounded once --

    |o_hip - o_exact| <= (u_P + u_O) * sum_i p_i |v_i|,     u_P = u_O = 2^-8 (bf16: 8 significant bits, round-to-nearest)

evaluated per element against a float64 dense reference on the same inputs.  ATTN_BOUND_SLACK covers the fp32
accumulation order, the fp32 score error through exp and v_exp_f32's own error (all << 2^-8).  The achieved error is
reported in units of that bound and in bf16 ulps at the output tensor's scale (max |ref|), next to the oracle's (fp32,
un-rounded P) own error, and the fraction of elements inside the tighter `2*err_oracle + 1 ulp(ref element)` band is
logged.
"""
import math

import torch

from oracle import flashinfer_ref as fr
from tests.conftest import parity_report

U_BF16 = 2.0 ** -8            # unit round-off of bf16: relative error of round-to-nearest with 8 significant bits
ATTN_BOUND_SLACK = 1.05


class capped_threads:
    """`with capped_threads():` -- at most `n` intra-op threads for a float64 YARDSTICK computation (never for an
    oracle whose bits are compared): on the GPU boxes' 256-core hosts torch's default pool makes small float64
    matmuls tens of times slower than 16 threads do (bench.py's cpu_baseline measured the same)."""

    def __init__(self, n=16):
        self.n = n

    def __enter__(self):
        self.prev = torch.get_num_threads()
        if self.prev > self.n:
            torch.set_num_threads(self.n)

    def __exit__(self, *exc):
        if torch.get_num_threads() != self.prev:
            torch.set_num_threads(self.prev)


def bf16_ulp(x):
    """Spacing of bf16 numbers at |x| (tensor, float64)."""
    ax = x.abs().clamp_min(2.0 ** -126)
    return torch.exp2(torch.floor(torch.log2(ax)) - 7)


def dense_attention_f64(q, cache, qo_indptr, indices, indptr, last, H, KH, D, causal=True, sm_scale=None):
    """float64 softmax(q k^T * scale) v on the paged cache (any float dtype; an fp8 cache is passed as the float32
    tensor of exact byte * scale products).  Returns (o [rows,H,D], bound [rows,H,D] = sum_i p_i |v_i|)."""
    g = H // KH
    if sm_scale is None:
        sm_scale = 1.0 / math.sqrt(D)
    rows = q.shape[0]
    out = torch.zeros(rows, H, D, dtype=torch.float64)
    bnd = torch.zeros(rows, H, D, dtype=torch.float64)
    B = indptr.numel() - 1
    with capped_threads():
        _dense_attention_f64_rows(out, bnd, q, cache, qo_indptr, indices, indptr, last, KH, g, D, causal, sm_scale, B)
    return out, bnd


def _dense_attention_f64_rows(out, bnd, q, cache, qo_indptr, indices, indptr, last, KH, g, D, causal, sm_scale, B):
    H = KH * g
    for b in range(B):
        q0, q1 = int(qo_indptr[b]), int(qo_indptr[b + 1])
        m = q1 - q0
        if m == 0:
            continue
        k, v = fr.gather_request_kv(cache, indices, indptr, last, b)
        ln = k.shape[0]
        if ln == 0:
            continue
        qf = q[q0:q1].double().view(m, KH, g, D)
        s = torch.einsum("mhgd,lhd->hgml", qf, k.double()) * sm_scale
        if causal:
            pos = torch.arange(ln).view(1, 1, 1, ln)
            lim = (ln - m + torch.arange(m)).view(1, 1, m, 1)
            s = s.masked_fill(pos > lim, float("-inf"))
        p = torch.nan_to_num(torch.softmax(s, dim=-1), nan=0.0)
        out[q0:q1] = torch.einsum("hgml,lhd->mhgd", p, v.double()).reshape(m, H, D)
        bnd[q0:q1] = torch.einsum("hgml,lhd->mhgd", p, v.double().abs()).reshape(m, H, D)


def check_attention(tag, out_hip, out_oracle, ref64, bound64):
    """Gate + report.  out_hip / out_oracle: bf16 (any device) ; ref64 / bound64 from dense_attention_f64."""
    o = out_hip.detach().cpu().double()
    assert not torch.isnan(o).any(), f"{tag}: NaN in the output"
    err = (o - ref64).abs()
    tol = ATTN_BOUND_SLACK * 2.0 * U_BF16 * bound64 + 1e-30
    ratio = (err / tol.clamp_min(1e-300))
    ratio = torch.where(bound64 > 0, ratio, torch.zeros_like(ratio))
    ulp = bf16_ulp(ref64)
    ulp_t = bf16_ulp(ref64.abs().max()).item() if ref64.numel() else 1.0     # ulp at the tensor's scale
    err_or = (out_oracle.detach().cpu().double() - ref64).abs()
    band = (err <= 2 * err_or + ulp).double().mean().item()
    worst = ratio.max().item()
    parity_report(f"[attn] {tag:36s} max err/bound {worst:5.3f}  mean {ratio.mean().item():5.3f}  | "
                  f"max err {err.max().item():.3e} = {err.max().item() / ulp_t:5.2f} ulp(max|ref|)  "
                  f"oracle(fp32 P) {err_or.max().item() / ulp_t:5.2f}  | in 2*oracle+1ulp(elem) band: {100 * band:6.2f}%")
    # elements whose exact value is 0 (empty key set) must be exactly 0
    assert bool((o[bound64 == 0] == 0).all()), f"{tag}: non-zero output for an empty key set"
    assert worst <= 1.0, f"{tag}: error {worst:.3f} x the forward bound (u_P + u_O) * sum p|v|"
    return worst
