"""GPU tests of the fp8 (OCP e4m3fn) KV-cache path (run with -m gpu).

The reference has no fp8 cache (BASELINE.json configs[4] / SURVEY.md section 8f-2 ask for it as the CDNA4 path), so the
specification is oracle/flashinfer_ref.py's `quantize_fp8` / `dequantize_cache_fp8`:
  * the quantiser (md_append_paged_kv, md_rope_append) is byte work: BIT-EXACT against the oracle;
  * attention / SnapKV scoring over an fp8 cache must equal the oracle run on the exactly-dequantised cache to the
    same tolerance as the bf16 kernels (the bytes convert exactly to bf16; scales are folded in fp32);
  * end to end the accuracy gate is the relative error of teacher-forced verify logits against the bf16-cache engine.
"""
import os
import tempfile

import numpy as np
import pytest
import torch

from oracle import flashinfer_ref as fr
from oracle import magicdec_ref as mr
from tests import golden_cfg as gc
from tests.conftest import parity_report
from tests.parity_util import check_attention, dense_attention_f64
from tests.test_gpu_ops import _ulp_close, bits, case_seed, make_paged

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16
F8 = torch.float8_e4m3fn


@pytest.fixture(scope="module")
def ops():
    from magicdec_amd import ops as _ops
    _ops._lib.load()
    return _ops


def u8(t):
    return t.contiguous().view(torch.uint8)


def quantize_cache(cache, k_scale, v_scale):
    """bf16 [pages,2,ps,KH,D] -> e4m3fn cache with the oracle's quantiser."""
    P, _, ps, KH, D = cache.shape
    out = torch.empty(cache.shape, dtype=F8)
    out[:, 0] = fr.quantize_fp8(cache[:, 0].reshape(-1, KH, D), k_scale).view(P, ps, KH, D)
    out[:, 1] = fr.quantize_fp8(cache[:, 1].reshape(-1, KH, D), v_scale).view(P, ps, KH, D)
    return out


def test_fp8_append_and_fused_rope_append_bit_exact(ops):
    """Quantised bytes equal the oracle's, including saturation at +-448, subnormals and signed zeros."""
    B, n, H, KH, D = 3, 4, 8, 2, 64
    lens = [200, 131, 4]
    _, indices, indptr, last, _ = make_paged(B, lens, KH, D, seed=9, scatter=True)
    npages = int(indices.max()) + 3
    tab_ref = fr.rope_table(2048, D, 10000.0, 1.0)
    tab = ops.RopeTable(2048, D, 10000.0, 1.0, device=DEV)
    g = torch.Generator().manual_seed(4)
    qkv = torch.randn(B * n, (H + 2 * KH) * D, generator=g)
    qkv[:, ::7] *= 40.0            # some values beyond 448*scale -> saturate
    qkv[:, 3::11] *= 1e-3          # some in the e4m3 subnormal range
    qkv[0, H * D + 5] = 0.0
    qkv[1, H * D + 6] = -0.0
    qkv = qkv.to(BF)
    q = qkv[:, :H * D].unflatten(1, (H, D))
    k = qkv[:, H * D:(H + KH) * D].unflatten(1, (KH, D))
    v = qkv[:, (H + KH) * D:].unflatten(1, (KH, D))
    ks = torch.tensor([0.037, 0.0625], dtype=torch.float32)
    vs = torch.tensor([0.011, 0.29], dtype=torch.float32)
    ip = torch.arange(B + 1, dtype=torch.int32) * n
    offsets = torch.tensor([l - n for l in lens], dtype=torch.int32)
    rq, rk = fr.apply_rope(q, k, ip, offsets, tab_ref)
    ref8 = torch.zeros(npages, 2, 128, KH, D, dtype=F8)
    fr.append_paged_kv_cache_fp8(rk, v, ip, ref8, indices, indptr, last, ks, vs)
    ref16 = torch.zeros(npages, 2, 128, KH, D, dtype=BF)
    fr.append_paged_kv_cache(rk, v, ip, ref16, indices, indptr, last)
    assert (ref8.float().abs() == 448).any() and ((ref8.float().abs() < 2 ** -6) & (ref8.float() != 0)).any()
    d = lambda t: t.to(DEV)
    dqkv = d(qkv)
    dq = dqkv[:, :H * D].unflatten(1, (H, D))
    dk = dqkv[:, H * D:(H + KH) * D].unflatten(1, (KH, D))
    dv = dqkv[:, (H + KH) * D:].unflatten(1, (KH, D))
    scales = (d(ks), d(vs))
    c1 = torch.zeros(npages, 2, 128, KH, D, dtype=F8, device=DEV)
    _, ok = ops.rope(dq, dk, d(ip), d(offsets), tab)
    ops.update_kv(ok, dv, d(ip), c1, d(indices), d(indptr), d(last), kv_scales=scales)
    assert torch.equal(u8(c1.cpu()), u8(ref8))
    c2 = torch.zeros(npages, 2, 128, KH, D, dtype=F8, device=DEV)
    c3 = torch.zeros(npages, 2, 128, KH, D, dtype=BF, device=DEV)
    oq = ops.rope_append(dq, dk, dv, d(ip), d(offsets), tab, c2, d(indices), d(indptr), d(last), c3, d(indices),
                         d(indptr), d(last), kv_scales=scales)
    assert torch.equal(bits(oq.cpu()), bits(rq))
    assert torch.equal(u8(c2.cpu()), u8(ref8))
    assert torch.equal(bits(c3.cpu()), bits(ref16))       # the second (draft) cache is bf16
    with pytest.raises(ValueError):
        ops.update_kv(ok, dv, d(ip), c1, d(indices), d(indptr), d(last))      # fp8 cache without scales


FP8_ATTN_CASES = [
    ("verify-8b-shape", 2, 4, 8, 2, 128, [300, 257], True, False),
    ("verify-tile-edge-33", 1, 4, 4, 1, 128, [33], True, False),
    ("verify-ragged-scattered-pages", 3, 4, 32, 8, 128, [1000, 129, 640], True, True),
    ("verify-split-kv", 2, 4, 8, 2, 128, [8069, 7000], True, False),
    ("qwen32b-tp8-shard-g5", 2, 4, 5, 1, 128, [4100, 3000], True, False),
    ("draft-1row-d64", 4, 1, 32, 8, 64, [260, 258, 300, 257], True, False),
    ("draft-2row-d64", 4, 2, 8, 2, 64, [260, 258, 300, 257], True, False),
    ("g8-two-mtiles", 2, 4, 16, 2, 128, [500, 300], True, False),
    ("prefill-chunk-128", 2, 128, 8, 2, 128, [384, 384], True, False),
    ("prefill-last-chunk-32", 2, 32, 8, 2, 128, [160, 160], True, False),
    ("prefill-d64", 2, 128, 8, 2, 64, [256, 256], True, False),
    ("non-causal", 2, 4, 8, 2, 128, [300, 257], False, False),
    ("empty-request", 2, 4, 8, 2, 128, [0, 200], True, False),
]


@pytest.mark.parametrize("name,B,n,H,KH,D,lens,causal,scatter", FP8_ATTN_CASES, ids=[c[0] for c in FP8_ATTN_CASES])
def test_fp8_paged_attention_vs_oracle(ops, name, B, n, H, KH, D, lens, causal, scatter):
    """Same measured bar as the bf16 kernel (tests/parity_util.py), against a float64 dense reference on the EXACTLY
    dequantised cache (byte * scale): the bytes convert exactly to bf16, the K scale is folded into the softmax
    scale and the V scale into the final 1/l in fp32."""
    cache, indices, indptr, last, max_pages = make_paged(B, lens, KH, D, seed=case_seed(name), scatter=scatter)
    ks = 0.013 * (1 + torch.arange(KH, dtype=torch.float32))
    vs = 0.021 / (1 + torch.arange(KH, dtype=torch.float32))
    c8 = quantize_cache(cache, ks, vs)
    g = torch.Generator().manual_seed(1)
    q = torch.randn(B * n, H, D, generator=g).to(BF)
    qo = torch.arange(B + 1, dtype=torch.int32) * n
    deq = fr.dequantize_cache_fp8(c8, ks, vs)
    oracle = fr.batch_prefill_paged(q, deq, qo, indices, indptr, last, H, KH, D, causal=causal)
    ref64, bnd = dense_attention_f64(q, deq, qo, indices, indptr, last, H, KH, D, causal=causal)
    ws = ops.AttnWorkspace(DEV)
    out = ops.paged_attention(q.to(DEV), c8.to(DEV), qo.to(DEV), indices.to(DEV), indptr.to(DEV), last.to(DEV), n,
                              max_pages, ws, causal=causal, kv_scales=(ks.to(DEV), vs.to(DEV)))
    check_attention("fp8/" + name, out, oracle, ref64, bnd)


def test_fp8_attention_ignores_garbage_beyond_length(ops):
    """e4m3fn NaN bytes (0x7f / 0xff) in stale slots past a request's length must not leak."""
    B, n, H, KH, D = 2, 4, 8, 2, 128
    cache, indices, indptr, last, max_pages = make_paged(B, [200, 130], KH, D, seed=3)
    ks = torch.tensor([0.02, 0.03])
    vs = torch.tensor([0.02, 0.01])
    c8 = quantize_cache(cache, ks, vs)
    dirty = u8(c8).clone()
    for b, ln in enumerate([200, 130]):
        pg = int(indices[int(indptr[b]) + ln // 128])
        dirty[pg, :, ln % 128:] = 0x7F
    dirty = dirty.view(F8)
    q = torch.randn(B * n, H, D, generator=torch.Generator().manual_seed(8)).to(BF)
    qo = torch.arange(B + 1, dtype=torch.int32) * n
    ws = ops.AttnWorkspace(DEV)
    sc = (ks.to(DEV), vs.to(DEV))
    a = ops.paged_attention(q.to(DEV), c8.to(DEV), qo.to(DEV), indices.to(DEV), indptr.to(DEV), last.to(DEV), n,
                            max_pages, ws, kv_scales=sc)
    b_ = ops.paged_attention(q.to(DEV), dirty.to(DEV), qo.to(DEV), indices.to(DEV), indptr.to(DEV), last.to(DEV), n,
                             max_pages, ws, kv_scales=sc)
    assert not torch.isnan(b_.float()).any()
    assert torch.equal(bits(a.cpu()), bits(b_.cpu()))


@pytest.mark.parametrize("tag", ["g5", "g4d128"])
def test_fp8_snapkv_select(ops, tag, golden_dir):
    """SnapKV select reading an fp8 full cache: scores equal the oracle's on the dequantised K up to isolated bf16
    ulp flips (>= 97% bit-equal, <= 4 ulp); order / tie-break exact on our scores; selected set equals the oracle's
    up to threshold near-ties; gathered draft rows are exactly bf16(byte * scale)."""
    z = np.load(f"{golden_dir}/snapkv_select.npz")
    g, KH, D, S, budget, B, W = [int(x) for x in z[f"{tag}_meta"]]
    q = gc.from_bits(z[f"{tag}_q"])
    k = gc.from_bits(z[f"{tag}_k"])
    v = gc.from_bits(z[f"{tag}_v"])
    ks = 0.017 * (1 + 0.5 * torch.arange(KH, dtype=torch.float32))
    vs = 0.009 * (1 + torch.arange(KH, dtype=torch.float32))
    npg = (S + 127) // 128
    cache = torch.zeros(B * npg, 2, 128, KH, D, dtype=BF)
    for b in range(B):
        kk = torch.zeros(npg * 128, KH, D, dtype=BF)
        vv = torch.zeros(npg * 128, KH, D, dtype=BF)
        kk[:S], vv[:S] = k[b], v[b]
        cache[b * npg:(b + 1) * npg, 0] = kk.view(npg, 128, KH, D)
        cache[b * npg:(b + 1) * npg, 1] = vv.view(npg, 128, KH, D)
    c8 = quantize_cache(cache, ks, vs)
    deq = fr.dequantize_cache_fp8(c8, ks, vs)
    n = q.shape[0] // B
    dppr = budget // 128 + 1
    dcache = torch.zeros(B * dppr, 2, 128, KH, D, dtype=BF, device=DEV)
    ws = ops.AttnWorkspace(DEV)
    idx, sc = ops.snapkv_select(q.to(DEV), c8.to(DEV), torch.arange(B * npg, dtype=torch.int32, device=DEV),
                                (torch.arange(B + 1, dtype=torch.int32) * npg).to(DEV), S, W, budget, 5, dcache,
                                torch.arange(B * dppr, dtype=torch.int32, device=DEV),
                                (torch.arange(B + 1, dtype=torch.int32) * dppr).to(DEV),
                                torch.ones(B, dtype=torch.int32, device=DEV), ws, return_scores=True,
                                kv_scales=(ks.to(DEV), vs.to(DEV)))
    idx, sc = idx.cpu().long(), sc.cpu()
    topk = budget - W
    dk = dcache.cpu()
    for b in range(B):
        kd = deq[b * npg:(b + 1) * npg, 0].reshape(-1, KH, D)[:S]
        vd = deq[b * npg:(b + 1) * npg, 1].reshape(-1, KH, D)[:S]
        ref_scores = mr.snapkv_scores(q[b * n:(b + 1) * n], kd, g, W)
        exact = (bits(sc[b]) == bits(ref_scores)).float().mean().item()
        parity_report(f"[snapkv] fp8/{tag} request {b}: pooled scores bit-equal to the oracle on the dequantised K: "
                      f"{100 * exact:.3f}%")
        assert exact >= 0.97, exact
        assert _ulp_close(sc[b], ref_scores, ulps=4)
        ref_idx = mr.topk_desc_stable(ref_scores, topk)
        for h in range(KH):
            s = sc[b, h].float()
            mine = idx[b, h]
            assert torch.equal(mine, torch.sort(s, descending=True, stable=True).indices[:topk]), "order / tie-break"
            diff = set(mine.tolist()) ^ set(ref_idx[h].tolist())
            thr = ref_scores[h].float()[ref_idx[h]].min()
            for p in diff:
                assert abs(ref_scores[h, p].float() - thr) <= 4 * thr * 2 ** -8 + 1e-30, (b, h, p)
            rows_k = dk[b * dppr:(b + 1) * dppr, 0].reshape(-1, KH, D)[:budget, h]
            rows_v = dk[b * dppr:(b + 1) * dppr, 1].reshape(-1, KH, D)[:budget, h]
            assert torch.equal(bits(rows_k[:topk]), bits(kd[mine, h].to(BF)))
            assert torch.equal(bits(rows_v[:topk]), bits(vd[mine, h].to(BF)))
            assert torch.equal(bits(rows_k[topk:]), bits(kd[S - W:, h].to(BF)))
            assert torch.equal(bits(rows_v[topk:]), bits(vd[S - W:, h].to(BF)))


@pytest.fixture(scope="module")
def ckpt_dir():
    from magicdec_amd.Engine import model_core
    d = tempfile.mkdtemp(prefix="md_ckpt_")
    for name in gc.TINY:
        cfg, sd = gc.tiny(name)
        os.makedirs(os.path.join(d, name), exist_ok=True)
        torch.save(sd, os.path.join(d, name, "model.pth"))
        model_core.transformer_configs[name] = gc.config_kwargs(cfg)
    return d


def test_fp8_engine_accuracy_gate_and_selfspec_loop(ckpt_dir):
    """Accuracy gate of the fp8 cache end to end: teacher-forced verify logits of the fp8 engine vs the bf16 engine
    (same weights, same prompt) differ by <= 5% relative L2; the self-spec SnapKV loop then runs on the fp8 cache (with
    and without hipGraphs) and generates tokens, graphs == eager bit for bit."""
    from magicdec_amd import harness
    from magicdec_amd.Engine.SnapKV.backend import LMBackend
    ids = next(iter(gc.synthetic_batches())).to(DEV)
    logits, outs = {}, {}
    for kvd, graphs in (("bf16", False), ("fp8", False), ("fp8", True)):
        eng = LMBackend(dtype=torch.bfloat16, device=DEV, dec_len=gc.GAMMA + 1, draft_dec_len=1)
        eng.load_model(os.path.join(ckpt_dir, "tinytgt", "model.pth"), use_tp=False)
        if graphs:
            eng.compile()
        eng.setup_caches(max_batch_size=gc.B, max_seq_length=gc.MAX_LEN, draft_budget=gc.BUDGET, kv_dtype=kvd)
        eng.encode(ids)
        kvc = eng.model.layers[0].attention.kv_cache
        assert kvc.kv_cache.dtype == (F8 if kvd == "fp8" else BF) and kvc.draft_cache.dtype == BF
        if kvd == "fp8":
            assert kvc.calibrated and (kvc.k_scale != 1).all()
        eng.verify(ids[:, :gc.GAMMA + 1].clone())
        logits[(kvd, graphs)] = eng.model._last_logits.float().cpu()
        st, _ = harness.run_selfspec_batch(eng, ids, gc.GAMMA, gc.MAX_LEN, gc.EOT_1, gc.EOT_2, False)
        outs[(kvd, graphs)] = st
        assert st.iters > 0 and (st.num_nodes.cpu() > ids.shape[1]).all()
    ref = logits[("bf16", False)]
    rel = (logits[("fp8", False)] - ref).norm() / ref.norm()
    assert rel <= 5e-2, rel
    assert torch.equal(logits[("fp8", False)], logits[("fp8", True)])
    assert torch.equal(outs[("fp8", False)].output.cpu(), outs[("fp8", True)].output.cpu())
