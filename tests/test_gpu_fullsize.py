"""Full-size GPU tests of the per-rank kernels of BASELINE configs[3] and [4] (run with -m gpu).

tests/test_gpu_ops.py checks the verify attention and the SnapKV select at configs[2]'s size (8B / 1B at TP1).  The
TP8 shards of configs[3] (Llama-3.1-70B: 8 query heads over ONE kv head, g = 8 -> two MFMA M tiles, B = 32, 32 K keys)
and configs[4] (Qwen2.5-32B: 5 query heads over one kv head, g = 5, B = 128, 64 K keys, fp8 cache, HND pages) take
other instantiations of the kernels (`paged_attn_kernel<128, 2, *>` with the split-KV + merge path) and had only run at
300-4100 keys.  The oracle cannot run these sizes in test time, so -- as at configs[2]'s size -- the size-independent
properties are asserted on the whole batch (softmax weights sum to one; exact linearity under a power-of-two scaling of
V; run-to-run determinism; SnapKV: exactly the stable descending top-k of the kernel's own pooled scores, gathered rows
bit-equal to the source rows) and ONE request is compared with the CPU oracle and the float64 dense reference under
the measured forward bound of tests/parity_util.py."""
import pytest
import torch

from oracle import flashinfer_ref as fr
from tests.conftest import parity_report
from tests.parity_util import check_attention, dense_attention_f64
from tests.test_gpu_ops import bits

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16
F8 = torch.float8_e4m3fn


@pytest.fixture(scope="module")
def ops():
    from magicdec_amd import ops as _ops
    _ops._lib.load()
    return _ops


@pytest.fixture(autouse=True)
def _capped_cpu_threads():
    from tests.parity_util import capped_threads
    with capped_threads():
        yield


def _cache(npages, KH, D, fp8, gen):
    """Logical NHD cache [pages, 2, 128, KH, D] generated on the device page by page (bounded scratch); fp8: the bytes
    ARE the ground truth (value = byte * scale), produced by rounding clamped normals."""
    out = torch.empty((npages, 2, 128, KH, D), dtype=F8 if fp8 else BF, device=DEV)
    step = 1024
    for p0 in range(0, npages, step):
        p1 = min(npages, p0 + step)
        x = torch.randn((p1 - p0, 2, 128, KH, D), device=DEV, generator=gen, dtype=torch.float32)
        out[p0:p1] = (x * 64.0).clamp_(-448.0, 448.0).to(F8) if fp8 else x.to(BF)
    return out


def _hnd(cache):
    return cache.permute(0, 1, 3, 2, 4).contiguous()


SHARDS = [
    # name, B, n, H, KH, D, S, fp8, layout
    ("cfg4-70b-tp8-shard-bf16-hnd", 32, 4, 8, 1, 128, 32645, False, "HND"),
    ("cfg5-qwen32b-tp8-shard-fp8-hnd", 128, 4, 5, 1, 128, 65444, True, "HND"),
    ("cfg2-8b-tp1-bf16-hnd", 32, 4, 32, 8, 128, 8068, False, "HND"),
]


@pytest.mark.parametrize("name,B,n,H,KH,D,S,fp8,layout", SHARDS, ids=[c[0] for c in SHARDS])
def test_verify_attention_full_size_shards(ops, name, B, n, H, KH, D, S, fp8, layout):
    mp = (S + 127) // 128
    gen = torch.Generator(device=DEV).manual_seed(11)
    cache = _cache(B * mp, KH, D, fp8, gen)
    q = torch.randn(B * n, H, D, device=DEV, generator=gen, dtype=torch.float32).to(BF)
    indices = torch.arange(B * mp, dtype=torch.int32, device=DEV)
    indptr = torch.arange(B + 1, dtype=torch.int32, device=DEV) * mp
    last = torch.full((B,), S - (mp - 1) * 128, dtype=torch.int32, device=DEV)
    qo = torch.arange(B + 1, dtype=torch.int32, device=DEV) * n
    ws = ops.AttnWorkspace(DEV)
    ks = (1.0 / 64.0) * (1 + 0.25 * torch.arange(KH, dtype=torch.float32)) if fp8 else None
    vs = (1.0 / 64.0) * torch.ones(KH, dtype=torch.float32) if fp8 else None
    run = lambda c, v_scale=vs: ops.paged_attention(
        q, _hnd(c) if layout == "HND" else c, qo, indices, indptr, last, n, mp, ws,
        kv_scales=(ks.to(DEV), v_scale.to(DEV)) if fp8 else None, kv_layout=layout)
    # (1) V == 1 everywhere -> the softmax weights sum to one
    ones = cache.clone()
    if fp8:
        ones[:, 1] = torch.full((1,), 64.0, device=DEV).to(F8)          # byte 64 * scale 1/64 == 1 exactly
    else:
        ones[:, 1] = 1.0
    o1 = run(ones)
    e1 = (o1.float() - 1.0).abs().max().item()
    parity_report(f"[attn] full-size {name}: V==1 -> max |o-1| = {e1:.3e} (bound (u_P+u_O)*1 = {2 ** -7:.3e})")
    assert e1 <= 1.05 * 2 ** -7
    del ones
    # (2) exact linearity under a power-of-two scaling of V, and determinism
    oa = run(cache)
    assert torch.equal(bits(oa.cpu()), bits(run(cache).cpu()))
    if fp8:
        ob = run(cache, vs * 0.5)                                         # the V scale is folded into the final 1/l
    else:
        v2 = cache.clone()
        v2[:, 1] = (cache[:, 1].float() * 0.5).to(BF)
        ob = run(v2)
        del v2
    assert torch.equal(oa.float() * 0.5, ob.float()), (oa.float() * 0.5 - ob.float()).abs().max().item()
    assert not torch.isnan(oa.float()).any()
    # (3) one request against the CPU oracle and the float64 dense reference
    b = B - 3
    sub = cache[b * mp:(b + 1) * mp].cpu()
    if fp8:
        sub = fr.dequantize_cache_fp8(sub, ks, vs)
    args = (q[b * n:(b + 1) * n].cpu(), sub, torch.tensor([0, n], dtype=torch.int32),
            torch.arange(mp, dtype=torch.int32), torch.tensor([0, mp], dtype=torch.int32), last[b:b + 1].cpu(), H, KH, D)
    oracle = fr.batch_prefill_paged(*args)
    ref64, bnd = dense_attention_f64(*args)
    check_attention(f"verify-full-size {name} (req {b})", oa[b * n:(b + 1) * n].to(BF), oracle, ref64, bnd)


def test_snapkv_select_full_size_cfg5_shard_fp8_hnd(ops):
    """SnapKV select of one configs[4] rank: g = 5, D = 128, one kv head, S = 65 440 keys + a 32-token window, budget 257,
    source cache fp8 e4m3 in HND pages -- the properties of test_snapkv_select_full_size_properties at that size."""
    B, KH, g, D, S, W, budget = 4, 1, 5, 128, 65440, 32, 257
    H = KH * g
    npg = (S + 127) // 128
    gen = torch.Generator(device=DEV).manual_seed(7)
    cache = _cache(B * npg, KH, D, True, gen)
    tail = S - (npg - 1) * 128
    raw = cache.view(torch.uint8)
    if tail < 128:
        for b in range(B):
            raw[(b + 1) * npg - 1, :, tail:] = 0x7F                      # e4m3fn NaN in the slots past S
    ks = torch.tensor([1.0 / 128.0]).to(DEV)
    vs = torch.tensor([1.0 / 32.0]).to(DEV)
    q = (torch.randn(B * W, H, D, device=DEV, generator=gen, dtype=torch.float32) * 0.3).to(BF)
    dppr = budget // 128 + 1
    indices = torch.arange(B * npg, dtype=torch.int32, device=DEV)
    indptr = (torch.arange(B + 1, dtype=torch.int32) * npg).to(DEV)
    dind = torch.arange(B * dppr, dtype=torch.int32, device=DEV)
    dptr = (torch.arange(B + 1, dtype=torch.int32) * dppr).to(DEV)
    dlast = torch.ones(B, dtype=torch.int32, device=DEV)
    ws = ops.AttnWorkspace(DEV)
    hnd = _hnd(cache)
    runs = []
    for _ in range(2):
        dcache = torch.zeros(B * dppr, 2, 128, KH, D, dtype=BF, device=DEV)
        idx, sc = ops.snapkv_select(q, hnd, indices, indptr, S, W, budget, 5, dcache, dind, dptr, dlast, ws,
                                    return_scores=True, kv_scales=(ks, vs), kv_layout="HND")
        runs.append((idx.cpu().long(), sc.cpu(), dcache.cpu()))
    (idx, sc, dk), (idx2, sc2, dk2) = runs
    assert torch.equal(idx, idx2) and torch.equal(bits(sc), bits(sc2)) and torch.equal(bits(dk), bits(dk2))
    assert not torch.isnan(sc.float()).any()
    topk = budget - W
    src = cache.cpu().float()
    for b in range(B):
        k_all = (src[b * npg:(b + 1) * npg, 0].reshape(-1, KH, D) * ks.cpu()[None, :, None]).to(BF)
        v_all = (src[b * npg:(b + 1) * npg, 1].reshape(-1, KH, D) * vs.cpu()[None, :, None]).to(BF)
        for h in range(KH):
            mine = idx[b, h]
            assert mine.min() >= 0 and mine.max() < S - W and len(set(mine.tolist())) == topk
            want = torch.sort(sc[b, h].float(), descending=True, stable=True).indices[:topk]
            assert torch.equal(mine, want), "not the stable descending top-k of the pooled scores"
            rows_k = dk[b * dppr:(b + 1) * dppr, 0].reshape(-1, KH, D)[:budget, h]
            rows_v = dk[b * dppr:(b + 1) * dppr, 1].reshape(-1, KH, D)[:budget, h]
            # the gather dequantises: draft rows are exactly bf16(byte * scale)
            assert torch.equal(bits(rows_k[:topk]), bits(k_all[mine, h]))
            assert torch.equal(bits(rows_v[:topk]), bits(v_all[mine, h]))
            assert torch.equal(bits(rows_k[topk:]), bits(k_all[S - W:S, h]))
            assert torch.equal(bits(rows_v[topk:]), bits(v_all[S - W:S, h]))
    parity_report(f"[snapkv] full-size cfg5 shard (g=5, D=128, S={S}, fp8 HND source): top-k order, gather and "
                  f"determinism hold on {B} requests")
