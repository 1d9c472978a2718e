"""GPU tests of the HND page layout option (MD_KV_LAYOUT_HND: cache[page][2][KH][page_size][D], run with -m gpu).

The layout changes addresses only -- which bytes a lane loads or stores -- never the arithmetic, so the bar is the
strongest one available: everything a kernel produces from an HND cache is BIT-IDENTICAL to what the same kernel
produces from the NHD cache holding the same values (NHD results are gated against the oracle / golden vectors in
test_gpu_ops.py, test_gpu_fp8.py and test_gpu_engine.py), and every cache a kernel writes equals the NHD cache
permuted.  The engine tests then run the HND engine in lock-step against the oracle itself."""
import numpy as np
import pytest
import torch

from oracle import harness_ref as hr
from oracle import magicdec_ref as mr
from tests import golden_cfg as gc
from tests.conftest import parity_report
from tests.test_gpu_ops import ATTN_CASES, bits, case_seed, check_append_overflow, make_paged
from tests.test_gpu_fp8 import quantize_cache

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _capped_cpu_threads():
    """The oracle runs of these tests are yardsticks behind tolerance gates (tokens within the measured logit gate,
    integer state teacher-forced), never bit-compared with the GPU: cap torch's intra-op pool, which on the GPU boxes'
    256-core hosts makes the small CPU matmuls of the oracle tens of times slower (tests/parity_util.capped_threads)."""
    from tests.parity_util import capped_threads
    with capped_threads():
        yield

DEV = "cuda"
BF = torch.bfloat16
F8 = torch.float8_e4m3fn


@pytest.fixture(scope="module")
def ops():
    from magicdec_amd import ops as _ops
    _ops._lib.load()      # fail loudly if the HIP library is missing
    return _ops


def to_hnd(cache):
    """[pages, 2, page_size, KH, D] -> [pages, 2, KH, page_size, D], same values."""
    return cache.permute(0, 1, 3, 2, 4).contiguous()


def raw(t):
    t = t.contiguous()
    return t.view(torch.uint8) if t.dtype == F8 else t.view(torch.int16)


HND_ATTN = [c for c in ATTN_CASES if c[0] in (
    "verify-8b-shape", "verify-tile-edge-33", "verify-short", "verify-ragged-scattered-pages", "verify-split-kv",
    "draft-1row-d64", "draft-2row-d64", "g8-two-mtiles", "g5-padded-mtile", "mha-g1", "prefill-chunk-128",
    "prefill-last-chunk-32", "prefill-d64", "non-causal", "empty-request")]


@pytest.mark.parametrize("fp8", [False, True], ids=["bf16", "fp8"])
@pytest.mark.parametrize("name,B,n,H,KH,D,lens,causal,scatter", HND_ATTN, ids=[c[0] for c in HND_ATTN])
def test_hnd_attention_bit_identical_to_nhd(ops, name, B, n, H, KH, D, lens, causal, scatter, fp8):
    cache, indices, indptr, last, max_pages = make_paged(B, lens, KH, D, seed=case_seed(name), scatter=scatter)
    scales = None
    if fp8:
        ks = 0.02 * (1 + 0.25 * torch.arange(KH, dtype=torch.float32))
        vs = 0.015 * (1 + 0.5 * torch.arange(KH, dtype=torch.float32))
        cache = quantize_cache(cache, ks, vs)
        scales = (ks.to(DEV), vs.to(DEV))
    g = torch.Generator().manual_seed(1)
    q = torch.randn(B * n, H, D, generator=g).to(BF).to(DEV)
    qo = (torch.arange(B + 1, dtype=torch.int32) * n).to(DEV)
    ws = ops.AttnWorkspace(DEV)
    args = (qo, indices.to(DEV), indptr.to(DEV), last.to(DEV), n, max_pages, ws)
    nhd = ops.paged_attention(q, cache.to(DEV), *args, causal=causal, kv_scales=scales)
    hnd = ops.paged_attention(q, to_hnd(cache).to(DEV), *args, causal=causal, kv_scales=scales, kv_layout="HND")
    assert not torch.isnan(hnd.float()).any()
    assert torch.equal(bits(hnd.cpu()), bits(nhd.cpu())), name


def test_hnd_layout_is_not_nhd(ops):
    """Guard against a silently ignored flag: reading an HND cache as NHD gives different numbers."""
    B, n, H, KH, D, lens = 2, 4, 8, 2, 128, [300, 257]
    cache, indices, indptr, last, max_pages = make_paged(B, lens, KH, D, seed=3)
    q = torch.randn(B * n, H, D, generator=torch.Generator().manual_seed(1)).to(BF).to(DEV)
    qo = (torch.arange(B + 1, dtype=torch.int32) * n).to(DEV)
    ws = ops.AttnWorkspace(DEV)
    args = (qo, indices.to(DEV), indptr.to(DEV), last.to(DEV), n, max_pages, ws)
    h = to_hnd(cache).to(DEV)
    right = ops.paged_attention(q, h, *args, kv_layout="HND")
    wrong = ops.paged_attention(q, h.view(cache.shape), *args)
    assert not torch.equal(bits(right.cpu()), bits(wrong.cpu()))
    with pytest.raises(ValueError):
        ops.paged_attention(q, h, *args, kv_layout="NDH")


@pytest.mark.parametrize("fp8", [False, True], ids=["bf16", "fp8"])
def test_hnd_append_and_fused_rope_append(ops, fp8):
    """md_append_paged_kv / md_rope_append into an HND first cache: the bytes are the NHD result permuted (scattered
    pages, rows crossing a page boundary, a 4-row request); the second (draft) cache stays NHD."""
    B, n, H, KH, D = 3, 4, 8, 2, 64
    lens = [200, 131, 4]
    cache, indices, indptr, last, _ = make_paged(B, lens, KH, D, seed=9, scatter=True)
    tab = ops.RopeTable(2048, D, 10000.0, 1.0, device=DEV)
    g = torch.Generator().manual_seed(4)
    qkv = torch.randn(B * n, (H + 2 * KH) * D, generator=g).to(BF)
    d = lambda t: t.to(DEV).clone()
    dqkv = d(qkv)
    dq = dqkv[:, :H * D].unflatten(1, (H, D))
    dk = dqkv[:, H * D:(H + KH) * D].unflatten(1, (KH, D))
    dv = dqkv[:, (H + KH) * D:].unflatten(1, (KH, D))
    ip = d(torch.arange(B + 1, dtype=torch.int32) * n)
    offsets = d(torch.tensor([l - n for l in lens], dtype=torch.int32))
    scales = None
    if fp8:
        scales = (d(torch.tensor([0.037, 0.0625])), d(torch.tensor([0.011, 0.29])))
        cache = torch.zeros(cache.shape, dtype=F8)
    tabs = (d(indices), d(indptr), d(last))
    # separate ops
    c_n, c_h = d(cache), d(to_hnd(cache))
    _, ok = ops.rope(dq, dk, ip, offsets, tab)
    ops.update_kv(ok, dv, ip, c_n, *tabs, kv_scales=scales)
    ops.update_kv(ok, dv, ip, c_h, *tabs, kv_scales=scales, kv_layout="HND")
    assert torch.equal(raw(c_h.cpu()), raw(to_hnd(c_n.cpu())))
    assert not torch.equal(raw(c_n.cpu()), raw(d(cache).cpu()))              # something was written
    # fused, two caches
    c_n2, c_h2 = d(cache), d(to_hnd(cache))
    s_n = torch.zeros(cache.shape, dtype=BF, device=DEV)
    s_h = torch.zeros(cache.shape, dtype=BF, device=DEV)
    q_n = ops.rope_append(dq, dk, dv, ip, offsets, tab, c_n2, *tabs, s_n, *tabs, kv_scales=scales)
    q_h = ops.rope_append(dq, dk, dv, ip, offsets, tab, c_h2, *tabs, s_h, *tabs, kv_scales=scales, kv_layout="HND")
    assert torch.equal(bits(q_h.cpu()), bits(q_n.cpu()))
    assert torch.equal(raw(c_h2.cpu()), raw(to_hnd(c_n2.cpu())))
    assert torch.equal(raw(c_n2.cpu()), raw(c_n.cpu()))
    assert torch.equal(bits(s_h.cpu()), bits(s_n.cpu()))                     # second cache: NHD in both runs


@pytest.mark.parametrize("fp8", [False, True], ids=["bf16", "fp8"])
def test_hnd_append_beyond_mapped_pages_is_dropped_and_counted(ops, fp8):
    check_append_overflow(ops, "HND", fp8)


@pytest.mark.parametrize("tag,fp8", [("g4", False), ("g5", False), ("g4d128", False), ("g5", True), ("g4d128", True)])
def test_hnd_snapkv_select_bit_identical_to_nhd(ops, tag, fp8, golden_dir):
    """md_snapkv_select reading an HND source cache: pooled scores, selected positions and the gathered (bf16 NHD)
    draft cache equal the NHD run bit for bit, on the reference's fixture inputs."""
    z = np.load(f"{golden_dir}/snapkv_select.npz")
    g, KH, D, S, budget, B, W = [int(x) for x in z[f"{tag}_meta"]]
    q = gc.from_bits(z[f"{tag}_q"])
    k = gc.from_bits(z[f"{tag}_k"])
    v = gc.from_bits(z[f"{tag}_v"])
    npg = (S + 127) // 128
    cache = torch.zeros(B * npg, 2, 128, KH, D, dtype=BF)
    for b in range(B):
        kk = torch.zeros(npg * 128, KH, D, dtype=BF)
        vv = torch.zeros(npg * 128, KH, D, dtype=BF)
        kk[:S], vv[:S] = k[b], v[b]
        cache[b * npg:(b + 1) * npg, 0] = kk.view(npg, 128, KH, D)
        cache[b * npg:(b + 1) * npg, 1] = vv.view(npg, 128, KH, D)
    scales = None
    if fp8:
        ks = 0.017 * (1 + 0.5 * torch.arange(KH, dtype=torch.float32))
        vs = 0.009 * (1 + torch.arange(KH, dtype=torch.float32))
        cache = quantize_cache(cache, ks, vs)
        scales = (ks.to(DEV), vs.to(DEV))
    dppr = budget // 128 + 1
    ws = ops.AttnWorkspace(DEV)

    def run(src, layout):
        dcache = torch.zeros(B * dppr, 2, 128, KH, D, dtype=BF, device=DEV)
        idx, sc = ops.snapkv_select(q.to(DEV), src.to(DEV), torch.arange(B * npg, dtype=torch.int32, device=DEV),
                                    (torch.arange(B + 1, dtype=torch.int32) * npg).to(DEV), S, W, budget, 5, dcache,
                                    torch.arange(B * dppr, dtype=torch.int32, device=DEV),
                                    (torch.arange(B + 1, dtype=torch.int32) * dppr).to(DEV),
                                    torch.ones(B, dtype=torch.int32, device=DEV), ws, return_scores=True,
                                    kv_scales=scales, kv_layout=layout)
        return idx.cpu(), sc.cpu(), dcache.cpu()
    i_n, s_n, d_n = run(cache, "NHD")
    i_h, s_h, d_h = run(to_hnd(cache), "HND")
    assert torch.equal(bits(s_h), bits(s_n))
    assert torch.equal(i_h, i_n)
    assert torch.equal(bits(d_h), bits(d_n))
    assert d_n.float().abs().sum().item() > 0


# ----------------------------------------------------------------------------------------- engines
from tests.test_gpu_engine import Recorder, _alt, _extra_ckpt, ckpt_dir, replay  # noqa: E402,F401


def test_hnd_selfspec_snapkv_engine_lockstep_with_oracle(ckpt_dir):
    """The SnapKV self-speculation engine with kv_layout="HND" (prefill, select, draft steps, verify writing both
    caches, rollback) replayed in lock-step against the oracle: same measured gates as the NHD engine; and the HND
    cache it ends with is the NHD engine's cache permuted, bit for bit."""
    from pathlib import Path
    from magicdec_amd.Engine.SnapKV.backend import LMBackend
    cfg, sd = gc.tiny("tinytgt")
    log = []
    eng = Recorder(mr.RefEngine("snapkv_self", cfg, sd, gc.B, gc.MAX_LEN, gc.BUDGET), "T", log)
    for ids in gc.synthetic_batches()[:1]:
        hr.selfspec_batch(eng, ids, gc.GAMMA, gc.MAX_LEN, gc.EOT_1, gc.EOT_2, False)
    caches = {}
    for layout in ("NHD", "HND"):
        e = LMBackend(dtype=torch.bfloat16, device=DEV, dec_len=gc.GAMMA + 1, draft_dec_len=1)
        e.load_model(Path(ckpt_dir) / "tinytgt" / "model.pth", use_tp=False)
        e.setup_caches(max_batch_size=gc.B, max_seq_length=gc.MAX_LEN, draft_budget=gc.BUDGET, kv_layout=layout)
        st = replay(log, {"T": e}, {"T": _alt("snapkv_self", cfg, sd, gc.B, gc.MAX_LEN, gc.BUDGET)})
        parity_report(st.line(f"selfspec/snapkv_self kv_layout={layout}"))
        caches[layout] = [b.attention.kv_cache.kv_cache.cpu() for b in e.model.layers]
        assert caches[layout][0].shape[2] == (cfg.n_local_heads if layout == "HND" else 128)
    for a, b in zip(caches["NHD"], caches["HND"]):
        assert torch.equal(bits(to_hnd(a)), bits(b))


def test_hnd_selfspec_streaming_engine_lockstep_with_oracle(ckpt_dir):
    """BASELINE.json configs[1]'s engine (StreamingLLM self-speculation: one model, the full-context cache verified
    against + the sink/window ring it drafts from) with the full-context cache stored HND (the ring stays NHD): the
    oracle's selfspec run replayed in lock-step, same measured gates; the HND cache it ends with is the NHD engine's
    cache permuted, bit for bit, and the rings are equal."""
    from pathlib import Path
    from magicdec_amd.Engine.StreamingLLM.backend import LMBackend
    cfg, sd = gc.tiny("tinytgt")
    log = []
    eng = Recorder(mr.RefEngine("stream_self", cfg, sd, gc.B, gc.MAX_LEN, gc.BUDGET), "T", log)
    for ids in gc.synthetic_batches()[:1]:
        hr.selfspec_batch(eng, ids, gc.GAMMA, gc.MAX_LEN, gc.EOT_1, gc.EOT_2, True)
    caches, rings = {}, {}
    for layout in ("NHD", "HND"):
        e = LMBackend(dtype=torch.bfloat16, device=DEV, dec_len=gc.GAMMA + 1)
        e.load_model(Path(ckpt_dir) / "tinytgt" / "model.pth", use_tp=False)
        e.setup_caches(max_batch_size=gc.B, max_seq_length=gc.MAX_LEN, draft_budget=gc.BUDGET, kv_layout=layout)
        st = replay(log, {"T": e}, {"T": _alt("stream_self", cfg, sd, gc.B, gc.MAX_LEN, gc.BUDGET)})
        parity_report(st.line(f"selfspec/stream_self kv_layout={layout}"))
        caches[layout] = [b.attention.kv_cache.kv_cache.cpu() for b in e.model.layers]
        rings[layout] = [b.attention.kv_cache.draft_cache.cpu() for b in e.model.layers]
        assert caches[layout][0].shape[2] == (cfg.n_local_heads if layout == "HND" else 128)
    for a, b in zip(caches["NHD"], caches["HND"]):
        assert torch.equal(bits(to_hnd(a)), bits(b))
    for a, b in zip(rings["NHD"], rings["HND"]):
        assert torch.equal(bits(a), bits(b))


def test_hnd_cfg5_layout_fp8_engine_lockstep_with_oracle():
    """BASELINE.json configs[4] in miniature with the fp8 cache stored HND (what the layout exists for): same replay
    against mr.RefEngine(kv_fp8=True) as test_gpu_engine.py's cfg5 test."""
    from magicdec_amd.Engine.SnapKV.backend import LMBackend
    cfg, sd, ck = _extra_ckpt("tinyqwen")
    log = []
    eng = Recorder(mr.RefEngine("snapkv_self", cfg, sd, gc.B, gc.MAX_LEN, gc.BUDGET, kv_fp8=True), "T", log)
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(4, cfg.vocab_size, (gc.B, gc.S), generator=g)
    hr.selfspec_batch(eng, ids, gc.GAMMA, gc.MAX_LEN, gc.EOT_1, gc.EOT_2, False)
    e = LMBackend(dtype=torch.bfloat16, device=DEV, dec_len=gc.GAMMA + 1, draft_dec_len=1)
    e.load_model(ck, use_tp=False)
    e.setup_caches(max_batch_size=gc.B, max_seq_length=gc.MAX_LEN, draft_budget=gc.BUDGET, kv_dtype="fp8",
                   kv_layout="HND")
    st = replay(log, {"T": e}, {"T": _alt("snapkv_self", cfg, sd, gc.B, gc.MAX_LEN, gc.BUDGET, kv_fp8=True)})
    kc = e.model.layers[0].attention.kv_cache
    hip = kc.kv_cache.float().cpu().permute(0, 1, 3, 2, 4).contiguous()          # back to [pages, 2, 128, KH, D]
    hip[:, 0] *= kc.k_scale.cpu().view(1, 1, -1, 1)
    hip[:, 1] *= kc.v_scale.cpu().view(1, 1, -1, 1)
    frac = (hip != eng.eng.caches[0]).float().mean().item()
    parity_report(st.line("cfg5: qwen-like g=5 selfspec, fp8 KV, HND pages") +
                  f"  layer-0 cache elements != oracle: {100 * frac:.3f}%")
    assert frac <= 0.02


def test_hnd_longspec_target_with_graphs_equals_nhd(ckpt_dir):
    """The longspec target back-end (plain paged cache, graph-captured decode steps) gives the same tokens with HND
    pages as with NHD pages, through harness.run_longspec_batch with a SnapKV draft."""
    from pathlib import Path
    from magicdec_amd import harness
    from magicdec_amd.Engine.SnapKV.backend import LMBackend
    from magicdec_amd.Engine.SnapKV.backend_draft import LMBackend_Draft
    outs = {}
    for layout in ("NHD", "HND"):
        eng = LMBackend(dtype=torch.bfloat16, device=DEV, dec_len=gc.GAMMA + 1)
        eng.load_model(Path(ckpt_dir) / "tinytgt" / "model.pth", use_tp=False)
        eng.compile()
        eng.setup_caches(max_batch_size=gc.B, max_seq_length=gc.MAX_LEN, kv_layout=layout)
        drf = LMBackend_Draft(dtype=torch.bfloat16, device=DEV, draft_budget=gc.BUDGET)
        drf.load_model(Path(ckpt_dir) / "tinytgt" / "model.pth", use_tp=False)
        drf.compile()
        drf.setup_caches(max_batch_size=gc.B, max_seq_length=gc.MAX_LEN, draft_budget=gc.BUDGET)
        ids = gc.synthetic_batches()[0].to(DEV)
        st, _ = harness.run_longspec_batch(eng, drf, ids, gc.GAMMA, gc.MAX_LEN, gc.EOT_1, gc.EOT_2)
        outs[layout] = (st.output.cpu(), st.num_nodes.cpu(), eng.cachelens.cpu())
    for a, b in zip(outs["NHD"], outs["HND"]):
        assert torch.equal(a, b)
