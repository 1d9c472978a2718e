"""Pins the oracle (oracle/magicdec_ref.py, oracle/harness_ref.py) against golden vectors recorded from
the REAL reference run on CPU (oracle/gen_golden.py).  CPU only.

Bit-exact for every integer (tokens, page tables, accept state, indices up to score ties) and for bf16
bytes (SnapKV scores, StreamingLLM cache contents)."""
import numpy as np
import pytest
import torch

from oracle import flashinfer_ref as fr
from oracle import harness_ref as hr
from oracle import magicdec_ref as mr
from tests import golden_cfg as gc


# ------------------------------------------------------------------ SnapKV select
@pytest.mark.parametrize("tag", ["g4", "g5", "g8", "g4d128", "g4s3104", "g4w16"])
def test_snapkv_scores_and_indices(tag, golden_dir):
    # g4s3104: a context of three 1024-column score chunks (per-chunk softmax statistics, combined), budget 257;
    # g4w16: --window_size 16 instead of the default 32
    z = np.load(f"{golden_dir}/snapkv_select_long.npz" if tag in ("g4s3104", "g4w16") else f"{golden_dir}/snapkv_select.npz")
    g, KH, D, S, budget, B, W = [int(x) for x in z[f"{tag}_meta"]]
    q = gc.from_bits(z[f"{tag}_q"]).view(B, W, g * KH, D)
    k = gc.from_bits(z[f"{tag}_k"])
    v = gc.from_bits(z[f"{tag}_v"])
    ref_scores = gc.from_bits(z[f"{tag}_scores"])          # [B, KH, S-W]
    ref_idx = torch.from_numpy(z[f"{tag}_idx"])            # [B, KH, budget-W]  (torch.topk order)
    newk = gc.from_bits(z[f"{tag}_newk"]).view(B, budget, KH, D)
    newv = gc.from_bits(z[f"{tag}_newv"]).view(B, budget, KH, D)
    topk = budget - W
    for b in range(B):
        idx, nk, nv, sc = mr.snapkv_select(q[b], k[b], v[b], g, W, budget)
        assert torch.equal(sc.view(torch.int16), ref_scores[b].view(torch.int16)), "pooled scores must be bit-exact"
        for h in range(KH):
            s = sc[h].float()
            mine, theirs = idx[h], ref_idx[b, h]
            # same multiset of selected scores, in the same (descending) order
            assert torch.equal(s[mine], s[theirs])
            thr = s[mine][-1]
            # every position strictly above the threshold score is selected by both
            strict = set(torch.nonzero(s > thr).flatten().tolist())
            assert strict <= set(mine.tolist()) and strict <= set(theirs.tolist())
            # ours: ties by lowest index
            assert torch.equal(mine, torch.sort(s, descending=True, stable=True).indices[:topk])
            # the trailing `window` rows are the last W positions, verbatim
            assert torch.equal(nk[topk:, h].view(torch.int16), newk[b, topk:, h].view(torch.int16))
            assert torch.equal(nv[topk:, h].view(torch.int16), newv[b, topk:, h].view(torch.int16))
            # reference rows are K/V gathered at ITS indices (checks the fixture/reshape convention)
            assert torch.equal(newk[b, :topk, h].view(torch.int16), k[b][theirs, h].view(torch.int16))


# ------------------------------------------------------------------ StreamingLLM eviction
def test_streaming_prefill_cache_bytes(golden_dir):
    z = np.load(f"{golden_dir}/stream_prefill.npz")
    B, KH, D, budget, ppr = [int(x) for x in z["meta"]]
    table = fr.rope_table(1024, D, 10000.0, 1.0)

    def rope(q, k, indptr, offsets):
        return fr.apply_rope(q, k, indptr, offsets, table)
    cache = torch.zeros(B * ppr, 2, 128, KH, D, dtype=torch.bfloat16)
    for step in range(int(z["nsteps"][0])):
        ctx, n, is_last, npr, last = [int(x) for x in z[f"info{step}"]]
        k = gc.from_bits(z[f"k{step}"])
        v = gc.from_bits(z[f"v{step}"])
        tab = dict(indices=torch.cat([torch.arange(i * ppr, i * ppr + npr, dtype=torch.int32) for i in range(B)]),
                   indptr=(torch.arange(B + 1) * npr).to(torch.int32), last=torch.full((B,), last, dtype=torch.int32))
        rot = mr.streaming_prefill_kv(cache, k, v, B, ctx, n, budget, tab, rope, bool(is_last))
        assert torch.equal(cache.view(torch.int16), gc.from_bits(z[f"cache{step}"]).view(torch.int16)), step
        assert torch.equal(rot.view(torch.int16), gc.from_bits(z[f"rot{step}"]).view(torch.int16)), step


# ------------------------------------------------------------------ accept loop
VARIANTS = {"longspec": dict(dr=lambda g: g, cap=lambda g: g, dbl=True, draft="draft"),
            "selfspec_snapkv": dict(dr=lambda g: g + 1, cap=lambda g: g + 1, dbl=False, draft="engine"),
            "selfspec_stream": dict(dr=lambda g: g, cap=lambda g: g, dbl=True, draft="engine")}


@pytest.mark.parametrize("fixture", ["accept_loop.json", "accept_loop_fuzz.json"])
@pytest.mark.parametrize("variant", list(VARIANTS))
def test_accept_loop_bit_exact(variant, fixture):
    """accept_loop_fuzz: unstructured cases (per-element rejections, EOT ids sprinkled over drafts and targets, gamma up to
    6, up to 130 rows) from the same exec of the reference's loop bodies."""
    cases = gc.load_json(fixture)[variant]
    V = VARIANTS[variant]
    assert len(cases) >= 20
    seen_term = seen_double = 0
    for c in cases:
        i, o = c["inp"], c["out"]
        G, B = i["gamma"], i["B"]
        tb = torch.tensor(i["tokens_buffer"])
        tt = torch.tensor(i["target_tokens"])
        output = torch.zeros(B, i["out_cols"], dtype=torch.long)
        nn = torch.tensor(i["num_nodes"])
        cl = torch.tensor(i["cachelens"], dtype=torch.int32)
        lp = torch.tensor(i["last_page_len"], dtype=torch.int32)
        pre = "draft_" if V["draft"] == "draft" else "engine_draft_"
        dcl = torch.tensor(i[pre + "cachelens"], dtype=torch.int32)
        dlp = torch.tensor(i[pre + "last_page_len"], dtype=torch.int32)
        res = mr.accept_step(tb, tt, output, nn, cl, lp, dcl, dlp, G, V["dr"](G), V["cap"](G), i["eot_1"], i["eot_2"],
                             i["prefix"] + 80, V["dbl"])
        assert res["terminal"] == o["terminal"]
        assert res["accept_nums"].tolist() == o["accept_nums"]
        assert res["bonus"].tolist() == o["bonus"]
        assert tb.tolist() == o["tokens_buffer"]
        assert cl.tolist() == o["cachelens"] and lp.tolist() == o["last_page_len"]
        assert dcl.tolist() == o[pre + "cachelens"] and dlp.tolist() == o[pre + "last_page_len"]
        assert nn.tolist() == o["num_nodes"]
        nz = np.nonzero(output.numpy())
        assert [[int(a), int(b)] for a, b in zip(*nz)] == o["output_nz"]
        assert output.numpy()[nz].tolist() == o["output_vals"]
        assert res["next_double"] == o["next_double"]
        if o["next_double"]:
            assert res["double_buffer"].tolist() == o["double_buffer"]
            assert res["cachelens_update"].tolist() == o["cachelens_update"]
            seen_double += 1
        seen_term += int(o["terminal"])
    assert seen_term > 0 and (seen_double > 0 or not V["dbl"])


# ------------------------------------------------------------------ TP sharding
def test_tp_head_partition_and_shards():
    j = gc.load_json("tp_shapes.json")
    for H, KH, world, r, s, e in j["select"]:
        assert mr.select_kv_heads(KH, r, world) == (s, e)
    cfg, sd = gc.tiny("tinytgt")
    for rec in j["shard"]:
        ssd, local = mr.shard_state_dict(sd, cfg, rec["rank"], rec["world"])
        assert [local.n_head, local.n_local_heads, local.dim] == rec["cfg"]
        for name, shape in rec["shapes"].items():
            assert list(ssd[name].shape) == shape, name
            assert abs(float(ssd[name].float().sum()) - rec["sums"][name]) <= 1e-3 * (1 + abs(rec["sums"][name])), name


# ------------------------------------------------------------------ whole-script traces
def _check_trace(mine, theirs, rename):
    assert len(mine) == len(theirs), (len(mine), len(theirs))
    for a, b in zip(mine, theirs):
        assert rename[b["cls"] + "." + b["fn"]] == a["fn"], (a["fn"], b["cls"], b["fn"])
        for key in ("inp", "out", "cachelen_update", "cachelens", "paged_kv_last_page_len", "paged_kv_indptr",
                    "draft_cachelens", "draft_paged_kv_last_page_len", "draft_paged_kv_indptr"):
            if key in b:
                assert a.get(key) == b[key], (a["fn"], key, a.get(key), b[key])


def _engines(kind):
    cfg_t, sd_t = gc.tiny("tinytgt")
    if kind == "longspec_snapkv":
        return (mr.RefEngine("target", cfg_t, sd_t, gc.B, gc.MAX_LEN),
                mr.RefEngine("snapkv_draft", cfg_t, sd_t, gc.B, gc.MAX_LEN, gc.BUDGET))
    if kind == "longspec_snapkv_rej":
        cfg_d, sd_d = gc.tiny("tinydrf")
        return (mr.RefEngine("target", cfg_t, sd_t, gc.B, gc.MAX_LEN),
                mr.RefEngine("snapkv_draft", cfg_d, sd_d, gc.B, gc.MAX_LEN, gc.BUDGET))
    if kind == "longspec_stream":
        return (mr.RefEngine("target", cfg_t, sd_t, gc.B, gc.MAX_LEN),
                mr.RefEngine("stream_draft", cfg_t, sd_t, gc.B, 0, gc.BUDGET))
    if kind == "longspec_stream_70b":          # configs[3] in miniature: g=8 target + a different, smaller draft
        cfg_7, sd_7 = gc.tiny("tiny70b")
        cfg_d, sd_d = gc.tiny("tinydrf")
        return (mr.RefEngine("target", cfg_7, sd_7, gc.B, gc.MAX_LEN),
                mr.RefEngine("stream_draft", cfg_d, sd_d, gc.B, 0, gc.BUDGET))
    if kind == "longspec_snapkv_fullkv":       # --draft_budget -1: a different draft model decoding over its full KV
        cfg_d, sd_d = gc.tiny("tinydrf")
        return (mr.RefEngine("target", cfg_t, sd_t, gc.B, gc.MAX_LEN),
                mr.RefEngine("snapkv_draft", cfg_d, sd_d, gc.B, gc.MAX_LEN, -1))
    if kind == "longspec_snapkv_b257":         # the headline budget: 3 draft pages per request
        return (mr.RefEngine("target", cfg_t, sd_t, gc.B, gc.MAX_LEN),
                mr.RefEngine("snapkv_draft", cfg_t, sd_t, gc.B, gc.MAX_LEN, 257))
    if kind == "longspec_stream_noevict":      # budget 513 > prompt + generated tokens: never evicts
        return (mr.RefEngine("target", cfg_t, sd_t, gc.B, gc.MAX_LEN),
                mr.RefEngine("stream_draft", cfg_t, sd_t, gc.B, 0, 513))
    if kind == "longspec_snapkv_eot":
        return (mr.RefEngine("target", cfg_t, sd_t, gc.B, gc.MAX_LEN),
                mr.RefEngine("snapkv_draft", cfg_t, sd_t, gc.B, gc.MAX_LEN, gc.BUDGET))
    raise KeyError(kind)


@pytest.mark.parametrize("kind,gamma", [("longspec_snapkv", 3), ("longspec_snapkv_rej", 1), ("longspec_stream", 3),
                                        ("longspec_stream_70b", 3), ("longspec_snapkv_fullkv", 3),
                                        ("longspec_snapkv_b257", 3), ("longspec_stream_noevict", 3),
                                        ("longspec_snapkv_eot", 3)])
def test_longspec_matches_reference_script(kind, gamma):
    j = gc.load_json(f"run_{kind}.json")
    eng, drf = _engines(kind)
    eot_1, eot_2 = (866, 1410) if kind.endswith("_eot") else (gc.EOT_1, gc.EOT_2)    # _eot: ids the model emits
    rename = {"SnapKV.LMBackend.encode": "T.encode", "SnapKV.LMBackend.inference": "T.inference",
              "SnapKV.LMBackend_Draft.encode": "D.encode", "SnapKV.LMBackend_Draft.inference": "D.inference",
              "StreamingLLM.LMBackend_Draft.encode": "D.encode", "StreamingLLM.LMBackend_Draft.inference": "D.inference"}
    # torch.topk's order among EQUAL bf16 scores is implementation-defined; the fixture's own tie resolution
    # is replayed so that every other integer of the run can be compared bit-exactly
    if "snapkv" in kind:
        drf.topk_replay = j["snapkv_topk"]
    trace = []
    last = None
    for ids in gc.synthetic_batches():
        last = hr.longspec_batch(eng, drf, ids, gamma, gc.MAX_LEN, eot_1, eot_2, trace)
    _check_trace(trace, j["trace"], rename)
    assert last["output"].tolist() == j["final"]["output"]
    assert last["num_nodes"].tolist() == j["final"]["num_nodes"]


@pytest.mark.parametrize("kind", ["selfspec_snapkv", "selfspec_stream", "selfspec_snapkv_qwen", "selfspec_snapkv_70b",
                                  "selfspec_stream_eot", "selfspec_stream_b257", "selfspec_snapkv_b257",
                                  "selfspec_stream_g5", "selfspec_snapkv_g5"])
def test_selfspec_matches_reference_script(kind):
    """_qwen / _70b: the reference run on a Qwen2.5-like (qkv bias, g=5, eps 1e-6) and a Llama-70B-like (g=8, D=128)
    tiny model -- pins the oracle's bias handling and the g != 4 SnapKV paths at the engine level."""
    j = gc.load_json(f"run_{kind}.json")
    cfg, sd = gc.tiny("tinyqwen" if kind.endswith("qwen") else "tiny70b" if kind.endswith("70b") else "tinytgt")
    streaming = kind.startswith("selfspec_stream")
    eot_1, eot_2 = (866, 1410) if kind.endswith("_eot") else (gc.EOT_1, gc.EOT_2)
    gamma = int(j["argv"][j["argv"].index("--gamma") + 1])            # 3, or the scripts' default 5 (_g5)
    budget = int(j["argv"][j["argv"].index("--draft_budget") + 1])    # 129, or the BASELINE budget 257 (_b257)
    eng = mr.RefEngine("stream_self" if streaming else "snapkv_self", cfg, sd, gc.B, gc.MAX_LEN, budget)
    mod = "StreamingLLM" if streaming else "SnapKV"
    rename = {f"{mod}.LMBackend.{f}": f"T.{f}" for f in ("encode", "draft_encode", "speculate", "verify")}
    if not streaming:
        eng.topk_replay = j["snapkv_topk"]
    trace = []
    last = None
    for ids in gc.synthetic_batches():
        last = hr.selfspec_batch(eng, ids, gamma, gc.MAX_LEN, eot_1, eot_2, streaming, trace)
    _check_trace(trace, j["trace"], rename)
    assert last["output"].tolist() == j["final"]["output"]
    assert last["num_nodes"].tolist() == j["final"]["num_nodes"]


@pytest.mark.parametrize("kind", ["baseline", "baseline_eot", "baseline_68m_b1"])
def test_baseline_matches_reference_script(kind):
    """baseline_68m_b1 = BASELINE.json configs[0]: the reference's "68m" entry (MHA), B = 1, prefix 129, max_len 256."""
    j = gc.load_json(f"run_{kind}.json")
    eot_1, eot_2 = (866, 1410) if kind.endswith("_eot") else (gc.EOT_1, gc.EOT_2)
    if kind == "baseline_68m_b1":
        from oracle.magicdec_ref import RefConfig, init_state_dict
        cfg = RefConfig(n_layer=2, n_head=12, n_local_heads=12, dim=768, intermediate_size=3072, vocab_size=32000)
        sd = init_state_dict(cfg, 68, wo_scale=0.1)
        B, max_len = 1, 256
        g = torch.Generator().manual_seed(123)
        ids_all = torch.randint(4, 32000, (7, 129), generator=g)
        ids_all[:, 0] = 1
        batches = [ids_all[i:i + 1] for i in range(7)]
    else:
        cfg, sd = gc.tiny("tinytgt")
        B, max_len, batches = gc.B, gc.MAX_LEN, gc.synthetic_batches()
    eng = mr.RefEngine("target", cfg, sd, B, max_len)
    rename = {"SnapKV.LMBackend.encode": "T.encode", "SnapKV.LMBackend.inference": "T.inference"}
    trace = []
    last = None
    for ids in batches:
        last = hr.baseline_batch(eng, ids, max_len, eot_1, eot_2, trace)
    _check_trace(trace, j["trace"], rename)
    assert last["output"].tolist() == j["final"]["output"]


@pytest.mark.parametrize("kind", ["longspec_snapkv_b1", "selfspec_stream_b1"])
def test_batch_size_one_matches_reference_script(kind):
    """B = 1 (the scripts' default): the oracle's loops against the reference's traces."""
    j = gc.load_json(f"run_{kind}.json")
    gamma = int(j["argv"][j["argv"].index("--gamma") + 1])
    cfg, sd = gc.tiny("tinytgt")
    g = torch.Generator().manual_seed(123)
    ids_all = torch.randint(4, 2048, (8, gc.S), generator=g)
    ids_all[:, 0] = 1
    trace = []
    last = None
    if kind.startswith("longspec"):
        eng = mr.RefEngine("target", cfg, sd, 1, gc.MAX_LEN)
        drf = mr.RefEngine("snapkv_draft", cfg, sd, 1, gc.MAX_LEN, gc.BUDGET)
        drf.topk_replay = j["snapkv_topk"]
        rename = {"SnapKV.LMBackend.encode": "T.encode", "SnapKV.LMBackend.inference": "T.inference",
                  "SnapKV.LMBackend_Draft.encode": "D.encode", "SnapKV.LMBackend_Draft.inference": "D.inference"}
        for b in range(8):
            last = hr.longspec_batch(eng, drf, ids_all[b:b + 1], gamma, gc.MAX_LEN, gc.EOT_1, gc.EOT_2, trace)
    else:
        eng = mr.RefEngine("stream_self", cfg, sd, 1, gc.MAX_LEN, gc.BUDGET)
        rename = {f"StreamingLLM.LMBackend.{f}": f"T.{f}" for f in ("encode", "draft_encode", "speculate", "verify")}
        for b in range(8):
            last = hr.selfspec_batch(eng, ids_all[b:b + 1], gamma, gc.MAX_LEN, gc.EOT_1, gc.EOT_2, True, trace)
    _check_trace(trace, j["trace"], rename)
    assert last["output"].tolist() == j["final"]["output"]
    assert last["num_nodes"].tolist() == j["final"]["num_nodes"]


def test_streaming_prefill_cache_bytes_budget_513():
    """KVCache.prefill at configs[3]'s draft budget (513 = 5 pages per request, two kv heads; eviction across several
    pages): after every chunk the oracle's cache and rotated cache hash to the REAL reference's digests
    (oracle/gen_golden.py stream_prefill_b513; inputs regenerated from the fixture's seed)."""
    import hashlib
    j = gc.load_json("stream_prefill_b513.json")
    m = j["meta"]
    B, KH, D, budget, ppr = m["B"], m["KH"], m["D"], m["budget"], m["ppr"]
    table = fr.rope_table(m["rope_positions"], D, 10000.0, 1.0)

    def rope(q, k, indptr, offsets):
        return fr.apply_rope(q, k, indptr, offsets, table)
    sha = lambda t: hashlib.sha256(t.contiguous().view(torch.int16).numpy().tobytes()).hexdigest()
    g = torch.Generator().manual_seed(m["seed"])
    cache = torch.zeros(B * ppr, 2, 128, KH, D, dtype=torch.bfloat16)
    evicted = 0
    for i, st in enumerate(j["steps"]):
        n, npr = st["n"], st["npr"]
        k = torch.randn(B * n, KH, D, generator=g).to(torch.bfloat16)
        v = torch.randn(B * n, KH, D, generator=g).to(torch.bfloat16)
        tab = dict(indices=torch.cat([torch.arange(b * ppr, b * ppr + npr, dtype=torch.int32) for b in range(B)]),
                   indptr=(torch.arange(B + 1) * npr).to(torch.int32), last=torch.full((B,), st["last"], dtype=torch.int32))
        rot = mr.streaming_prefill_kv(cache, k, v, B, st["ctx"], n, budget, tab, rope, bool(st["is_last"]))
        assert sha(cache) == st["cache_sha256"], i
        assert sha(rot) == st["rot_sha256"], i
        evicted += int(st["ctx"] + n > budget)
    assert evicted >= 4 and j["steps"][-1]["is_last"] == 1


def test_tp_shards_four_kv_heads_over_2_3_4_ranks():
    """Per-rank shapes and sums of the reference's apply_tp on the four-kv-head model, incl. the uneven three-rank split
    (2 | 1 | 1 kv heads, 342 | 342 | 340 ffn rows, 683 | 683 | 682 vocab rows): the oracle's shard_state_dict agrees."""
    j = gc.load_json("tp_shapes_kh4.json")
    cfg, sd = gc.tiny("tinykh4")
    assert {(r["world"], r["rank"]) for r in j["shard"]} == {(w, r) for w in (2, 3, 4) for r in range(w)}
    for rec in j["shard"]:
        ssd, local = mr.shard_state_dict(sd, cfg, rec["rank"], rec["world"])
        assert [local.n_head, local.n_local_heads, local.dim] == rec["cfg"]
        for name, shape in rec["shapes"].items():
            assert list(ssd[name].shape) == shape, (rec["world"], rec["rank"], name)
            assert abs(float(ssd[name].float().sum()) - rec["sums"][name]) <= 1e-3 * (1 + abs(rec["sums"][name])), name
