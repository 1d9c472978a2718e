"""Engine lock-step on RANDOM geometry (run with -m gpu): the four back-end pairings of the reference on model shapes,
batch sizes, prefix lengths, gamma and draft budgets nobody picked by hand.

Per seed: a head geometry (head dim 64 / 128, 1 / 2 / 4 kv heads, group size 1 .. 8), a model family (Llama-3.1 RoPE
scaling, plain RoPE, Qwen-style qkv bias + eps 1e-6), widths, vocabulary, batch 1 .. 4, a prefix length (128 k + 1 .. 40 for the
StreamingLLM pairings -- the reference's page tables do not grow during decode --, 128 k + 32 for SnapKV, whose reference
takes the last chunk as the window), gamma 1 .. 5 and a draft budget are drawn; the target / draft weights are PEAKED (tests/golden_cfg.peaked_pair's
construction: head tied to the embedding through a permutation, the draft mispredicting every k-th token id), so that the
oracle's argmaxes are decided by hundreds of bf16 ulps and "accepted-token sequences identical" can be asserted literally:
the HIP back-ends replay the oracle's recorded call sequence (tests/test_gpu_engine.replay: integer state bit-exact after
every call, logits inside the measured gate) and NO token may differ; then the product's own free-running loop (hipGraph
steps, accept kernel, one host read per iteration) must end with the oracle run's output buffer, lengths and iteration
count.  MAGICDEC_FUZZ_CASES=<n> widens the sweep.
"""
import os
import random
import tempfile
from pathlib import Path

import pytest
import torch

from oracle import harness_ref as hr
from oracle import magicdec_ref as mr
from tests import golden_cfg as gc
from tests.conftest import parity_report
from tests.test_gpu_engine import DEV, Recorder, _alt, _ulp_at, replay

pytestmark = pytest.mark.gpu
N_CASES = int(os.environ.get("MAGICDEC_FUZZ_CASES", "0"))
MODES = ("longspec_snapkv", "longspec_stream", "selfspec_snapkv", "selfspec_stream")


@pytest.fixture(autouse=True)
def _capped_cpu_threads():
    from tests.parity_util import capped_threads
    with capped_threads():
        yield


def draw(seed, tp=1):
    """tp > 1: a geometry whose kv heads (and feed-forward width) divide over `tp` tensor-parallel ranks."""
    r = random.Random(8000 + seed + 100003 * (tp - 1))
    mode = MODES[seed % 4] if seed < 8 else r.choice(MODES)          # the first eight seeds cover every pairing twice
    shapes = [(D, KH, g) for D in (64, 128) for KH in (1, 2, 4) for g in (1, 2, 4, 5, 8) if 256 <= D * KH * g <= 1024
              and KH % tp == 0]
    if "snapkv" in mode:
        shapes = [s for s in shapes if 8 * s[2] >= 32]               # the reference raises for 8 g < window (model.py:415)
    D, KH, g = r.choice(shapes)
    fam = r.choice(["llama31", "plain", "qwen"])
    kw = dict(rope_base=500000.0, scaling_factor=8, high_freq_factor=4, low_freq_factor=1,
              original_max_position_embeddings=8192) if fam == "llama31" else \
        dict(rope_base=1000000.0, norm_eps=1e-6, qkv_bias=True) if fam == "qwen" else dict(rope_base=10000.0)
    dim = D * KH * g
    geo = dict(n_head=KH * g, n_local_heads=KH, dim=dim, intermediate_size=128 * tp * r.randint(3, 12 if tp == 1 else 12 // tp + 2),
               vocab_size=r.choice([1024, 2048, 3000]))
    cfg_t = mr.RefConfig(n_layer=r.choice([1, 2, 3]), **geo, **kw)
    cfg_d = mr.RefConfig(n_layer=1, **{**geo, "intermediate_size": 128 * tp * r.randint(2, 8 if tp == 1 else 8 // tp + 1)}, **kw)
    budget = r.choice([129, 129, 257])
    # the reference's page tables do not grow during decode (Engine/SnapKV/backend.py:129-159 only bumps last_page_len): the
    # prefix's last page must have room for the 80 generated tokens + gamma + 1 verify rows + the bonus, i.e. S % 128 <= 40 here
    S = 128 * r.randint((budget + 40 + 127) // 128, 4) + r.randint(1, 40)
    if "snapkv" in mode:
        # the reference's gen_draft_kv scores the LAST prefill chunk's queries as its observation window (model.py:389-403:
        # a window x window causal mask on whatever the chunk holds), so its scripts run prefixes of 128 k + window
        # (16 032 = 125 * 128 + 32); the oracle and md_snapkv_select take exactly that shape
        S = 128 * r.randint((budget + 127) // 128, 4) + 32
    gamma = r.randint(1, 5)
    if mode == "longspec_snapkv":
        # the reference's loop rolls back draft.paged_kv_last_page_len (tests/SnapKV/longspec_benchmark.py:247-256) while the
        # SnapKV draft's steps append through draft_paged_kv_last_page_len (backend_draft.py:113-173), which therefore grows by
        # one per draft step and is never rolled back: the compressed cache's last page is full after 127 draft steps.  Its
        # own runs (gamma 3) end earlier; so must these (reproduced bug for bug by the oracle and by the product)
        gamma = r.randint(1, 3)
    return dict(mode=mode, cfg_t=cfg_t, cfg_d=cfg_d, fam=fam, B=r.randint(1, 4), S=S, max_len=S + 96 + r.randint(0, 40),
                gamma=gamma, budget=budget, miss_every=r.choice([3, 4, 7]), wseed=r.randint(0, 10 ** 6))


def peaked(cfg_t, cfg_d, seed, miss_every, emb_gain=16.0, peak=12.0):
    """tests/golden_cfg.peaked_pair for arbitrary configs of one width and vocabulary."""
    sd_t = dict(mr.init_state_dict(cfg_t, seed, wo_scale=0.1))
    sd_d = dict(mr.init_state_dict(cfg_d, seed + 1, wo_scale=0.1))
    g = torch.Generator().manual_seed(seed + 2)
    V, dim = cfg_t.vocab_size, cfg_t.dim
    emb = torch.randn(V, dim, generator=g) * 0.02 * emb_gain
    perm = torch.arange(V)
    perm[4:] = 4 + torch.randperm(V - 4, generator=g)               # ids 0..3 (BOS / EOT ids of the tests) stay fixed
    c = peak / (dim * 0.02 * emb_gain)
    perm_d = perm.clone()
    miss = torch.arange(4, V, miss_every)
    perm_d[miss] = perm[torch.roll(miss, 1)]
    for sd, p in ((sd_t, perm), (sd_d, perm_d)):
        sd["tok_embeddings.weight"] = emb.to(torch.bfloat16)
        sd["output.weight"] = (emb[p] * c).to(torch.bfloat16)
    return sd_t, sd_d


def _register(tmp, name, cfg, sd):
    from magicdec_amd.Engine import model_core
    os.makedirs(os.path.join(tmp, name))
    torch.save(sd, os.path.join(tmp, name, "model.pth"))
    model_core.transformer_configs[name] = gc.config_kwargs(cfg)
    return Path(tmp) / name / "model.pth"


@pytest.mark.parametrize("seed", list(range(N_CASES or 8)))
def test_fuzz_engine_lockstep_token_identity(seed):
    c = draw(seed)
    mode, cfg_t, cfg_d, B, S, max_len, gamma, budget = (c[k] for k in ("mode", "cfg_t", "cfg_d", "B", "S", "max_len", "gamma",
                                                                      "budget"))
    sd_t, sd_d = peaked(cfg_t, cfg_d, c["wseed"], c["miss_every"])
    tmp = tempfile.mkdtemp(prefix="md_fuzz_")
    ck_t = _register(tmp, f"fuzz{seed}t", cfg_t, sd_t)
    ck_d = _register(tmp, f"fuzz{seed}d", cfg_d, sd_d)
    g = torch.Generator().manual_seed(c["wseed"] + 3)
    ids = torch.randint(4, cfg_t.vocab_size, (B, S), generator=g)
    ids[:, 0] = 1
    log = []
    if mode.startswith("longspec"):
        from magicdec_amd.Engine.SnapKV.backend import LMBackend
        dkind = "snapkv_draft" if mode.endswith("snapkv") else "stream_draft"
        dargs = (B, max_len, budget) if dkind == "snapkv_draft" else (B, 0, budget)
        tgt = Recorder(mr.RefEngine("target", cfg_t, sd_t, B, max_len), "T", log)
        drf = Recorder(mr.RefEngine(dkind, cfg_d, sd_d, *dargs), "D", log)
        res = hr.longspec_batch(tgt, drf, ids, gamma, max_len, gc.EOT_1, gc.EOT_2)
        e_t = LMBackend(dtype=torch.bfloat16, device=DEV, dec_len=gamma + 1)
        e_t.load_model(ck_t, use_tp=False)
        e_t.setup_caches(max_batch_size=B, max_seq_length=max_len)
        if dkind == "snapkv_draft":
            from magicdec_amd.Engine.SnapKV.backend_draft import LMBackend_Draft
            e_d = LMBackend_Draft(dtype=torch.bfloat16, device=DEV, draft_budget=budget)
            e_d.load_model(ck_d, use_tp=False)
            e_d.setup_caches(max_batch_size=B, max_seq_length=max_len, draft_budget=budget)
        else:
            from magicdec_amd.Engine.StreamingLLM.backend_draft import LMBackend_Draft
            e_d = LMBackend_Draft(dtype=torch.bfloat16, device=DEV)
            e_d.load_model(ck_d, use_tp=False)
            e_d.setup_caches(max_batch_size=B, draft_budget=budget)
        engines = {"T": e_t, "D": e_d}
        alt = {"T": _alt("target", cfg_t, sd_t, B, max_len), "D": _alt(dkind, cfg_d, sd_d, *dargs)}
    else:
        streaming = mode.endswith("stream")
        kind = "stream_self" if streaming else "snapkv_self"
        eng = Recorder(mr.RefEngine(kind, cfg_t, sd_t, B, max_len, budget), "T", log)
        res = hr.selfspec_batch(eng, ids, gamma, max_len, gc.EOT_1, gc.EOT_2, streaming)
        if streaming:
            from magicdec_amd.Engine.StreamingLLM.backend import LMBackend
            e = LMBackend(dtype=torch.bfloat16, device=DEV, dec_len=gamma + 1)
        else:
            from magicdec_amd.Engine.SnapKV.backend import LMBackend
            e = LMBackend(dtype=torch.bfloat16, device=DEV, dec_len=gamma + 1, draft_dec_len=1)
        e.load_model(ck_t, use_tp=False)
        e.setup_caches(max_batch_size=B, max_seq_length=max_len, draft_budget=budget)
        engines = {"T": e}
        alt = {"T": _alt(kind, cfg_t, sd_t, B, max_len, budget)}
    # the zero below is only meaningful if the oracle's own argmaxes are decided by a wide margin
    npos = wide = 0
    for rec in log:
        lg = rec["logits"].view(-1, rec["logits"].shape[-1])
        top2 = lg.topk(2, dim=-1).values
        ulp = torch.tensor([_ulp_at(float(v)) for v in top2[:, 0]])
        npos += lg.shape[0]
        wide += int(((top2[:, 0] - top2[:, 1]) >= 16 * ulp).sum())
    st = replay(log, engines, alt)
    tag = (f"fuzz-{seed} {mode} {c['fam']} L{cfg_t.n_layer}/{cfg_d.n_layer} H{cfg_t.n_head} KH{cfg_t.n_local_heads} "
           f"D{cfg_t.head_dim} ffn{cfg_t.intermediate_size} V{cfg_t.vocab_size} B{B} S{S} gamma{gamma} budget{budget}")
    parity_report(st.line(tag[:34]) + f"  | {tag}: top-2 gap >= 16 ulp on {wide}/{npos} positions, {res.get('iters')} iterations")
    assert wide >= 0.97 * npos, (tag, wide, npos)
    # replay() admits a differing token only inside the oracle's own near-tie margin (<= 2 x the logit gate, a few ulps): on
    # the positions decided by >= 16 ulps none can differ, so the count is bounded by the narrow positions -- normally zero
    assert st.nties <= npos - wide, f"{tag}: {st.nties} tokens differ from the oracle's, {npos - wide} narrow positions"
    # ... and the FREE-RUNNING product loop (magicdec_amd.harness: hipGraph steps, the fused accept / rollback kernel, one
    # host read per iteration) on the same prompts: with every argmax decided by a wide margin it must end with the oracle
    # run's output buffer, lengths and iteration count -- no teacher forcing in between
    if wide == npos:
        from magicdec_amd import harness
        for e in engines.values():
            e.compile()
        if mode.startswith("longspec"):
            hst, _ = harness.run_longspec_batch(engines["T"], engines["D"], ids.to(DEV), gamma, max_len, gc.EOT_1, gc.EOT_2)
        else:
            hst, _ = harness.run_selfspec_batch(engines["T"], ids.to(DEV), gamma, max_len, gc.EOT_1, gc.EOT_2,
                                                mode.endswith("stream"))
        assert hst.iters == res["iters"], (tag, hst.iters, res["iters"])
        assert torch.equal(hst.num_nodes.cpu(), res["num_nodes"]), tag
        assert torch.equal(hst.output.cpu(), res["output"]), f"{tag}: the free-running loop's output differs from the oracle's"


@pytest.mark.parametrize("world,seed", [(2, s) for s in range(N_CASES or 2)] + [(4, s) for s in range(min(N_CASES, 12) or 1)])
def test_fuzz_tensor_parallel_free_running_loop(world, seed):
    """The same random geometry TENSOR-PARALLEL: `world` ranks sharing the box's GPU (tests/_tp_fuzz_worker.py: HIP kernels on
    kv-head shards, weights sliced by Engine/tp.py, the per-layer all-reduces through the one-shot IPC kernel, argmax merge and
    token broadcast over gloo) run the product's free-running loop.  Every rank must end with the same replicated state, no
    peer time-outs -- and, the argmaxes being decided by wide margins, with the output buffer, lengths and iteration count of
    the TP = 1 ORACLE run (a TP run rounds its partial sums to bf16 before the all-reduce, SURVEY.md section 0.9: a
    perturbation of a few ulps, far inside the margins)."""
    import json
    import subprocess
    import sys
    c = draw(seed, tp=world)
    mode, cfg_t, cfg_d, B, S, max_len, gamma, budget = (c[k] for k in ("mode", "cfg_t", "cfg_d", "B", "S", "max_len", "gamma",
                                                                      "budget"))
    sd_t, sd_d = peaked(cfg_t, cfg_d, c["wseed"], c["miss_every"])
    g = torch.Generator().manual_seed(c["wseed"] + 3)
    ids = torch.randint(4, cfg_t.vocab_size, (B, S), generator=g)
    ids[:, 0] = 1
    log = []
    if mode.startswith("longspec"):
        dkind = "snapkv_draft" if mode.endswith("snapkv") else "stream_draft"
        dargs = (B, max_len, budget) if dkind == "snapkv_draft" else (B, 0, budget)
        tgt = Recorder(mr.RefEngine("target", cfg_t, sd_t, B, max_len), "T", log)
        drf = Recorder(mr.RefEngine(dkind, cfg_d, sd_d, *dargs), "D", log)
        res = hr.longspec_batch(tgt, drf, ids, gamma, max_len, gc.EOT_1, gc.EOT_2)
    else:
        streaming = mode.endswith("stream")
        eng = Recorder(mr.RefEngine("stream_self" if streaming else "snapkv_self", cfg_t, sd_t, B, max_len, budget), "T", log)
        res = hr.selfspec_batch(eng, ids, gamma, max_len, gc.EOT_1, gc.EOT_2, streaming)
    narrow = 0
    for rec in log:
        lg = rec["logits"].view(-1, rec["logits"].shape[-1])
        top2 = lg.topk(2, dim=-1).values
        ulp = torch.tensor([_ulp_at(float(v)) for v in top2[:, 0]])
        narrow += int(((top2[:, 0] - top2[:, 1]) < 16 * ulp).sum())
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tempfile.mkdtemp(prefix="md_tpfuzz_out_")
    port = 29900 + (os.getpid() % 300) + 7 * seed + world
    procs = []
    for r_ in range(world):
        env = dict(os.environ, LOCAL_RANK=str(r_), LOCAL_WORLD_SIZE=str(world), RANK=str(r_), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MD_OUT=out, MD_SEED=str(seed),
                   MAGICDEC_TP_SINGLE_GPU="1", MAGICDEC_ONESHOT_AR="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "tests", "_tp_fuzz_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    try:
        for p_ in procs:
            logs.append(p_.communicate(timeout=600)[0])
    finally:
        for p_ in procs:
            if p_.poll() is None:
                p_.kill()
    assert all(p_.returncode == 0 for p_ in procs), "\n".join(l[-3000:] for l in logs)
    rs = [json.load(open(os.path.join(out, f"rank{r_}.json"))) for r_ in range(world)]
    tag = (f"tp{world} fuzz-{seed} {mode} {c['fam']} H{cfg_t.n_head} KH{cfg_t.n_local_heads} D{cfg_t.head_dim} "
           f"ffn{cfg_t.intermediate_size} V{cfg_t.vocab_size} B{B} S{S} gamma{gamma} budget{budget}")
    for r_ in rs:
        assert r_["local_heads"] == [cfg_t.n_head // world, cfg_t.n_local_heads // world], tag
        assert all(r_["oneshot"]) and all(s_ == 0 for s_ in r_["ar_status"]), (tag, r_["oneshot"], r_["ar_status"])
        assert r_["output"] == rs[0]["output"] and r_["num_nodes"] == rs[0]["num_nodes"] and r_["iters"] == rs[0]["iters"], tag
    same = (rs[0]["iters"] == res["iters"] and rs[0]["num_nodes"] == res["num_nodes"].tolist()
            and rs[0]["output"] == res["output"].tolist())
    parity_report(f"[tp-fuzz] {tag}: {world} ranks agree; output / lengths / {rs[0]['iters']} iterations equal to the TP = 1 "
                  f"oracle run: {same} ({narrow} narrow positions in the oracle run)")
    assert same or narrow > 0, f"{tag}: the tensor-parallel loop's output differs from the oracle's"
