"""Generate the golden fixtures under tests/golden/ by running the REAL reference on CPU.

Runs only in the build container (needs the read-only reference checkout, see
oracle/ref_import.py); the fixtures it writes are committed, this script is the
record of how they were made:

    python -m oracle.gen_golden            # all scenarios (each in its own subprocess)
    python -m oracle.gen_golden --scenario snapkv_select

A fixture is data only: seeds/inputs and the reference's outputs (token ids, page
table traces, selected indices, cache bytes as uint16 bf16 bit patterns).  The
reference registers its custom ops inside setup_caches (a second call in the same
process raises), hence one subprocess per scenario.

Scenarios
  snapkv_select     Attention.gen_draft_kv              Engine/SnapKV/model.py:389-439      g = 4, 5, 8
  stream_prefill    KVCache.prefill                     Engine/StreamingLLM/model_draft.py:102-143
  accept_loop       the verify-loop body, exec'd from   tests/SnapKV/longspec_benchmark.py:208-281
  tp_shapes         _select_kv_heads / apply_tp         Engine/tp.py:36-52,184-207
  run_*             the five benchmark scripts run unmodified under runpy with a stub tokenizer,
                    a seeded synthetic dataset and call tracing of the Engine methods
"""
from __future__ import annotations

import argparse
import json
import os
import re
import runpy
import subprocess
import sys
import tempfile
import types
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
GOLD = ROOT / "tests" / "golden"
sys.path.insert(0, str(ROOT))

from oracle import ref_import  # noqa: E402
from oracle.magicdec_ref import RefConfig, init_state_dict  # noqa: E402

BF16 = torch.bfloat16

# tiny GQA configs shared with the tests (tests/golden_cfg.py mirrors these numbers)
TINY = {
    # name: (config kwargs in the REFERENCE's ModelArgs vocabulary, weight seed)
    "tinytgt": (dict(block_size=4096, n_layer=2, n_head=8, n_local_heads=2, dim=512, intermediate_size=1024,
                     vocab_size=2048, rope_base=500000.0, scaling_factor=8, high_freq_factor=4, low_freq_factor=1,
                     original_max_position_embeddings=8192), 11, 0.1),
    "tinydrf": (dict(block_size=4096, n_layer=1, n_head=8, n_local_heads=2, dim=512, intermediate_size=1024,
                     vocab_size=2048, rope_base=10000.0), 12, 0.1),
    # other families of the reference's zoo: Qwen2.5-style (qkv bias, g = 5, eps 1e-6, theta 1e6) and
    # Llama-3.1-70B-style (g = 8, D = 128)
    "tinyqwen": (dict(block_size=4096, n_layer=2, n_head=10, n_local_heads=2, dim=640, intermediate_size=1280,
                      vocab_size=2048, rope_base=1000000.0, norm_eps=1e-6, qkv_bias=True), 21, 0.1),
    "tiny70b": (dict(block_size=4096, n_layer=2, n_head=16, n_local_heads=2, dim=2048, intermediate_size=2048,
                     vocab_size=2048, rope_base=500000.0, scaling_factor=8, high_freq_factor=4, low_freq_factor=1,
                     original_max_position_embeddings=8192), 22, 0.1),
    # four kv heads (g = 4, D = 64): shards evenly over TP4 (target) and TP2 (draft sub-group), the README's
    # "target on all ranks, draft on half of them" topology in miniature
    "tinykh4": (dict(block_size=4096, n_layer=2, n_head=16, n_local_heads=4, dim=1024, intermediate_size=1024,
                     vocab_size=2048, rope_base=500000.0, scaling_factor=8, high_freq_factor=4, low_freq_factor=1,
                     original_max_position_embeddings=8192), 24, 0.1),
}


def tp_world(tag):
    """World size of a TP scenario from its tag: ..._tp2 -> 2, ..._tp4d2 (target TP4, draft TP2) -> 4; else 0."""
    import re
    m = re.search(r"_tp(\d+)(?:d\d+)?$", tag)
    return int(m.group(1)) if m else 0


def bits(t):
    return t.contiguous().view(torch.int16).numpy().astype(np.uint16) if t.dtype == BF16 else t.numpy()


def ref_cfg(name):
    kw, seed, wo_scale = TINY[name]
    kw = {k: v for k, v in kw.items() if k != "block_size"}
    return RefConfig(n_layer=kw["n_layer"], n_head=kw["n_head"], n_local_heads=kw["n_local_heads"], dim=kw["dim"],
                     intermediate_size=kw["intermediate_size"], vocab_size=kw["vocab_size"],
                     rope_base=kw.get("rope_base", 10000.0), scaling_factor=kw.get("scaling_factor", 1.0),
                     low_freq_factor=kw.get("low_freq_factor"), high_freq_factor=kw.get("high_freq_factor"),
                     original_max_position_embeddings=kw.get("original_max_position_embeddings"),
                     norm_eps=kw.get("norm_eps", 1e-5), qkv_bias=kw.get("qkv_bias", False)), seed, wo_scale


def write_checkpoints(tmp):
    """<tmp>/<cfgname>/model.pth for every tiny config; the parent dir name selects the config
    (Engine/SnapKV/model.py:45-57)."""
    paths = {}
    for name in TINY:
        cfg, seed, wo_scale = ref_cfg(name)
        d = Path(tmp) / name
        d.mkdir(parents=True, exist_ok=True)
        torch.save(init_state_dict(cfg, seed, wo_scale=wo_scale), d / "model.pth")
        paths[name] = d / "model.pth"
    return paths


def inject_configs():
    for mod in ("Engine.SnapKV.model", "Engine.SnapKV.model_draft", "Engine.StreamingLLM.model",
                "Engine.StreamingLLM.model_draft"):
        m = ref_import.module(mod)
        for name, (kw, _, _) in TINY.items():
            m.transformer_configs[name] = dict(kw)


# ------------------------------------------------------------------------------------ snapkv_select
SNAPKV_CASES = {"g4": (4, 2, 64, 416, 129, 2, 0), "g5": (5, 2, 64, 288, 129, 2, 1), "g8": (8, 1, 128, 416, 257, 2, 2),
                "g4d128": (4, 1, 128, 544, 257, 1, 3)}
# a context of several 1024-column score chunks (the GPU kernel's softmax statistics are computed per chunk and
# combined): 3104 = 3072 + 32 columns, the headline budget 257
SNAPKV_LONG_CASES = {"g4s3104": (4, 1, 64, 3104, 257, 1, 7),
                     # --window_size 16 (the flag of the SnapKV scripts; every other fixture uses the default 32)
                     "g4w16": (4, 2, 64, 400, 129, 2, 5, 16)}


def scen_snapkv_select(cases=None, fname="snapkv_select.npz"):
    M = ref_import.module("Engine.SnapKV.model")
    out = {}
    for tag, case in (cases or SNAPKV_CASES).items():
        g, KH, D, S, budget, B, seed = case[:7]
        W = case[7] if len(case) > 7 else 32
        H = g * KH
        torch.manual_seed(seed)
        cfg = M.ModelArgs(block_size=max(2048, S), n_layer=1, n_head=H, n_local_heads=KH, dim=H * D, intermediate_size=256,
                          vocab_size=128)
        att = M.Attention(cfg)
        att.is_spec, att.draft_budget, att.window_size, att.pooling, att.kernel_size = True, budget, W, "avgpool", 5
        npg = (S + 127) // 128
        kfull = torch.randn(B, S, KH, D).to(BF16)
        vfull = torch.randn(B, S, KH, D).to(BF16)
        cache = torch.zeros(B * npg, 2, 128, KH, D, dtype=BF16)
        for b in range(B):
            kk = torch.zeros(npg * 128, KH, D, dtype=BF16)
            vv = torch.zeros(npg * 128, KH, D, dtype=BF16)
            kk[:S], vv[:S] = kfull[b], vfull[b]
            cache[b * npg:(b + 1) * npg, 0] = kk.view(npg, 128, KH, D)
            cache[b * npg:(b + 1) * npg, 1] = vv.view(npg, 128, KH, D)
        q = torch.randn(B * W, H, D).to(BF16)
        cap = {}

        class KC:
            def update_draft(self, k, v, indptr, idx, ip, last):
                cap["k"], cap["v"] = k.clone(), v.clone()
        att.kv_cache = KC()
        orig_topk = torch.Tensor.topk

        def tk(self, k, dim=-1):
            cap["scores"] = self.clone()
            r = orig_topk(self, k, dim=dim)
            cap["idx"] = r.indices.clone()
            return r
        torch.Tensor.topk = tk
        try:
            att.gen_draft_kv(q, cache[:, 0], cache[:, 1], B, W, torch.tensor(S), (torch.arange(B + 1) * budget).int(),
                             None, None, None)
        finally:
            torch.Tensor.topk = orig_topk
        out.update({f"{tag}_meta": np.array([g, KH, D, S, budget, B, W]), f"{tag}_q": bits(q), f"{tag}_k": bits(kfull),
                    f"{tag}_v": bits(vfull), f"{tag}_scores": bits(cap["scores"]), f"{tag}_idx": cap["idx"].numpy(),
                    f"{tag}_newk": bits(cap["k"]), f"{tag}_newv": bits(cap["v"])})
    np.savez_compressed(GOLD / fname, **out)


def scen_snapkv_select_long():
    scen_snapkv_select(SNAPKV_LONG_CASES, "snapkv_select_long.npz")


# ------------------------------------------------------------------------------------ stream_prefill
def scen_stream_prefill():
    ref_import.module("Engine.utils")       # defines mylib::update_kv
    M = ref_import.module("Engine.StreamingLLM.model_draft")
    fr = sys.modules["flashinfer"]
    B, KH, D, budget = 2, 1, 64, 129
    ppr = budget // 128 + 1
    table = fr.rope_table(1024, D, 10000.0, 1.0)

    def rope(q, k, indptr, offsets):
        return fr.apply_rope(q, k, indptr, offsets, table)
    kvc = M.KVCache(B * ppr, 128, KH, D, BF16, budget)
    torch.manual_seed(5)
    out = {"meta": np.array([B, KH, D, budget, ppr])}
    ctx = 0
    S = 128 * 4 + 37
    npr = 0
    step = 0
    for st in range(0, S, 128):
        n = min(128, S - st)
        is_last = n != 128
        k = torch.randn(B * n, KH, D).to(BF16)
        v = torch.randn(B * n, KH, D).to(BF16)
        if ctx + n <= budget:       # StreamingLLM/backend_draft.py:155-160
            npr += 1
            last = n
        else:
            npr = ppr
            last = budget % 128
        indices = torch.cat([torch.arange(i * ppr, i * ppr + npr, dtype=torch.int32) for i in range(B)])
        indptr = (torch.arange(B + 1) * npr).to(torch.int32)
        lastt = torch.full((B,), last, dtype=torch.int32)
        append_indptr = (torch.arange(B + 1) * n).to(torch.int32)
        rot = kvc.prefill(k, v, append_indptr, indices, indptr, lastt, B, torch.tensor(ctx), n, KH, D, rope, is_last)
        out[f"k{step}"], out[f"v{step}"] = bits(k), bits(v)
        out[f"cache{step}"] = bits(kvc.kv_cache.clone())
        out[f"rot{step}"] = bits(rot.clone())
        out[f"info{step}"] = np.array([ctx, n, int(is_last), npr, last])
        ctx = min(ctx + n, budget)
        step += 1
    out["nsteps"] = np.array([step])
    np.savez_compressed(GOLD / "stream_prefill.npz", **out)


def scen_stream_prefill_b513():
    """The same KVCache.prefill at BASELINE configs[3]'s draft budget (513 -> 5 pages per request, two kv heads): the
    eviction shifts rows across several pages.  Cache snapshots of this size do not belong in the repository: the
    fixture holds the per-chunk bookkeeping and SHA-256 digests of the cache / rotated-cache bytes; the inputs are
    regenerated from the seed (torch.Generator(7), one randn per chunk for k, one for v)."""
    import hashlib
    ref_import.module("Engine.utils")
    M = ref_import.module("Engine.StreamingLLM.model_draft")
    fr = sys.modules["flashinfer"]
    B, KH, D, budget = 2, 2, 64, 513
    ppr = budget // 128 + 1
    table = fr.rope_table(2048, D, 10000.0, 1.0)

    def rope(q, k, indptr, offsets):
        return fr.apply_rope(q, k, indptr, offsets, table)
    kvc = M.KVCache(B * ppr, 128, KH, D, BF16, budget)
    g = torch.Generator().manual_seed(7)
    sha = lambda t: hashlib.sha256(t.contiguous().view(torch.int16).numpy().tobytes()).hexdigest()
    steps = []
    ctx, npr = 0, 0
    S = 128 * 9 + 37
    for st in range(0, S, 128):
        n = min(128, S - st)
        is_last = n != 128
        k = torch.randn(B * n, KH, D, generator=g).to(BF16)
        v = torch.randn(B * n, KH, D, generator=g).to(BF16)
        if ctx + n <= budget:       # StreamingLLM/backend_draft.py:155-160
            npr += 1
            last = n
        else:
            npr = ppr
            last = budget % 128
        indices = torch.cat([torch.arange(i * ppr, i * ppr + npr, dtype=torch.int32) for i in range(B)])
        indptr = (torch.arange(B + 1) * npr).to(torch.int32)
        lastt = torch.full((B,), last, dtype=torch.int32)
        append_indptr = (torch.arange(B + 1) * n).to(torch.int32)
        rot = kvc.prefill(k, v, append_indptr, indices, indptr, lastt, B, torch.tensor(ctx), n, KH, D, rope, is_last)
        steps.append(dict(ctx=ctx, n=n, is_last=int(is_last), npr=npr, last=last, cache_sha256=sha(kvc.kv_cache),
                          rot_sha256=sha(rot)))
        ctx = min(ctx + n, budget)
    (GOLD / "stream_prefill_b513.json").write_text(json.dumps(dict(
        meta=dict(B=B, KH=KH, D=D, budget=budget, ppr=ppr, rope_positions=2048, seed=7), steps=steps)))
    print("stream_prefill_b513", len(steps), "chunks")


# ------------------------------------------------------------------------------------ accept_loop
def _loop_body_source(script, first_marker, last_marker):
    """The verify-loop body text, read from the reference at generation time (never stored in the repo)."""
    lines = (Path(ref_import.REFERENCE_ROOT) / script).read_text().splitlines()
    i0 = next(i for i, l in enumerate(lines) if first_marker in l)
    i1 = next(i for i, l in enumerate(lines) if last_marker in l and i > i0)
    body = lines[i0:i1]
    ind = min(len(l) - len(l.lstrip()) for l in body if l.strip())
    return "\n".join(l[ind:] for l in body)


def scen_accept_loop(fuzz=False):
    """Executes the reference's own loop-body statements on crafted integer inputs (EOT in draft / bonus,
    all-accept rows, mixed rows, length termination) and records inputs and resulting state."""
    variants = {
        "longspec": ("tests/SnapKV/longspec_benchmark.py", "draft_tokens = tokens_buffer[:, 1:args.gamma+1]",
                     "if not terminal:\n"),
        "selfspec_snapkv": ("tests/SnapKV/selfspec_benchmark.py", "draft_tokens = tokens_buffer[:, 1:args.gamma+1]",
                            "if not terminal:\n"),
        "selfspec_stream": ("tests/StreamingLLM/selfspec_benchmark.py",
                            "draft_tokens = tokens_buffer[:, 1:args.gamma+1]", "if not terminal:\n"),
    }
    out = {}
    rng = np.random.default_rng(11 if fuzz else 7)
    for vname, (script, m0, _) in variants.items():
        lines = (Path(ref_import.REFERENCE_ROOT) / script).read_text().splitlines()
        i0 = next(i for i, l in enumerate(lines) if m0 in l)
        # body ends right before the benchmark-timing tail: the second top-level `if not terminal:` after i0
        nt = [i for i, l in enumerate(lines) if i > i0 and l.strip() == "if not terminal:"]
        i1 = nt[1] if len(nt) > 1 else nt[0]
        body = lines[i0:i1]
        ind = min(len(l) - len(l.lstrip()) for l in body if l.strip())
        src = "\n".join(l[ind:] for l in body)
        # the else-branch that writes the bonus token on termination follows; restate its effect from the
        # script lines too: locate `for i in range(BATCH_SIZE):` after i1
        j0 = next(i for i, l in enumerate(lines) if i > i1 and "for i in range(BATCH_SIZE):" in l)
        tail = lines[j0:j0 + 3]
        tind = min(len(l) - len(l.lstrip()) for l in tail if l.strip())
        tail_src = "\n".join(l[tind:] for l in tail)
        code, tail_code = compile(src, script, "exec"), compile(tail_src, script, "exec")
        cases = []
        for case in range(20 if fuzz else 24):
            gamma = int(rng.choice([1, 2, 3, 5, 6] if fuzz else [1, 3, 4]))
            B = int(rng.choice([2, 16, 64, 130] if fuzz else [1, 3, 8]))      # 130: more than one wavefront of rows
            prefix = 300
            out_cols = prefix + 128 + 1
            eot_1, eot_2 = 7, 9
            tb = torch.from_numpy(rng.integers(10, 30, size=(B, gamma + 1))).long()
            tt = tb.roll(-1, dims=1).clone()
            tt[:, -1] = torch.from_numpy(rng.integers(10, 30, size=(B,)))
            mode = -1 if fuzz else case % 6
            if fuzz:            # unstructured: per-element rejections (p in {0.05, 0.3}), EOT ids sprinkled over drafts and
                # targets (p = 0.01: accepted EOT drafts, EOT bonus tokens, EOT after a rejection), random lengths
                p_rej = float(rng.choice([0.05, 0.3]))
                rej = torch.from_numpy(rng.random((B, gamma + 1)) < p_rej)
                tt = torch.where(rej, tt + 100, tt)
                e = torch.from_numpy(rng.random((B, gamma + 1)) < 0.01)
                which = torch.from_numpy(rng.integers(0, 2, size=(B, gamma + 1))).bool()
                eot = torch.where(which, torch.tensor(eot_1), torch.tensor(eot_2))
                tb = torch.where(e, eot, tb)
                tt[:, :-1] = torch.where(e[:, 1:] & ~rej[:, :-1], eot[:, 1:], tt[:, :-1])   # an EOT draft the target agrees on
                tt[:, -1] = torch.where(torch.from_numpy(rng.random(B) < 0.01), torch.tensor(eot_2), tt[:, -1])
            elif mode == 0:     # all accepted somewhere
                pass
            elif mode == 1:     # random rejections
                rej = torch.from_numpy(rng.integers(0, 2, size=(B, gamma + 1))).bool()
                tt = torch.where(rej, tt + 100, tt)
            elif mode == 2:     # EOT among drafts
                tb[0, min(1, gamma)] = eot_1
                tt[0, 0] = eot_1
            elif mode == 3:     # EOT as bonus
                tt[:, :] = tt + 100
                tt[B - 1, 0] = eot_2
            elif mode == 4:     # mixed full / partial
                tt[::2] = tt[::2] + 100
            base = torch.from_numpy(rng.integers(prefix, prefix + (79 if fuzz and case % 4 == 0 else 40), size=(B,))).int()
            if mode == 5:       # length termination
                base[:] = prefix + 78
            ns = types.SimpleNamespace
            engine = ns(cachelens=(base + gamma + 1).clone(), paged_kv_last_page_len=(base % 128 + gamma + 1).clone(),
                        draft_cachelens=(base + gamma + 2).clone(),
                        draft_paged_kv_last_page_len=((base + 1) % 128 + gamma + 1).clone())
            draft = ns(cachelens=(base % 129 + 129 + gamma).clone(), paged_kv_last_page_len=(base % 100 + gamma + 2).clone())
            num_nodes = (base.long() + 1).clone()
            output = torch.zeros(B, out_cols).long()
            env = dict(torch=torch, args=ns(gamma=gamma, prefix_len=prefix), tokens_buffer=tb.clone(),
                       target_tokens=tt.clone(), eot_1=eot_1, eot_2=eot_2, engine=engine, draft=draft, output=output,
                       num_nodes=num_nodes, DEVICE="cpu", BATCH_SIZE=B, terminal=False, use_tp=False, rank=0,
                       next_double=False, double_buffer=None, cachelens_update=None, benchmark=False)
            rec = dict(gamma=gamma, B=B, prefix=prefix, out_cols=out_cols, eot_1=eot_1, eot_2=eot_2,
                       tokens_buffer=tb.tolist(), target_tokens=tt.tolist(), cachelens=engine.cachelens.tolist(),
                       last_page_len=engine.paged_kv_last_page_len.tolist(),
                       engine_draft_cachelens=engine.draft_cachelens.tolist(),
                       engine_draft_last_page_len=engine.draft_paged_kv_last_page_len.tolist(),
                       draft_cachelens=draft.cachelens.tolist(), draft_last_page_len=draft.paged_kv_last_page_len.tolist(),
                       num_nodes=num_nodes.tolist())
            exec(code, env)
            if env["terminal"]:
                exec(tail_code, env)      # for i: output[i, num_nodes[i]] = bonus[i] ; num_nodes += 1
            res = dict(terminal=bool(env["terminal"]), accept_nums=env["accept_nums"].flatten().tolist(),
                       bonus=env["bonus_tokens"].flatten().tolist(), tokens_buffer=env["tokens_buffer"].tolist(),
                       cachelens=env["engine"].cachelens.tolist(), last_page_len=env["engine"].paged_kv_last_page_len.tolist(),
                       engine_draft_cachelens=env["engine"].draft_cachelens.tolist(),
                       engine_draft_last_page_len=env["engine"].draft_paged_kv_last_page_len.tolist(),
                       draft_cachelens=env["draft"].cachelens.tolist(),
                       draft_last_page_len=env["draft"].paged_kv_last_page_len.tolist(),
                       num_nodes=env["num_nodes"].tolist(), output_nz=[[int(c), int(v)] for c, v in
                                                                       zip(*np.nonzero(env["output"].numpy()))],
                       output_vals=env["output"].numpy()[np.nonzero(env["output"].numpy())].tolist(),
                       next_double=bool(env.get("next_double", False)),
                       double_buffer=env["double_buffer"].tolist() if env.get("next_double") else None,
                       cachelens_update=env["cachelens_update"].tolist() if env.get("next_double") else None)
            cases.append(dict(inp=rec, out=res))
        out[vname] = cases
    (GOLD / ("accept_loop_fuzz.json" if fuzz else "accept_loop.json")).write_text(json.dumps(out))


def scen_accept_loop_fuzz():
    scen_accept_loop(fuzz=True)


# ------------------------------------------------------------------------------------ tp_shapes
def scen_tp_shapes():
    tp = ref_import.module("Engine.tp")
    M = ref_import.module("Engine.SnapKV.model")
    res = {"select": [], "shard": []}
    for (H, KH, world) in [(32, 8, 8), (32, 8, 4), (64, 8, 8), (40, 8, 8), (28, 4, 8), (8, 2, 2), (32, 8, 3)]:
        for r in range(world):
            os.environ["LOCAL_RANK"] = str(r)
            s, e = tp._select_kv_heads(KH, list(range(world)))
            res["select"].append([H, KH, world, r, s, e])
    inject_configs()
    cfg, seed, wo_scale = ref_cfg("tinytgt")
    sd = init_state_dict(cfg, seed, wo_scale=wo_scale)
    for world in (2,):
        for r in range(world):
            os.environ["LOCAL_RANK"] = str(r)
            model = M.Transformer.from_name("tinytgt")
            model.load_state_dict(sd, assign=True)

            class G:
                pass
            import torch.distributed as dist
            # apply_tp needs dist.get_world_size/get_rank(process_group) only
            orig_ws, orig_rk = dist.get_world_size, dist.get_rank
            dist.get_world_size = lambda g=None: world
            dist.get_rank = lambda g=None: r
            try:
                tp.apply_tp(model, list(range(world)), group="G")
            finally:
                dist.get_world_size, dist.get_rank = orig_ws, orig_rk
            shapes = {k: list(v.shape) for k, v in model.state_dict().items()}
            sums = {k: float(v.float().sum()) for k, v in model.state_dict().items()}
            res["shard"].append(dict(world=world, rank=r, shapes=shapes, sums=sums,
                                     cfg=[model.config.n_head, model.config.n_local_heads, model.config.dim]))
    (GOLD / "tp_shapes.json").write_text(json.dumps(res))


MODEL_PATH_NAMES = ["Meta-Llama-3.1-8B", "Meta-Llama-3.1-8B-Instruct", "Meta-Llama-3.1-70B", "Llama-3.2-1B",
                    "Llama-3.2-3B", "Qwen2.5-7B", "Qwen2.5-14B", "Qwen2.5-32B", "llama-68m", "JackFram/llama-68m",
                    "Llama-2-7b-hf", "Llama-2-7B-32K", "Llama-2-13b-hf", "Llama-2-70b-hf", "Mistral-7B-v0.1",
                    "Yarn-Llama-2-7b-128k", "TinyLlama-1.1B", "llama-160m", "Meta-Llama-3-8B", "Meta-Llama-3-70B"]


def scen_model_configs():
    """The model zoo at the config level: every entry of the reference's transformer_configs (the four model modules
    hold the same table) resolved through ModelArgs (derived intermediate_size / head_dim / n_local_heads), and the
    fuzzy lookup ModelArgs.from_name (Engine/SnapKV/model.py:46-58) for the checkpoint directory names users pass."""
    mods = [ref_import.module(m) for m in ("Engine.SnapKV.model", "Engine.SnapKV.model_draft",
                                           "Engine.StreamingLLM.model", "Engine.StreamingLLM.model_draft")]
    assert all(m.transformer_configs == mods[0].transformer_configs for m in mods)
    M = mods[0]
    fields = ["block_size", "vocab_size", "n_layer", "n_head", "dim", "intermediate_size", "n_local_heads", "head_dim",
              "rope_base", "norm_eps", "scaling_factor", "low_freq_factor", "high_freq_factor",
              "original_max_position_embeddings", "qkv_bias"]
    res = {"table": {}, "lookup": {}}
    for name in M.transformer_configs:
        a = M.ModelArgs.from_name(name)
        res["table"][name] = {f: getattr(a, f, None) for f in fields}
    import contextlib
    import io
    for path in MODEL_PATH_NAMES:
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                a = M.ModelArgs.from_name(path)
            res["lookup"][path] = {f: getattr(a, f, None) for f in fields}
        except Exception as e:  # noqa: BLE001 -- an ambiguous / unknown name: the product must fail the same way
            res["lookup"][path] = {"error": type(e).__name__}
    (GOLD / "model_configs.json").write_text(json.dumps(res))
    print("model_configs", len(res["table"]), "entries,", len(res["lookup"]), "lookups,",
          sum(1 for v in res["lookup"].values() if "error" in v), "errors")


def scen_cli_args():
    """The command-line contract of the five entry points: every parser.add_argument(...) of the reference scripts, read
    from their source with ast (flags, type, default, nargs, action)."""
    import ast
    scripts = ["tests/SnapKV/longspec_benchmark.py", "tests/StreamingLLM/longspec_benchmark.py",
               "tests/SnapKV/selfspec_benchmark.py", "tests/StreamingLLM/selfspec_benchmark.py",
               "tests/baseline_benchmark.py"]
    res = {}
    for sc in scripts:
        tree = ast.parse((Path(ref_import.REFERENCE_ROOT) / sc).read_text())
        args = []
        for node in ast.walk(tree):
            if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "add_argument":
                kw = {k.arg: ast.unparse(k.value) for k in node.keywords if k.arg != "help"}
                args.append(dict(flags=[ast.literal_eval(a) for a in node.args], **kw))
        res[sc] = args
    (GOLD / "cli_args.json").write_text(json.dumps(res))
    print("cli_args", {k: len(v) for k, v in res.items()})


def scen_tp_shapes_kh4():
    """apply_tp (Engine/tp.py:184-207) of the four-kv-head model over 2, 3 (uneven: 2 | 1 | 1 kv heads; torch.chunk of
    1024 ffn rows -> 342 | 342 | 340, of 2048 vocab rows -> 683 | 683 | 682) and 4 ranks: per-rank tensor shapes and sums."""
    tp = ref_import.module("Engine.tp")
    M = ref_import.module("Engine.SnapKV.model")
    inject_configs()
    cfg, seed, wo_scale = ref_cfg("tinykh4")
    sd = init_state_dict(cfg, seed, wo_scale=wo_scale)
    import torch.distributed as dist
    res = {"shard": []}
    for world in (2, 3, 4):
        for r in range(world):
            os.environ["LOCAL_RANK"] = str(r)
            model = M.Transformer.from_name("tinykh4")
            model.load_state_dict(sd, assign=True)
            orig_ws, orig_rk = dist.get_world_size, dist.get_rank
            dist.get_world_size = lambda g=None: world
            dist.get_rank = lambda g=None: r
            try:
                tp.apply_tp(model, list(range(world)), group="G")
            finally:
                dist.get_world_size, dist.get_rank = orig_ws, orig_rk
            shapes = {k: list(v.shape) for k, v in model.state_dict().items()}
            sums = {k: float(v.float().sum()) for k, v in model.state_dict().items()}
            res["shard"].append(dict(world=world, rank=r, shapes=shapes, sums=sums,
                                     cfg=[model.config.n_head, model.config.n_local_heads, model.config.dim]))
    (GOLD / "tp_shapes_kh4.json").write_text(json.dumps(res))
    print("tp_shapes_kh4", len(res["shard"]), "shards")


# ------------------------------------------------------------------------------------ run_* (whole scripts)
class StubTokenizer:
    eos_token = "</s>"
    eos_token_id = 2
    unk_token_id = 0
    bos_token_id = 1
    pad_token = None

    @classmethod
    def from_pretrained(cls, *a, **k):
        return cls()

    def decode(self, ids, **k):
        return " ".join(str(int(i)) for i in ids)

    def encode(self, s, **k):
        return [3]


def synthetic_dataset(prefix_len, n_seq, vocab, seed=123):
    """PG-19-shaped batch: ids uniform in [4, vocab), column 0 = BOS (Data/data_converter.py:54)."""
    from torch.utils.data import TensorDataset
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(4, vocab, (n_seq, prefix_len), generator=g)
    ids[:, 0] = 1
    return TensorDataset(ids)


def _tp_cpu_shims():
    """The reference's init_dist (Engine/tp.py:54-64) hard-codes torch.cuda.set_device + backend "nccl"; on the CPU
    of this container the same code runs over gloo (nothing in the checkout is touched)."""
    import torch.distributed as dist
    torch.cuda.set_device = lambda *a, **k: None
    orig = dist.init_process_group

    def init_pg(backend=None, **kw):
        kw.pop("device_id", None)
        return orig(backend="gloo", **kw)
    dist.init_process_group = init_pg


def run_script(script, argv, classes, vocab, prefix_len, n_seq, tag):
    """Run a reference benchmark script unmodified under runpy, tracing Engine method calls."""
    inject_configs()
    if tp_world(tag):
        _tp_cpu_shims()
    import transformers
    transformers.AutoTokenizer = StubTokenizer
    dc = ref_import.module("Data.data_converter")
    dc.convert_pg19_dataset = lambda tokenizer=None, seq_len=0, **k: synthetic_dataset(prefix_len, n_seq, vocab)
    torch.cuda.synchronize = lambda *a, **k: None
    trace = []

    def wrap(cls, name, state_attrs):
        orig = getattr(cls, name)

        def f(self, input_ids, *a, **kw):
            r = orig(self, input_ids, *a, **kw)
            rec = dict(cls=cls.__module__.split(".")[-2] + "." + cls.__name__, fn=name,
                       inp=input_ids.tolist() if input_ids.shape[1] <= 8 else [int(input_ids.shape[1])],
                       out=r.tolist() if r.shape[1] <= 8 else r[:, -1:].tolist())
            if "cachelen_update" in kw and kw["cachelen_update"] is not None:
                rec["cachelen_update"] = kw["cachelen_update"].flatten().tolist()
            for at in state_attrs:
                if hasattr(self, at) and getattr(self, at) is not None:
                    rec[at] = getattr(self, at).tolist()
            trace.append(rec)
            return r
        setattr(cls, name, f)
    attrs = ["cachelens", "paged_kv_last_page_len", "paged_kv_indptr", "draft_cachelens", "draft_paged_kv_last_page_len",
             "draft_paged_kv_indptr"]
    for modname, clsname, fns in classes:
        cls = getattr(ref_import.module(modname), clsname)
        for fn in fns:
            if hasattr(cls, fn):
                wrap(cls, fn, attrs)
    topk_calls = []
    orig_topk = torch.Tensor.topk

    def tk(self, k, dim=-1, **kw):
        r = orig_topk(self, k, dim=dim, **kw)
        if self.dim() == 3:                      # SnapKV select: [B, KH, S-W]
            topk_calls.append(r.indices.tolist())
        return r
    torch.Tensor.topk = tk
    old = sys.argv
    sys.argv = [script] + argv
    import contextlib
    import io
    captured = io.StringIO()
    try:
        with (contextlib.redirect_stdout(captured) if tag.startswith("cli_") else contextlib.nullcontext()):
            g = runpy.run_path(str(Path(ref_import.REFERENCE_ROOT) / script), run_name="__main__")
    finally:
        sys.argv = old
        torch.Tensor.topk = orig_topk
    final = {}
    for key in ("output", "num_nodes"):
        if key in g and torch.is_tensor(g[key]):
            final[key] = g[key].tolist()
    if tag.startswith("cli_"):       # what a user SEES: the scripts' printed lines with the wall-clock numbers masked
        (GOLD / f"{tag}.json").write_text(json.dumps(dict(
            script=script, args=[a for a in argv if "magicdec_ckpt_" not in a], models=[Path(a).parent.name for a in argv
                                                                                     if "magicdec_ckpt_" in a],
            stdout=cli_lines(captured.getvalue()), snapkv_topk=topk_calls)))
        print(tag, len(cli_lines(captured.getvalue())), "lines")
        return
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    if lr == 0:          # TP runs: the replicated state is identical on every rank
        (GOLD / f"{tag}.json").write_text(json.dumps(dict(argv=argv, trace=trace, final=final,
                                                          snapkv_topk=topk_calls)))
    elif topk_calls:     # ... but each rank resolves the top-k ties of ITS kv heads: needed to replay rank r
        (GOLD / f"{tag}_topk_rank{lr}.json").write_text(json.dumps(dict(snapkv_topk=topk_calls)))
    elif lr == tp_world(tag) - 1 and re.search(r"_tp\d+d\d+$", tag):   # a rank outside the draft sub-group: target
        # calls only, draft tokens by broadcast
        (GOLD / f"{tag}_trace_rank{lr}.json").write_text(json.dumps(dict(trace=trace, final=final)))


CLI_KEEP = ("Using device", "eot_1", "Sequence", "total time", "target time", "Final tokens per second",
            "Tokens per second")


def cli_lines(text):
    """The lines of a benchmark script's stdout that are its user-visible report (device, EOT ids, decoded sequences,
    per-batch and final summaries) with the wall-clock dependent numbers replaced by <t>; counts and the
    tokens-per-sentence average stay.  tests/test_host_cpu.py applies the same function to the product's output."""
    out = []
    for ln in text.splitlines():
        ln = ln.rstrip()
        if not (ln.startswith(CLI_KEEP) or re.fullmatch(r"[0-9 ]+", ln)):
            continue
        ln = re.sub(r"(total time :|time per iter :|target time :|draft time :)[0-9.eE+-]+s", r"\1<t>s", ln)
        ln = re.sub(r"(avg latency: |verify loop : |Final tokens per second :|Tokens per second :)[0-9.eE+-]+", r"\1<t>", ln)
        out.append(ln)
    return out


def scen_run(tag):
    tmp = tempfile.mkdtemp(prefix="magicdec_ckpt_")
    ck = write_checkpoints(tmp)
    vocab = TINY["tinytgt"][0]["vocab_size"]
    B, S, ML, G = 2, 416, 512, 3
    common = ["--B", str(B), "--prefix_len", str(S), "--max_len", str(ML), "--gamma", str(G), "--rank_group", "0"]
    if tag == "run_longspec_snapkv":          # draft = target weights (high acceptance, double-buffer path)
        run_script("tests/SnapKV/longspec_benchmark.py",
                   ["--target", str(ck["tinytgt"]), "--model", str(ck["tinytgt"]), "--draft_budget", "129",
                    "--draft_rank_group", "0"] + common,
                   [("Engine.SnapKV.backend", "LMBackend", ["encode", "inference"]),
                    ("Engine.SnapKV.backend_draft", "LMBackend_Draft", ["encode", "inference"])], vocab, S, 6 * B, tag)
    elif tag == "run_longspec_snapkv_fullkv":  # --draft_budget -1 (the script's DEFAULT): the draft model decodes over its
        # full KV cache, no SnapKV select (Engine/SnapKV/backend_draft.py:15: is_compress False); different draft weights
        run_script("tests/SnapKV/longspec_benchmark.py",
                   ["--target", str(ck["tinytgt"]), "--model", str(ck["tinydrf"]), "--draft_budget", "-1",
                    "--draft_rank_group", "0"] + common,
                   [("Engine.SnapKV.backend", "LMBackend", ["encode", "inference"]),
                    ("Engine.SnapKV.backend_draft", "LMBackend_Draft", ["encode", "inference"])], vocab, S, 6 * B, tag)
    elif tag in ("run_longspec_snapkv_b1", "run_selfspec_stream_b1"):
        # B = 1, the scripts' default batch size: the loop bodies' squeeze()s yield 0-d tensors there
        # (tests/SnapKV/longspec_benchmark.py:273-279); 8 sequences so that two timed batches remain after the warm-up
        common_b1 = ["--B", "1"] + common[2:]
        if tag == "run_longspec_snapkv_b1":
            # gamma = 1: at gamma = 3 the reference's never-rolled-back SnapKV draft table overflows its spare page in the
            # sixth sequence (see run_longspec_snapkv_rej); accept_nums still reaches gamma + 1 (the two-token draft step)
            common_b1 = [c if c != str(G) else "1" for c in common_b1]
            run_script("tests/SnapKV/longspec_benchmark.py",
                       ["--target", str(ck["tinytgt"]), "--model", str(ck["tinytgt"]), "--draft_budget", "129",
                        "--draft_rank_group", "0"] + common_b1,
                       [("Engine.SnapKV.backend", "LMBackend", ["encode", "inference"]),
                        ("Engine.SnapKV.backend_draft", "LMBackend_Draft", ["encode", "inference"])], vocab, S, 8, tag)
        else:
            run_script("tests/StreamingLLM/selfspec_benchmark.py",
                       ["--model", str(ck["tinytgt"]), "--draft_budget", "129"] + common_b1,
                       [("Engine.StreamingLLM.backend", "LMBackend", ["encode", "draft_encode", "speculate", "verify"])],
                       vocab, S, 8, tag)
    elif tag.startswith("cli_"):
        # the five entry points as a user runs them: --printoutput (decoded sequences) and, where the script has it,
        # --benchmark (per-phase line); 14 sequences = 7 batches of 2 (the first 5/6 are the scripts' warm-up)
        model_args = {"cli_longspec_snapkv": ("tests/SnapKV/longspec_benchmark.py", True, "Engine.SnapKV.backend_draft"),
                      "cli_longspec_stream": ("tests/StreamingLLM/longspec_benchmark.py", True, "Engine.StreamingLLM.backend_draft"),
                      "cli_selfspec_snapkv": ("tests/SnapKV/selfspec_benchmark.py", False, None),
                      "cli_selfspec_stream": ("tests/StreamingLLM/selfspec_benchmark.py", False, None),
                      "cli_baseline": ("tests/baseline_benchmark.py", False, None)}[tag]
        script, two_models, _ = model_args
        argv = (["--target", str(ck["tinytgt"])] if two_models else []) + ["--model", str(ck["tinytgt"])]
        if tag == "cli_baseline":
            argv += ["--B", str(B), "--prefix_len", str(S), "--max_len", str(ML), "--rank_group", "0", "--printoutput"]
        else:
            argv += ["--draft_budget", "129"] + (["--draft_rank_group", "0"] if two_models else []) + common + \
                    ["--printoutput", "--benchmark"]
        run_script(script, argv, [], vocab, S, 7 * B, tag)
    elif tag == "run_longspec_snapkv_b257":   # the headline draft budget (257 rows = 3 draft pages per request)
        run_script("tests/SnapKV/longspec_benchmark.py",
                   ["--target", str(ck["tinytgt"]), "--model", str(ck["tinytgt"]), "--draft_budget", "257",
                    "--draft_rank_group", "0"] + common,
                   [("Engine.SnapKV.backend", "LMBackend", ["encode", "inference"]),
                    ("Engine.SnapKV.backend_draft", "LMBackend_Draft", ["encode", "inference"])], vocab, S, 6 * B, tag)
    elif tag == "run_longspec_stream_noevict":  # prompt shorter than the StreamingLLM budget (513 > 416 + 96): the draft
        # cache never evicts, neither in prefill nor in decode
        run_script("tests/StreamingLLM/longspec_benchmark.py",
                   ["--target", str(ck["tinytgt"]), "--model", str(ck["tinytgt"]), "--draft_budget", "513",
                    "--draft_rank_group", "0"] + common,
                   [("Engine.SnapKV.backend", "LMBackend", ["encode", "inference"]),
                    ("Engine.StreamingLLM.backend_draft", "LMBackend_Draft", ["encode", "inference"])], vocab, S, 6 * B,
                   tag)
    elif tag in ("run_longspec_snapkv_eot", "run_selfspec_stream_eot", "run_baseline_eot"):
        # EOT-driven termination: the tokenizer's eos / unk ids are set to tokens these models actually emit (866 and
        # 1410 sit in a loop of the first sequence), so batches end on an accepted EOT draft token, on an EOT bonus
        # token, or -- baseline -- on an EOT step (tests/SnapKV/longspec_benchmark.py:212-226,262-264)
        StubTokenizer.eos_token_id, StubTokenizer.unk_token_id = 866, 1410
        if tag == "run_longspec_snapkv_eot":
            run_script("tests/SnapKV/longspec_benchmark.py",
                       ["--target", str(ck["tinytgt"]), "--model", str(ck["tinytgt"]), "--draft_budget", "129",
                        "--draft_rank_group", "0"] + common,
                       [("Engine.SnapKV.backend", "LMBackend", ["encode", "inference"]),
                        ("Engine.SnapKV.backend_draft", "LMBackend_Draft", ["encode", "inference"])], vocab, S, 6 * B, tag)
        elif tag == "run_selfspec_stream_eot":
            run_script("tests/StreamingLLM/selfspec_benchmark.py",
                       ["--model", str(ck["tinytgt"]), "--draft_budget", "129"] + common,
                       [("Engine.StreamingLLM.backend", "LMBackend", ["encode", "draft_encode", "speculate", "verify"])],
                       vocab, S, 6 * B, tag)
        else:
            run_script("tests/baseline_benchmark.py",
                       ["--model", str(ck["tinytgt"]), "--B", str(B), "--prefix_len", str(S), "--max_len", str(ML),
                        "--rank_group", "0"],
                       [("Engine.SnapKV.backend", "LMBackend", ["encode", "inference"])], vocab, S, 6 * B, tag)
    elif tag == "run_baseline_int8":
        # weight-only int8 end to end: the reference's own quantiser (Engine/quantize.py:50-70) makes model_int8.pth from
        # the tiny checkpoint, the reference's loader (Engine/utils.py:201-205: "int8" in the path) swaps the linears,
        # baseline_benchmark.py decodes with it
        Q = ref_import.module("Engine.quantize")
        M = ref_import.module("Engine.SnapKV.model")
        inject_configs()
        m = M.Transformer.from_name("tinytgt")
        m.load_state_dict(torch.load(str(ck["tinytgt"]), weights_only=True), assign=True)
        q8 = Path(tmp) / "tinytgt" / "model_int8.pth"
        torch.save(Q.WeightOnlyInt8QuantHandler(m).create_quantized_state_dict(), q8)
        run_script("tests/baseline_benchmark.py",
                   ["--model", str(q8), "--B", str(B), "--prefix_len", str(S), "--max_len", str(ML), "--rank_group", "0"],
                   [("Engine.SnapKV.backend", "LMBackend", ["encode", "inference"])], vocab, S, 6 * B, tag)
    elif tag in ("run_selfspec_stream_b257", "run_selfspec_snapkv_b257", "run_selfspec_stream_g5", "run_selfspec_snapkv_g5"):
        # self-speculation at the BASELINE budgets (configs[1] / configs[4]: 257 rows = 3 draft pages) and at the
        # scripts' default speculation length gamma = 5
        kind = "StreamingLLM" if "stream" in tag else "SnapKV"
        budget = "257" if tag.endswith("b257") else "129"
        cm = [c if c != str(G) else "5" for c in common] if tag.endswith("g5") else common
        fns = ["encode", "draft_encode", "speculate", "verify"] if kind == "StreamingLLM" else ["encode", "speculate", "verify"]
        run_script(f"tests/{kind}/selfspec_benchmark.py", ["--model", str(ck["tinytgt"]), "--draft_budget", budget] + cm,
                   [(f"Engine.{kind}.backend", "LMBackend", fns)], vocab, S, 6 * B, tag)
    elif tag == "run_longspec_snapkv_rej":    # different draft model (frequent rejections); gamma=1 because the
        # SnapKV draft table is never rolled back by the harness (it rebinds draft.paged_kv_last_page_len, the
        # compressed path uses draft_paged_kv_last_page_len) and would overflow its spare page at gamma=3
        common1 = [c if c != str(G) else "1" for c in common]
        run_script("tests/SnapKV/longspec_benchmark.py",
                   ["--target", str(ck["tinytgt"]), "--model", str(ck["tinydrf"]), "--draft_budget", "129",
                    "--draft_rank_group", "0"] + common1,
                   [("Engine.SnapKV.backend", "LMBackend", ["encode", "inference"]),
                    ("Engine.SnapKV.backend_draft", "LMBackend_Draft", ["encode", "inference"])], vocab, S, 6 * B, tag)
    elif tag == "run_longspec_stream":
        run_script("tests/StreamingLLM/longspec_benchmark.py",
                   ["--target", str(ck["tinytgt"]), "--model", str(ck["tinytgt"]), "--draft_budget", "129",
                    "--draft_rank_group", "0"] + common,
                   [("Engine.SnapKV.backend", "LMBackend", ["encode", "inference"]),
                    ("Engine.StreamingLLM.backend_draft", "LMBackend_Draft", ["encode", "inference"])], vocab, S, 6 * B,
                   tag)
    elif tag == "run_longspec_stream_tp2":    # target TP2 + draft TP2 over gloo (2 processes, see main())
        common2 = [c for c in common[:-2]] + ["--rank_group", "0", "1"]
        run_script("tests/StreamingLLM/longspec_benchmark.py",
                   ["--target", str(ck["tinytgt"]), "--model", str(ck["tinytgt"]), "--draft_budget", "129",
                    "--draft_rank_group", "0", "1"] + common2,
                   [("Engine.SnapKV.backend", "LMBackend", ["encode", "inference"]),
                    ("Engine.StreamingLLM.backend_draft", "LMBackend_Draft", ["encode", "inference"])], vocab, S, 6 * B,
                   tag)
    elif tag == "run_longspec_snapkv_tp2":    # the headline layout in miniature: target TP2 + SnapKV draft TP2
        common2 = [c for c in common[:-2]] + ["--rank_group", "0", "1"]
        run_script("tests/SnapKV/longspec_benchmark.py",
                   ["--target", str(ck["tinytgt"]), "--model", str(ck["tinytgt"]), "--draft_budget", "129",
                    "--draft_rank_group", "0", "1"] + common2,
                   [("Engine.SnapKV.backend", "LMBackend", ["encode", "inference"]),
                    ("Engine.SnapKV.backend_draft", "LMBackend_Draft", ["encode", "inference"])], vocab, S, 6 * B, tag)
    elif tag == "run_longspec_snapkv_tp4d2":  # README.md:69's topology in miniature: target on all 4 ranks, SnapKV draft
        # on the sub-group {0, 1}; the gamma draft tokens reach ranks 2, 3 by broadcast (longspec_benchmark.py:189)
        common4 = [c for c in common[:-2]] + ["--rank_group", "0", "1", "2", "3"]
        run_script("tests/SnapKV/longspec_benchmark.py",
                   ["--target", str(ck["tinykh4"]), "--model", str(ck["tinykh4"]), "--draft_budget", "129",
                    "--draft_rank_group", "0", "1"] + common4,
                   [("Engine.SnapKV.backend", "LMBackend", ["encode", "inference"]),
                    ("Engine.SnapKV.backend_draft", "LMBackend_Draft", ["encode", "inference"])], vocab, S, 6 * B, tag)
    elif tag == "run_longspec_stream_70b":    # configs[3]'s layout in miniature: 70B-like target (g=8) + a different,
        # smaller draft model with the StreamingLLM cache (rejections exercise the cachelen_update / rollback paths)
        run_script("tests/StreamingLLM/longspec_benchmark.py",
                   ["--target", str(ck["tiny70b"]), "--model", str(ck["tinydrf"]), "--draft_budget", "129",
                    "--draft_rank_group", "0"] + common,
                   [("Engine.SnapKV.backend", "LMBackend", ["encode", "inference"]),
                    ("Engine.StreamingLLM.backend_draft", "LMBackend_Draft", ["encode", "inference"])], vocab, S, 6 * B,
                   tag)
    elif tag == "run_selfspec_stream_tp2":
        run_script("tests/StreamingLLM/selfspec_benchmark.py",
                   ["--model", str(ck["tinytgt"]), "--draft_budget", "129"] + common[:-2] + ["--rank_group", "0", "1"],
                   [("Engine.StreamingLLM.backend", "LMBackend", ["encode", "draft_encode", "speculate", "verify"])],
                   vocab, S, 6 * B, tag)
    elif tag == "run_selfspec_stream_tp3":    # UNEVEN kv-head shards: 4 kv heads over 3 ranks -> 2, 1, 1 (Engine/tp.py:36-52
        # gives the remainder to the lowest ranks); wqkv / wo slices and the per-rank caches differ in size
        run_script("tests/StreamingLLM/selfspec_benchmark.py",
                   ["--model", str(ck["tinykh4"]), "--draft_budget", "129"] + common[:-2] + ["--rank_group", "0", "1", "2"],
                   [("Engine.StreamingLLM.backend", "LMBackend", ["encode", "draft_encode", "speculate", "verify"])],
                   vocab, S, 6 * B, tag)
    elif tag == "run_selfspec_snapkv_tp4":    # configs[4]'s per-rank layout exactly: ONE kv head per rank (four-kv-head model
        # over 4 ranks), every rank runs the SnapKV select / gather of its own head
        run_script("tests/SnapKV/selfspec_benchmark.py",
                   ["--model", str(ck["tinykh4"]), "--draft_budget", "129"] + common[:-2] + ["--rank_group", "0", "1", "2", "3"],
                   [("Engine.SnapKV.backend", "LMBackend", ["encode", "speculate", "verify"])], vocab, S, 6 * B, tag)
    elif tag == "run_selfspec_snapkv_tp2":    # BASELINE configs[4]'s layout in miniature: TP self-speculation, SnapKV cache
        run_script("tests/SnapKV/selfspec_benchmark.py",
                   ["--model", str(ck["tinytgt"]), "--draft_budget", "129"] + common[:-2] + ["--rank_group", "0", "1"],
                   [("Engine.SnapKV.backend", "LMBackend", ["encode", "speculate", "verify"])], vocab, S, 6 * B, tag)
    elif tag in ("run_selfspec_snapkv_qwen", "run_selfspec_snapkv_70b"):
        name = "tinyqwen" if tag.endswith("qwen") else "tiny70b"
        run_script("tests/SnapKV/selfspec_benchmark.py",
                   ["--model", str(ck[name]), "--draft_budget", "129"] + common,
                   [("Engine.SnapKV.backend", "LMBackend", ["encode", "speculate", "verify"])], vocab, S, 6 * B, tag)
    elif tag == "run_selfspec_snapkv":
        run_script("tests/SnapKV/selfspec_benchmark.py",
                   ["--model", str(ck["tinytgt"]), "--draft_budget", "129"] + common,
                   [("Engine.SnapKV.backend", "LMBackend", ["encode", "speculate", "verify"])], vocab, S, 6 * B, tag)
    elif tag == "run_selfspec_stream":
        run_script("tests/StreamingLLM/selfspec_benchmark.py",
                   ["--model", str(ck["tinytgt"]), "--draft_budget", "129"] + common,
                   [("Engine.StreamingLLM.backend", "LMBackend", ["encode", "draft_encode", "speculate", "verify"])],
                   vocab, S, 6 * B, tag)
    elif tag == "run_baseline":
        run_script("tests/baseline_benchmark.py",
                   ["--model", str(ck["tinytgt"]), "--B", str(B), "--prefix_len", str(S), "--max_len", str(ML),
                    "--rank_group", "0"],
                   [("Engine.SnapKV.backend", "LMBackend", ["encode", "inference"])], vocab, S, 6 * B, tag)
    elif tag == "run_baseline_68m_b1":
        # BASELINE.json configs[0] ("llama-68m autoregressive baseline_benchmark.py on CPU, B=1 prefix=128"): the reference's
        # OWN "68m" table entry (Engine/SnapKV/model.py:67: MHA, 12 heads, dim 768, vocab 32000), seeded random weights,
        # B = 1 (0-d tensors after the script's squeeze()s), prefix_len 129 (a multiple of 128 overflows the page)
        from oracle.magicdec_ref import RefConfig as RC
        cfg68 = RC(n_layer=2, n_head=12, n_local_heads=12, dim=768, intermediate_size=3072, vocab_size=32000)
        d = Path(tmp) / "llama-68m"
        d.mkdir(parents=True, exist_ok=True)
        torch.save(init_state_dict(cfg68, 68, wo_scale=0.1), d / "model.pth")
        run_script("tests/baseline_benchmark.py",
                   ["--model", str(d / "model.pth"), "--B", "1", "--prefix_len", "129", "--max_len", "256",
                    "--rank_group", "0"],
                   [("Engine.SnapKV.backend", "LMBackend", ["encode", "inference"])], 32000, 129, 7, tag)
    else:
        raise SystemExit(f"unknown scenario {tag}")


def scen_mylib_schemas():
    """The operator schemas the reference registers (the inner boundary): the real back-ends are
    constructed on CPU, their setup_caches() runs the torch.library.define calls, the registered schemas are read
    back from the dispatcher.  Engine/utils.py:31-34, Engine/SnapKV/backend.py:56-107, backend_draft.py:42-92,
    Engine/SnapKV/model.py:134-137, model_draft.py (draft_rope)."""
    inject_configs()
    tmp = tempfile.mkdtemp(prefix="md_gold_")
    ck = write_checkpoints(tmp)
    B = ref_import.module("Engine.SnapKV.backend")
    BD = ref_import.module("Engine.SnapKV.backend_draft")
    e = B.LMBackend(dtype=BF16, device="cpu", dec_len=4)       # the longspec pairing: target + stand-alone draft
    e.load_model(ck["tinytgt"], use_tp=False)
    e.setup_caches(max_batch_size=2, max_seq_length=512)
    d = BD.LMBackend_Draft(dtype=BF16, device="cpu", dec_len=[1], draft_budget=129)
    d.load_model(ck["tinytgt"], use_tp=False)
    d.setup_caches(max_batch_size=2, max_seq_length=512, draft_budget=129)
    out = {}
    for name in ("update_kv", "rope", "draft_rope", "target_decode", "target_prefill", "draft_decode",
                 "draft_prefill"):
        out[name] = str(getattr(torch.ops.mylib, name).default._schema)
    with open(GOLD / "mylib_schemas.json", "w") as f:
        json.dump(out, f, indent=1)
    print(out)


def scen_convert_hf():
    """The REAL reference converter (convert_hf_checkpoint.py:79-163) run on seeded tiny HF-layout checkpoints
    (tests/hf_fixture.py): a sharded-safetensors Llama-style one and a single-file Qwen-style one with q/k/v biases
    (:94-99) and a tied lm head (:147-149).  Recorded: key list, shapes, dtypes and sha256 of every tensor of the
    model.pth it writes."""
    import importlib.util
    from tests import hf_fixture
    inject_configs()
    tmp = tempfile.mkdtemp(prefix="md_gold_hf_")
    ref_import.install()
    spec = importlib.util.spec_from_file_location("ref_convert_hf", os.path.join(ref_import.REFERENCE_ROOT,
                                                                               "convert_hf_checkpoint.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = {}
    for case, (name, tied, sharded) in hf_fixture.CASES.items():
        cfg, _, _ = ref_cfg(name)
        d = hf_fixture.write_hf_checkpoint(tmp, case, cfg)
        mod.convert_hf_checkpoint(checkpoint_dir=Path(d), model_name=name)
        sd = torch.load(os.path.join(d, "model.pth"), map_location="cpu", weights_only=True)
        out[case] = hf_fixture.describe(sd)
    with open(GOLD / "convert_hf.json", "w") as f:
        json.dump(out, f, indent=0)
    print({c: len(v) for c, v in out.items()})


def scen_pg19():
    """The REAL Data/data_converter.py:44-58 (convert_pg19_dataset) run on a seeded synthetic corpus with a
    deterministic stub tokenizer (tests/pg19_fixture.py), once with and once without a BOS id.  Recorded: shape,
    sha256 of the int64 tensor, its first row and per-book chunk structure (row count)."""
    import hashlib
    from tests import pg19_fixture as pf
    tmp = tempfile.mkdtemp(prefix="md_gold_pg19_")
    pf.write_corpus(tmp)
    dc = ref_import.module("Data.data_converter")
    out = {}
    cwd = os.getcwd()
    os.chdir(tmp)          # the reference reads the relative path "Data/pg19/"
    try:
        for tag, bos in (("bos", True), ("no_bos", False)):
            ds = dc.convert_pg19_dataset(tokenizer=pf.WordTokenizer(bos), seq_len=pf.SEQ_LEN, end=pf.END)
            t = ds.tensors[0]
            out[tag] = dict(shape=list(t.shape), dtype=str(t.dtype),
                            sha256=hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest(),
                            first_row=t[0].tolist(), col0=sorted(set(t[:, 0].tolist())))
    finally:
        os.chdir(cwd)
    with open(GOLD / "pg19.json", "w") as f:
        json.dump(out, f)
    print({k: v["shape"] for k, v in out.items()})


def scen_int8_quant():
    """The REAL Engine/quantize.py: dynamically_quantize_per_channel (:7-41) on a seeded weight matrix (rows with
    all-zero, all-negative and huge entries included) and WeightOnlyInt8Linear.forward (:84-86) on seeded activations."""
    import hashlib
    Q = ref_import.module("Engine.quantize")
    g = torch.Generator().manual_seed(31)
    w = torch.randn(96, 256, generator=g) * 0.05
    w[3] = 0.0
    w[5] = -w[5].abs()
    w[7, 11] = 40.0
    q, sc, zp = Q.dynamically_quantize_per_channel(w.float(), -128, 127, torch.int8)
    lin = Q.WeightOnlyInt8Linear(256, 96)
    lin.weight, lin.scales = q, sc.to(BF16)
    x = torch.randn(5, 256, generator=g).to(BF16)
    y = lin(x)
    h = lambda t: hashlib.sha256(t.contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()
    out = dict(q_sha=h(q), scales_sha=h(sc), scales_bf16_sha=h(sc.to(BF16)), y_sha=h(y), q_row7=q[7].tolist(),
               scales_head=sc[:8].tolist(), zp_all_zero=bool((zp == 0).all()), y_dtype=str(y.dtype))
    with open(GOLD / "int8_quant.json", "w") as f:
        json.dump(out, f)
    print({k: (v if not isinstance(v, list) else len(v)) for k, v in out.items()})


# ------------------------------------------------------------------------------------ benchmark=True ("run, then undo
# the length updates", Engine/SnapKV/backend.py:140-143 and twins): no reference script passes it, so the engines are
# driven directly with a fixed program of calls; tests/test_host_cpu.py runs the same program on the product
BENCHFLAG_PROGRAMS = {
    # name: (engines {key: (module, class, ctor kwargs, setup_caches kwargs)}, program [(key, fn, ncols, kwargs)])
    "benchflag_snapkv_self": (
        {"T": ("Engine.SnapKV.backend", "LMBackend", dict(dec_len=4, draft_dec_len=1), dict(max_seq_length=512, draft_budget=129))},
        [("T", "encode", 416, {}), ("T", "speculate", 1, dict(benchmark=True)), ("T", "speculate", 1, {}),
         ("T", "speculate", 1, dict(benchmark=True)), ("T", "speculate", 1, {}), ("T", "speculate", 1, {}),
         ("T", "verify", 4, dict(benchmark=True)), ("T", "verify", 4, {}), ("T", "speculate", 1, {}),
         ("T", "verify", 4, dict(benchmark=True))]),
    "benchflag_longspec_snapkv": (
        {"T": ("Engine.SnapKV.backend", "LMBackend", dict(dec_len=4), dict(max_seq_length=512)),
         "D": ("Engine.SnapKV.backend_draft", "LMBackend_Draft", dict(draft_budget=129), dict(max_seq_length=512, draft_budget=129))},
        [("T", "encode", 416, {}), ("D", "encode", 416, {}), ("D", "inference", 1, dict(benchmark=True)),
         ("D", "inference", 1, {}), ("D", "inference", 2, dict(benchmark=True, cachelen_update=[1, 2])),
         ("D", "inference", 2, dict(cachelen_update=[2, 1])), ("D", "inference", 1, {}),
         ("T", "inference", 4, dict(benchmark=True)), ("T", "inference", 4, {}), ("T", "inference", 1, dict(benchmark=True)),
         ("T", "inference", 1, {})]),
    "benchflag_longspec_stream": (
        {"D": ("Engine.StreamingLLM.backend_draft", "LMBackend_Draft", dict(), dict(draft_budget=129))},
        [("D", "encode", 416, {}), ("D", "inference", 1, dict(benchmark=True)), ("D", "inference", 1, {}),
         ("D", "inference", 2, dict(benchmark=True, cachelen_update=[1, 2])),
         ("D", "inference", 2, dict(cachelen_update=[2, 1])), ("D", "inference", 1, {})]),
    "benchflag_stream_self": (
        {"T": ("Engine.StreamingLLM.backend", "LMBackend", dict(dec_len=4), dict(max_seq_length=512, draft_budget=129))},
        [("T", "encode", 416, {}), ("T", "draft_encode", 416, {}), ("T", "speculate", 1, dict(benchmark=True)),
         ("T", "speculate", 1, {}), ("T", "speculate", 2, dict(benchmark=True, cachelen_update=[1, 2])),
         ("T", "speculate", 2, dict(cachelen_update=[2, 1])), ("T", "speculate", 1, {}),
         ("T", "verify", 4, dict(benchmark=True)), ("T", "verify", 4, {})]),
}
BENCHFLAG_ATTRS = ["cachelens", "paged_kv_last_page_len", "paged_kv_indptr", "draft_cachelens",
                   "draft_paged_kv_last_page_len", "draft_paged_kv_indptr"]


def benchflag_inputs(ncols, call_index, vocab=2048, B=2):
    """Token ids of call `call_index` of a program (seeded; the same function is used by the product-side test)."""
    g = torch.Generator().manual_seed(1000 + call_index)
    ids = torch.randint(4, vocab, (B, ncols), generator=g)
    if ncols > 8:
        ids[:, 0] = 1
    return ids


def scen_benchflag(tag):
    inject_configs()
    tmp = tempfile.mkdtemp(prefix="magicdec_ckpt_")
    ck = write_checkpoints(tmp)
    engines_spec, program = BENCHFLAG_PROGRAMS[tag]
    engines = {}
    for key, (mod, cls, ctor, caches) in engines_spec.items():
        e = getattr(ref_import.module(mod), cls)(dtype=BF16, device="cpu", **ctor)
        e.load_model(ck["tinytgt"], use_tp=False)
        e.setup_caches(max_batch_size=2, **caches)
        engines[key] = e
    trace = []
    topk_calls = []          # torch.topk's order among (near-)equal scores is implementation-defined: recorded so that
    orig_topk = torch.Tensor.topk     # a replay can put the same rows in the same order into the draft cache

    def tk(self, k, dim=-1, **kw):
        r = orig_topk(self, k, dim=dim, **kw)
        if self.dim() == 3:
            topk_calls.append(r.indices.tolist())
        return r
    torch.Tensor.topk = tk
    for i, (key, fn, ncols, kw) in enumerate(program):
        e = engines[key]
        ids = benchflag_inputs(ncols, i)
        kwargs = dict(kw)
        if "cachelen_update" in kwargs:
            kwargs["cachelen_update"] = torch.tensor(kwargs["cachelen_update"])
        r = getattr(e, fn)(ids, **kwargs)
        rec = dict(key=key, fn=fn, benchmark=bool(kw.get("benchmark", False)),
                   out=r.tolist() if r.shape[1] <= 8 else r[:, -1:].tolist())
        for at in BENCHFLAG_ATTRS:
            if hasattr(e, at) and getattr(e, at) is not None:
                rec[at] = getattr(e, at).tolist()
        trace.append(rec)
    (GOLD / f"{tag}.json").write_text(json.dumps(dict(
        engines={k: dict(module=m, cls=c, ctor=ct, caches=ca) for k, (m, c, ct, ca) in engines_spec.items()},
        program=[dict(key=k, fn=f, ncols=n, kwargs=kw) for k, f, n, kw in program],
        inputs="call i: torch.randint(4, 2048, (2, ncols), generator=manual_seed(1000 + i)); column 0 = 1 if ncols > 8",
        trace=trace, snapkv_topk=topk_calls)))
    torch.Tensor.topk = orig_topk
    print(tag, len(trace), "calls", len(topk_calls), "top-k calls")


SCENARIOS = {"int8_quant": scen_int8_quant, "pg19": scen_pg19, "convert_hf": scen_convert_hf, "mylib_schemas": scen_mylib_schemas, "snapkv_select": scen_snapkv_select, "snapkv_select_long": scen_snapkv_select_long, "stream_prefill": scen_stream_prefill, "stream_prefill_b513": scen_stream_prefill_b513,
             "accept_loop": scen_accept_loop, "accept_loop_fuzz": scen_accept_loop_fuzz, "tp_shapes": scen_tp_shapes, "tp_shapes_kh4": scen_tp_shapes_kh4,
             "model_configs": scen_model_configs, "cli_args": scen_cli_args}
RUNS = ["run_longspec_snapkv", "run_longspec_snapkv_rej", "run_longspec_stream", "run_selfspec_snapkv",
        "run_selfspec_stream", "run_baseline", "run_longspec_stream_tp2", "run_longspec_snapkv_tp2",
        "run_selfspec_snapkv_tp2", "run_selfspec_snapkv_qwen", "run_selfspec_snapkv_70b",
        "run_longspec_stream_70b", "run_selfspec_stream_tp2", "run_longspec_snapkv_tp4d2",
        "run_selfspec_stream_tp3", "run_selfspec_snapkv_tp4", "run_baseline_68m_b1",
        "run_longspec_snapkv_fullkv", "run_longspec_snapkv_b1", "run_selfspec_stream_b1", "run_longspec_snapkv_b257",
        "run_longspec_stream_noevict", "run_longspec_snapkv_eot", "run_selfspec_stream_eot",
        "run_baseline_eot", "run_baseline_int8", "run_selfspec_stream_b257",
        "run_selfspec_snapkv_b257", "run_selfspec_stream_g5", "run_selfspec_snapkv_g5", "cli_longspec_snapkv",
        "cli_longspec_stream", "cli_selfspec_snapkv", "cli_selfspec_stream", "cli_baseline"]


def _spawn_tp(scenario, world=2):
    """torchrun-style launch of `world` CPU processes of one scenario (rendezvous on 127.0.0.1)."""
    port = 29900 + os.getpid() % 90
    procs = []
    for r in range(world):
        env = dict(os.environ, LOCAL_RANK=str(r), RANK=str(r), LOCAL_WORLD_SIZE=str(world), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2", MD_GOLDEN_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, "-m", "oracle.gen_golden", "--scenario", scenario],
                                      cwd=str(ROOT), env=env))
    for p in procs:
        if p.wait() != 0:
            raise SystemExit(f"{scenario}: a rank failed")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenario", default=None)
    a = ap.parse_args()
    GOLD.mkdir(parents=True, exist_ok=True)
    if a.scenario is None:
        for s in list(SCENARIOS) + list(BENCHFLAG_PROGRAMS) + RUNS:
            print("==", s, flush=True)
            subprocess.run([sys.executable, "-m", "oracle.gen_golden", "--scenario", s], cwd=str(ROOT), check=True)
        return
    if tp_world(a.scenario) and os.environ.get("MD_GOLDEN_CHILD") != "1":
        _spawn_tp(a.scenario, tp_world(a.scenario))
        return
    torch.manual_seed(0)
    with torch.inference_mode():
        if a.scenario in BENCHFLAG_PROGRAMS:
            scen_benchflag(a.scenario)
        elif a.scenario in SCENARIOS:
            SCENARIOS[a.scenario]()
        else:
            scen_run(a.scenario)


if __name__ == "__main__":
    main()
