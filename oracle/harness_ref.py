"""CPU restatement of the reference's benchmark loops (the L5 harness of SURVEY.md section 1).

TEST INFRASTRUCTURE ONLY (see oracle/magicdec_ref.py).  Each function drives
RefEngine objects exactly as the corresponding reference script drives its
back-ends, one batch at a time, and returns the per-call trace plus the final
`output` / `num_nodes` so that tests can compare with the fixtures recorded from
the real scripts (oracle/gen_golden.py run_*).

  longspec  tests/SnapKV/longspec_benchmark.py:131-295 (== tests/StreamingLLM/longspec_benchmark.py)
  selfspec  tests/SnapKV/selfspec_benchmark.py:106-211, tests/StreamingLLM/selfspec_benchmark.py:106-238
  baseline  tests/baseline_benchmark.py:60-100
"""
from __future__ import annotations

import torch

from .magicdec_ref import accept_step


RICH = False   # tests set this to also record full inputs and the oracle's top-2 logits per position


def _rec(trace, name, eng, inp, out, cachelen_update=None):
    if trace is None:
        return
    r = dict(fn=name, inp=inp.tolist() if inp.shape[1] <= 8 else [int(inp.shape[1])],
             out=out.tolist() if out.shape[1] <= 8 else out[:, -1:].tolist())
    if RICH:
        import torch as _t
        r["inp_full"] = inp.clone()
        r["out_full"] = out.clone()
        lg = eng.model.last_logits.float()
        r["top2"] = _t.topk(lg, 2, dim=-1)
        r["logits"] = lg.clone() if lg.numel() <= (1 << 18) else None
    if cachelen_update is not None:
        r["cachelen_update"] = cachelen_update.flatten().tolist()
    for at in ("cachelens", "paged_kv_last_page_len", "paged_kv_indptr", "draft_cachelens",
               "draft_paged_kv_last_page_len", "draft_paged_kv_indptr"):
        if hasattr(eng, at) and getattr(eng, at) is not None:
            r[at] = getattr(eng, at).tolist()
    trace.append(r)


def longspec_batch(engine, draft, input_ids, gamma, max_len, eot_1, eot_2, trace=None, tag_t="T", tag_d="D"):
    """One batch of tests/SnapKV/longspec_benchmark.py:131-295 with use_tp=False."""
    B, S = input_ids.shape
    tokens_buffer = torch.zeros((B, gamma + 1), dtype=torch.long)
    output = torch.zeros(B, max_len + 1, dtype=torch.long)
    output[:, :S] = input_ids
    num_nodes = torch.zeros(B, dtype=torch.long) + S
    t = engine.encode(input_ids)
    _rec(trace, tag_t + ".encode", engine, input_ids, t)
    tokens_buffer[:, :1] = t[:, -1:]
    d = draft.encode(input_ids)
    _rec(trace, tag_d + ".encode", draft, input_ids, d)
    next_double, double_buffer, cu = False, None, None
    terminal = False
    iters = 0
    while not terminal:
        for i in range(gamma):
            if i == 0 and next_double:
                nt = draft.inference(double_buffer, cachelen_update=cu)
                _rec(trace, tag_d + ".inference", draft, double_buffer, nt, cu)
                tokens_buffer[:, 1:2] = nt.gather(1, cu.view(-1, 1) - 1)
                next_double = False
            else:
                inp = tokens_buffer[:, i].view(-1, 1)
                nt = draft.inference(inp)
                _rec(trace, tag_d + ".inference", draft, inp, nt)
                tokens_buffer[:, i + 1:i + 2] = nt
        target_tokens = engine.inference(tokens_buffer)
        _rec(trace, tag_t + ".inference", engine, tokens_buffer, target_tokens)
        iters += 1
        # the harness rebinds draft.cachelens / draft.paged_kv_last_page_len (:244-256); for the compressed
        # SnapKV draft the latter is NOT the table its decode uses -- reproduced by passing those attributes
        res = accept_step(tokens_buffer, target_tokens, output, num_nodes, engine.cachelens,
                          engine.paged_kv_last_page_len, draft.cachelens, draft.paged_kv_last_page_len, gamma,
                          draft_rollback=gamma, draft_cap=gamma, eot_1=eot_1, eot_2=eot_2, max_nodes=S + 80,
                          use_double=True)
        terminal = res["terminal"]
        if res["next_double"]:
            next_double, double_buffer, cu = True, res["double_buffer"], res["cachelens_update"]
    return dict(output=output, num_nodes=num_nodes, iters=iters)


def selfspec_batch(engine, input_ids, gamma, max_len, eot_1, eot_2, streaming, trace=None, tag="T"):
    """One batch of tests/SnapKV/selfspec_benchmark.py:106-211 (streaming=False) or
    tests/StreamingLLM/selfspec_benchmark.py:106-238 (streaming=True)."""
    B, S = input_ids.shape
    tokens_buffer = torch.zeros((B, gamma + 1), dtype=torch.long)
    output = torch.zeros(B, max_len + 1, dtype=torch.long)
    output[:, :S] = input_ids
    num_nodes = torch.zeros(B, dtype=torch.long) + S
    t = engine.encode(input_ids)
    _rec(trace, tag + ".encode", engine, input_ids, t)
    tokens_buffer[:, :1] = t[:, -1:]
    if streaming:
        d = engine.draft_encode(input_ids)
        _rec(trace, tag + ".draft_encode", engine, input_ids, d)
    next_double, double_buffer, cu = False, None, None
    terminal = False
    iters = 0
    while not terminal:
        for i in range(gamma):
            if i == 0 and next_double:
                nt = engine.speculate(double_buffer, cachelen_update=cu)
                _rec(trace, tag + ".speculate", engine, double_buffer, nt, cu)
                tokens_buffer[:, 1:2] = nt.gather(1, cu.view(-1, 1) - 1)
                next_double = False
            else:
                inp = tokens_buffer[:, i].view(-1, 1)
                nt = engine.speculate(inp)
                _rec(trace, tag + ".speculate", engine, inp, nt)
                tokens_buffer[:, i + 1:i + 2] = nt
        target_tokens = engine.verify(tokens_buffer)
        _rec(trace, tag + ".verify", engine, tokens_buffer, target_tokens)
        iters += 1
        if streaming:
            res = accept_step(tokens_buffer, target_tokens, output, num_nodes, engine.cachelens,
                              engine.paged_kv_last_page_len, engine.draft_cachelens,
                              engine.draft_paged_kv_last_page_len, gamma, draft_rollback=gamma, draft_cap=gamma,
                              eot_1=eot_1, eot_2=eot_2, max_nodes=S + 80, use_double=True)
        else:
            res = accept_step(tokens_buffer, target_tokens, output, num_nodes, engine.cachelens,
                              engine.paged_kv_last_page_len, engine.draft_cachelens,
                              engine.draft_paged_kv_last_page_len, gamma, draft_rollback=gamma + 1,
                              draft_cap=gamma + 1, eot_1=eot_1, eot_2=eot_2, max_nodes=S + 80, use_double=False)
        terminal = res["terminal"]
        if res["next_double"]:
            next_double, double_buffer, cu = True, res["double_buffer"], res["cachelens_update"]
    return dict(output=output, num_nodes=num_nodes, iters=iters)


def baseline_batch(engine, input_ids, max_len, eot_1, eot_2, trace=None, tag="T"):
    """One batch of tests/baseline_benchmark.py:72-90: greedy autoregressive decode until max_len or EOT."""
    output = input_ids.clone()
    t = engine.encode(input_ids)
    _rec(trace, tag + ".encode", engine, input_ids, t)
    next_tokens = t[:, -1:]
    output = torch.cat((output, next_tokens), dim=-1)
    steps = 0
    terminate = False
    while output.size(1) < max_len and not terminate:
        inp = next_tokens.clone()
        next_tokens = engine.inference(inp)
        _rec(trace, tag + ".inference", engine, inp, next_tokens)
        output = torch.cat((output, next_tokens), dim=-1)
        steps += 1
        if bool((next_tokens[:, -1] == eot_1).any()) or bool((next_tokens[:, -1] == eot_2).any()):
            terminate = True
    return dict(output=output, steps=steps)
