"""TEST-ONLY stand-ins for magicdec_amd.ops built from the oracle, so that the product's HOST logic (back-end state
machines, model wiring, TP sharding, decode loops) can be exercised on a machine without a GPU.

The product has no CPU path: these functions are injected by the `cpu_ops` pytest fixture via monkeypatch and
exist only under tests/.  Signatures mirror magicdec_amd/ops.py."""
from __future__ import annotations

import torch

from oracle import flashinfer_ref as fr
from oracle import magicdec_ref as mr

TOPK_REPLAY = {"table": None, "pos": 0}     # reference tie resolution replay (see tests/test_oracle_golden.py)
FP8_DTYPE = torch.float8_e4m3fn


def _append(k, v, ip, cache, indices, indptr, last, kv_scales):
    if cache.dtype == FP8_DTYPE:
        fr.append_paged_kv_cache_fp8(k, v, ip, cache, indices, indptr, last, *kv_scales)
    else:
        fr.append_paged_kv_cache(k, v, ip, cache, indices, indptr, last)


def _nhd(cache, kv_layout):
    """NHD view [pages, 2, page_size, KH, D] of a cache; for an HND cache a permuted view of the same memory (index
    writes through it land in the HND tensor)."""
    if kv_layout == "HND":
        return cache.permute(0, 1, 3, 2, 4)
    assert kv_layout == "NHD", kv_layout
    return cache


def _deq(cache, kv_scales):
    return fr.dequantize_cache_fp8(cache, *kv_scales) if cache.dtype == FP8_DTYPE else cache


class RopeTable:
    def __init__(self, max_pos, head_dim, rope_theta, rope_scale, low_freq_factor=None, high_freq_factor=None,
                 old_context_len=None, device="cpu"):
        self.table = fr.rope_table(max_pos, head_dim, rope_theta, rope_scale, low_freq_factor, high_freq_factor,
                                   old_context_len)
        self.max_pos, self.head_dim = max_pos, head_dim


class AttnWorkspace:
    def __init__(self, device="cpu"):
        self.device = device

    def get(self, n):
        return None


def update_kv(k, v, ip, cache, indices, indptr, last, n_max=None, kv_scales=None, kv_layout="NHD"):
    _append(k, v, ip, _nhd(cache, kv_layout), indices, indptr, last, kv_scales)


def rope(q, k, indptr, offsets, table, n_max=None):
    if k is None:
        rq, _ = fr.apply_rope(q, q[:, :1], indptr, offsets, table.table)
        return rq, None
    return fr.apply_rope(q, k, indptr, offsets, table.table)


def rope_append(q, k, v, indptr, offsets, table, cache, indices, iptr, last, cache2=None, indices2=None, iptr2=None,
                last2=None, n_max=None, kv_scales=None, kv_layout="NHD"):
    rq, rk = fr.apply_rope(q, k, indptr, offsets, table.table)
    _append(rk, v, indptr, _nhd(cache, kv_layout), indices, iptr, last, kv_scales)
    if cache2 is not None:
        fr.append_paged_kv_cache(rk, v, indptr, cache2, indices2, iptr2, last2)
    return rq


def paged_attention(q, cache, qo_indptr, indices, indptr, last, n_max, max_pages, workspace, causal=True,
                    sm_scale=None, out=None, kv_scales=None, kv_layout="NHD"):
    cache = _nhd(cache, kv_layout)
    return fr.batch_prefill_paged(q, _deq(cache, kv_scales), qo_indptr, indices, indptr, last, q.shape[1],
                                  cache.shape[3], q.shape[2], causal=causal, sm_scale=sm_scale)


def snapkv_select(q_win, cache, indices, indptr, ctx_len, window, budget, pool_kernel, draft_cache, dindices, dindptr,
                  dlast, workspace, return_scores=False, kv_scales=None, kv_layout="NHD"):
    cache = _deq(_nhd(cache, kv_layout), kv_scales)          # fp8: rows are gathered as bf16(byte * scale)
    B = indptr.numel() - 1
    H, KH = q_win.shape[1], cache.shape[3]
    g = H // KH
    n = q_win.shape[0] // B
    last = torch.full((B,), 1, dtype=torch.int32)
    idxs, nk, nv, scs = [], [], [], []
    for b in range(B):
        npg = int(indptr[b + 1]) - int(indptr[b])
        pages = indices[int(indptr[b]):int(indptr[b + 1])].long()
        kk = cache[pages, 0].reshape(npg * cache.shape[2], KH, -1)[:ctx_len]
        vv = cache[pages, 1].reshape(npg * cache.shape[2], KH, -1)[:ctx_len]
        ov = None
        if TOPK_REPLAY["table"] is not None:
            ov = torch.as_tensor(TOPK_REPLAY["table"][TOPK_REPLAY["pos"]])[b]
        idx, k2, v2, sc = mr.snapkv_select(q_win[b * n:(b + 1) * n], kk, vv, g, window, budget, idx=ov)
        idxs.append(idx)
        scs.append(sc)
        nk.append(k2)
        nv.append(v2)
    if TOPK_REPLAY["table"] is not None:
        TOPK_REPLAY["pos"] += 1
    ip = (torch.arange(B + 1) * budget).to(torch.int32)
    fr.append_paged_kv_cache(torch.cat(nk).to(draft_cache.dtype), torch.cat(nv).to(draft_cache.dtype), ip, draft_cache,
                             dindices, dindptr, dlast)
    if return_scores:
        return torch.stack(idxs).to(torch.int32), torch.stack(scs)
    return torch.stack(idxs).to(torch.int32)


def streaming_shift_append(k_new, v_new, cache, n_new, kv_len, sink, ppr):
    B = k_new.shape[0] // n_new
    KH, D = cache.shape[3], cache.shape[4]
    for half, new in ((0, k_new), (1, v_new)):
        flat = cache[:, half].reshape(B, -1, KH, D)
        merged = torch.cat((flat[:, sink:kv_len], new.reshape(B, n_new, KH, D)), dim=1)[:, -(kv_len - sink):]
        flat2 = flat.clone()
        flat2[:, sink:kv_len] = merged
        cache[:, half] = flat2.reshape(cache[:, half].shape)


def streaming_rotate(cache, rot, B, valid, ppr, table):
    KH, D = cache.shape[3], cache.shape[4]
    if rot is not cache:
        rot.copy_(cache)
    keys = cache[:, 0].reshape(B, -1, KH, D)[:, :valid].reshape(-1, KH, D)
    ip = (torch.arange(B + 1) * valid).to(torch.int32)
    rk = fr.apply_rope(keys, keys, ip, torch.zeros(B, dtype=torch.int32), table.table)[1].reshape(B, valid, KH, D)
    flat = rot[:, 0].reshape(B, -1, KH, D).clone()
    flat[:, :valid] = rk
    rot[:, 0] = flat.reshape(rot[:, 0].shape)


def rmsnorm(x, w, eps):
    return mr.rmsnorm(x, w, eps)


def add_rmsnorm(x, r, w, eps):
    h = x + r
    return h, mr.rmsnorm(h, w, eps)


def silu_mul(a, b):
    return torch.nn.functional.silu(a) * b


def argmax(logits, index_offset=0, return_values=False):
    vals, idx = torch.max(logits, dim=-1)
    idx = idx + index_offset
    return (vals, idx) if return_values else idx


def argmax_tp_slots(logits, tp_rank, tp_world, index_offset=0):
    vals, idx = torch.max(logits, dim=-1)                     # Engine/SnapKV/model.py:178-184
    all_v = torch.zeros((logits.shape[0], tp_world), dtype=logits.dtype)
    all_i = torch.zeros((logits.shape[0], tp_world), dtype=torch.long)
    all_v[:, tp_rank] = vals
    all_i[:, tp_rank] = idx + index_offset
    return all_v, all_i


def tp_argmax_merge(vals, idx):
    return mr.tp_argmax_merge(vals, idx)


def accept_rollback(tokens_buffer, target_tokens, output, num_nodes, cachelens, last_page_len, draft_cachelens,
                    draft_last_page_len, gamma, draft_rollback, draft_cap, eot_1, eot_2, max_nodes, accept_nums, bonus,
                    double_buffer, cachelens_update, flags):
    res = mr.accept_step(tokens_buffer, target_tokens, output, num_nodes, cachelens, last_page_len, draft_cachelens,
                         draft_last_page_len, gamma, draft_rollback, draft_cap, eot_1, eot_2, max_nodes,
                         double_buffer is not None)
    accept_nums.copy_(res["accept_nums"])
    bonus.copy_(res["bonus"])
    flags[0] = int(res["terminal"])
    flags[1] = int(res["next_double"])
    if res["next_double"]:
        double_buffer.copy_(res["double_buffer"])
        cachelens_update.copy_(res["cachelens_update"])


ALL = ["RopeTable", "AttnWorkspace", "update_kv", "rope", "rope_append", "paged_attention", "snapkv_select",
       "streaming_shift_append", "streaming_rotate", "rmsnorm", "add_rmsnorm", "silu_mul", "argmax", "argmax_tp_slots", "tp_argmax_merge",
       "accept_rollback"]


def model_linear(self, x2d, lin, swiglu_w13=None):
    """Stand-in for Transformer._linear (the GEMMs are device ops too).  The product evaluates w1 | w3 as ONE GEMM over
    the fused operand; the reference as two (Engine/SnapKV/model.py:451-455).  On the GPU that is the same arithmetic per
    output element, but torch's CPU GEMM is not column-independent for every shape (at M = 256 the [684, 1024] product
    and the two [342, 1024] products differ in ~10 elements: blocking), which would flip near-tie tokens of the
    uneven-shard TP=3 trace.  The stand-in therefore splits the fused operand like the reference."""
    if swiglu_w13 is not None:
        w, s = swiglu_w13
        inter = w.shape[0] // 2
        h1 = mr.linear(x2d, w[:inter], None, s[:inter] if s is not None else None)
        h3 = mr.linear(x2d, w[inter:], None, s[inter:] if s is not None else None)
        return silu_mul(h1, h3)
    return mr.linear(x2d, lin.weight, lin.bias, getattr(lin, "scales", None))


def install(monkeypatch=None):
    """Patch magicdec_amd.ops (and the model's GEMM dispatch) in place -- with pytest's monkeypatch when given, else
    permanently for a subprocess."""
    import sys
    from magicdec_amd import ops
    from magicdec_amd.Engine import model_core
    me = sys.modules[__name__]
    for name in ALL:
        if monkeypatch is not None:
            monkeypatch.setattr(ops, name, getattr(me, name))
        else:
            setattr(ops, name, getattr(me, name))
    if monkeypatch is not None:
        monkeypatch.setattr(model_core.Transformer, "_linear", model_linear)
    else:
        model_core.Transformer._linear = model_linear
