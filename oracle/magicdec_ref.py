"""CPU restatement ("port") of MagicDec's speculative draft/verify decode path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by magicdec_amd/.  It is the checker the HIP
path is compared with and the CPU baseline timed beside it.

Pinning: everything here that restates *reference* code (page-table state
machines, SnapKV select, StreamingLLM eviction, model wiring, TP sharding, the
accept loop) is checked against fixtures produced by running the real reference
on CPU (oracle/gen_golden.py -> tests/golden/, tests/test_oracle_golden.py).
The flashinfer boundary below it is "parity unpinned" (see oracle/flashinfer_ref.py).

Structure is deliberately different from the reference (one functional model +
one engine class parameterised by mode) -- each function cites the reference
lines it follows.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from itertools import accumulate
from typing import Optional

import torch
import torch.nn.functional as F

from . import flashinfer_ref as fr

BF16 = torch.bfloat16
PAGE = 128          # Engine/SnapKV/backend.py:31
CHUNK = 128         # Engine/SnapKV/backend.py:236
SINK = 16           # Engine/StreamingLLM/model_draft.py:124
POOL_KERNEL = 5     # Engine/SnapKV/model.py:169


# --------------------------------------------------------------------------- config
@dataclass
class RefConfig:
    """Engine/SnapKV/model.py:17-43 (ModelArgs), post-init values resolved."""
    n_layer: int
    n_head: int
    n_local_heads: int
    dim: int
    intermediate_size: int
    vocab_size: int
    rope_base: float = 10000.0
    norm_eps: float = 1e-5
    scaling_factor: float = 1.0
    low_freq_factor: Optional[float] = None
    high_freq_factor: Optional[float] = None
    original_max_position_embeddings: Optional[int] = None
    qkv_bias: bool = False

    @property
    def head_dim(self):
        return self.dim // self.n_head


def init_state_dict(cfg: RefConfig, seed: int, std: float = 0.02, wo_scale: float = 1.0):
    """Seeded normal(0, std) bf16 weights under the reference's parameter names.  wo_scale < 1 damps the
    attention branch so that a sparse-KV draft agrees with the target often enough to exercise the
    accept paths in tiny test models (random full-strength weights give ~0 acceptance)."""
    g = torch.Generator().manual_seed(seed)

    def w(*shape):
        return (torch.randn(*shape, generator=g) * std).to(BF16)

    D = cfg.head_dim
    sd = {"tok_embeddings.weight": w(cfg.vocab_size, cfg.dim), "norm.weight": torch.ones(cfg.dim, dtype=BF16),
          "output.weight": w(cfg.vocab_size, cfg.dim)}
    for i in range(cfg.n_layer):
        p = f"layers.{i}."
        sd[p + "attention.wqkv.weight"] = w((cfg.n_head + 2 * cfg.n_local_heads) * D, cfg.dim)
        if cfg.qkv_bias:
            sd[p + "attention.wqkv.bias"] = w((cfg.n_head + 2 * cfg.n_local_heads) * D)
        sd[p + "attention.wo.weight"] = (w(cfg.dim, cfg.dim).float() * wo_scale).to(BF16)
        sd[p + "feed_forward.w1.weight"] = w(cfg.intermediate_size, cfg.dim)
        sd[p + "feed_forward.w3.weight"] = w(cfg.intermediate_size, cfg.dim)
        sd[p + "feed_forward.w2.weight"] = w(cfg.dim, cfg.intermediate_size)
        sd[p + "attention_norm.weight"] = (1.0 + 0.1 * torch.randn(cfg.dim, generator=g)).to(BF16)
        sd[p + "ffn_norm.weight"] = (1.0 + 0.1 * torch.randn(cfg.dim, generator=g)).to(BF16)
    return sd


# --------------------------------------------------------------------------- TP sharding
def select_kv_heads(num_kv_heads: int, rank: int, world: int):
    """Engine/tp.py:36-52: remainder heads go to the lowest ranks."""
    base, rem = divmod(num_kv_heads, world)
    dist_ = [base + (1 if i < rem else 0) for i in range(world)]
    cum = list(accumulate(dist_))
    start = 0 if rank == 0 else cum[rank - 1]
    return start, cum[rank]


def shard_state_dict(sd, cfg: RefConfig, rank: int, world: int):
    """Engine/tp.py:67-207: wqkv by kv-head range (q rows of the matching groups), wo columns by the same q
    range, w1/w3 row-chunked, w2 column-chunked, lm head vocab-chunked; embeddings and norms replicated.
    Returns (sharded state dict, local config)."""
    D = cfg.head_dim
    g = cfg.n_head // cfg.n_local_heads
    s, e = select_kv_heads(cfg.n_local_heads, rank, world)
    qs, qe = s * g * D, e * g * D
    ks, ke = s * D, e * D
    kv_size = cfg.n_local_heads * D
    out = {}
    for name, t in sd.items():
        if name.endswith("attention.wqkv.weight") or name.endswith("attention.wqkv.bias"):
            q, k, v = t.split([cfg.dim, kv_size, kv_size], dim=0)
            out[name] = torch.cat([q[qs:qe], k[ks:ke], v[ks:ke]], dim=0)
        elif name.endswith("attention.wo.weight"):
            out[name] = t[:, qs:qe]
        elif name.endswith("feed_forward.w1.weight") or name.endswith("feed_forward.w3.weight"):
            out[name] = torch.chunk(t, world, dim=0)[rank]
        elif name.endswith("feed_forward.w2.weight"):
            out[name] = torch.chunk(t, world, dim=1)[rank]
        elif name == "output.weight":
            out[name] = torch.chunk(t, world, dim=0)[rank]
        else:
            out[name] = t
    lkh = e - s
    local = RefConfig(**{**cfg.__dict__, "n_head": lkh * g, "n_local_heads": lkh,
                         "dim": cfg.dim * lkh // cfg.n_local_heads})
    return out, local


# --------------------------------------------------------------------------- small ops
LINEAR_MODE = "fp32"     # "fp32": torch CPU bf16 linear (fp32 accumulation in the library's own order, one rounding);
#                          "fp64": products and sums in float64, ONE rounding to bf16 -- the correctly rounded result
#                          of the same bf16 operands.  Tests run the oracle in both modes: the distance between the two
#                          is the noise floor of "a valid bf16 implementation with another summation order", the
#                          yardstick the HIP engine's logits are gated with (tests/test_gpu_engine.py).


def linear(x, w, b=None, scales=None):
    """F.linear at the reference's rounding points (bf16 in, bf16 out), see LINEAR_MODE.  int8 `w` with per-row
    `scales`: WeightOnlyInt8Linear.forward (Engine/quantize.py:84-86) = F.linear(x, w.to(x.dtype)) * scales."""
    if w.dtype == torch.int8:
        return linear(x, w.to(x.dtype), b) * scales
    if LINEAR_MODE == "fp64":
        y = x.double() @ w.double().t()
        if b is not None:
            y = y + b.double()
        return y.to(x.dtype)
    return F.linear(x, w, b)


def rmsnorm(x, weight, eps):
    """Engine/SnapKV/model.py:458-469: fp32 norm -> cast -> * weight (bf16 multiply)."""
    xf = x.float()
    y = (xf * torch.rsqrt(torch.mean(xf * xf, dim=-1, keepdim=True) + eps)).type_as(x)
    return y * weight


def feed_forward(x, w1, w3, w2, s1=None, s3=None, s2=None):
    """Engine/SnapKV/model.py:451-455."""
    return linear(F.silu(linear(x, w1, None, s1)) * linear(x, w3, None, s3), w2, None, s2)


FP8_MAX, FP8_MARGIN = 448.0, 1.5


def calibrate_fp8_scales(k, v, margin=FP8_MARGIN):
    """Static per-kv-head scales of the fp8 cache from the first prefill chunk's UN-rotated k [rows, KH, D] and v:
    K's bound is the largest rotary-pair norm (what a rotation can put into one component), V's the largest |v|;
    scale = bound * margin / 448, floored at 2^-20.  (Specification of the product's KVCache.calibrate.)"""
    rows, KH, D = k.shape
    kb = k.float().view(rows, KH, D // 2, 2).square().sum(-1).amax(dim=(0, 2)).sqrt()
    vb = v.float().abs().amax(dim=(0, 2))
    return ((kb * (margin / FP8_MAX)).clamp_min(2.0 ** -20), (vb * (margin / FP8_MAX)).clamp_min(2.0 ** -20))


def tp_argmax_merge(vals, idx):
    """Engine/SnapKV/model.py:178-188 after the two all-reduces: vals/idx [.., tp] -> global index of the
    first (lowest-rank) maximum."""
    sel = torch.argmax(vals.float(), dim=-1, keepdim=True)
    return torch.gather(idx, -1, sel).squeeze(-1)


# --------------------------------------------------------------------------- SnapKV select
def snapkv_scores(q_win, k_ctx, g, window, kernel=POOL_KERNEL):
    """Engine/SnapKV/model.py:389-429 for ONE request: q_win [W, H, D] (rotated), k_ctx [S, KH, D].
    Returns the pooled, group-summed bf16 scores [KH, S-W] the top-k runs on, reproducing the
    reference's rounding sequence."""
    W = window
    S, KH, D = k_ctx.shape
    L = g * W
    cr = 8 * g                                                        # :406 chunk_size = nrepeat*8
    mask = torch.full((W, W), torch.finfo(BF16).min)
    ar = torch.arange(W)
    mask.masked_fill_(ar.view(1, -1) < (ar + 1).view(-1, 1), 0)       # :401-403
    out = []
    for h in range(KH):
        K = k_ctx[:, h].float()
        Q = q_win[:, h * g:(h + 1) * g].permute(1, 0, 2).reshape(L, D).float()   # rows (r,l)  :395
        acc = torch.zeros(g, S - W, dtype=BF16)                       # :409
        for c in range((L + cr - 1) // cr):
            if LINEAR_MODE == "fp64":     # correctly rounded scores (yardstick of the summation-order noise)
                s = (Q[c * cr:(c + 1) * cr].double() @ K.double().T).to(BF16)
            else:
                s = (Q[c * cr:(c + 1) * cr] @ K.T).to(BF16)           # :414 bf16 einsum, UNSCALED
            s[-W:, -W:] = (s[-W:, -W:].float() + mask).to(BF16)       # :415 last W rows x last W cols
            if LINEAR_MODE == "fp64":     # correctly rounded softmax (no dependence on expf / the sum order)
                p = torch.softmax(s.double(), dim=-1).to(BF16)
            else:
                p = torch.softmax(s.float(), dim=-1).to(BF16)         # :416
            gs = p.view(g, 8, S)[:, :, :S - W].float().sum(dim=1).to(BF16)   # :417-418 rows -> (r'=g, l'=8)
            acc = (acc.float() + gs.float()).to(BF16)
        xp = F.pad(acc.float(), (kernel // 2, kernel // 2))           # :421 avg_pool1d, count_include_pad
        pooled = xp[:, 0:S - W].clone()
        for i in range(1, kernel):
            pooled = pooled + xp[:, i:i + S - W]
        pooled = (pooled / kernel).to(BF16)
        out.append(pooled.float().sum(dim=0).to(BF16))                # :428-429
    return torch.stack(out)


def topk_desc_stable(scores, k):
    """Deterministic top-k: descending score, ties -> lowest index.  (torch.topk's tie order is
    implementation-defined; the reference's is therefore only comparable tie-aware.)"""
    return torch.sort(scores.float(), dim=-1, descending=True, stable=True).indices[..., :k]


def snapkv_select(q_win, k_ctx, v_ctx, g, window, budget, idx=None):
    """Engine/SnapKV/model.py:431-439 for one request: returns (idx [KH, budget-W], new_k, new_v
    [budget, KH, D]) -- selected rows in descending-score order followed by the last W rows.
    `idx` overrides the top-k (used only to replay the reference's own tie resolution from a fixture)."""
    sc = snapkv_scores(q_win, k_ctx, g, window)
    if idx is None:
        idx = topk_desc_stable(sc, budget - window)
    S, KH, D = k_ctx.shape
    nk = torch.empty(budget, KH, D, dtype=k_ctx.dtype)
    nv = torch.empty(budget, KH, D, dtype=v_ctx.dtype)
    for h in range(KH):
        nk[:budget - window, h] = k_ctx[idx[h], h]
        nv[:budget - window, h] = v_ctx[idx[h], h]
        nk[budget - window:, h] = k_ctx[S - window:, h]
        nv[budget - window:, h] = v_ctx[S - window:, h]
    return idx, nk, nv, sc


# --------------------------------------------------------------------------- StreamingLLM eviction
def streaming_prefill_kv(cache, k, v, B, ctx, n, kv_len, tab, rope_fn, is_last):
    """KVCache.prefill, Engine/StreamingLLM/model_draft.py:102-143 (== prefill_draft, StreamingLLM/model.py:116-157).
    `cache` holds UN-rotated keys until the last chunk; returns the rotated copy (positions = slot index)
    the chunk's attention runs on.  Sinks = 16, window = kv_len-16; the shift covers the whole slot range
    [16, kv_len) even while its tail has never been written (bug-for-bug)."""
    KH, D = k.shape[1], k.shape[2]
    ppr = cache.shape[0] // B
    zeros = torch.zeros(B, dtype=torch.int32)
    if ctx + n <= kv_len:
        ip = (torch.arange(B + 1) * n).to(torch.int32)
        fr.append_paged_kv_cache(k, v, ip, cache, tab["indices"], tab["indptr"], tab["last"])
        valid = ctx + n
    else:
        flat_k = cache[:, 0].reshape(B, -1, KH, D)
        flat_v = cache[:, 1].reshape(B, -1, KH, D)
        new_k = torch.cat((flat_k[:, SINK:kv_len], k.reshape(B, n, KH, D)), dim=1)[:, -(kv_len - SINK):]
        new_v = torch.cat((flat_v[:, SINK:kv_len], v.reshape(B, n, KH, D)), dim=1)[:, -(kv_len - SINK):]
        ip = (torch.arange(B + 1) * (kv_len - SINK)).to(torch.int32)
        fr.append_paged_kv_cache(new_k.reshape(-1, KH, D).contiguous(), new_v.reshape(-1, KH, D).contiguous(), ip,
                                 cache, tab["indices"], tab["indptr"], tab["last"])
        valid = kv_len
    rot = cache.clone()
    keys = rot[:, 0].reshape(B, -1, KH, D)
    kr = keys[:, :valid].reshape(-1, KH, D)
    ip = (torch.arange(B + 1) * valid).to(torch.int32)
    keys[:, :valid] = rope_fn(kr, kr, ip, zeros)[1].reshape(B, valid, KH, D)
    rot[:, 0] = keys.reshape(B * ppr, PAGE, KH, D)
    if ctx + n > kv_len and is_last:
        cache.copy_(rot)                                               # :141-142
    return rot


# --------------------------------------------------------------------------- accept loop
def accept_step(tokens_buffer, target_tokens, output, num_nodes, cachelens, last_page_len, draft_cachelens,
                draft_last_page_len, gamma, draft_rollback, draft_cap, eot_1, eot_2, max_nodes, use_double):
    """The verify-loop body, tests/SnapKV/longspec_benchmark.py:208-285 (longspec: draft_rollback=gamma,
    draft_cap=gamma, use_double=True), tests/SnapKV/selfspec_benchmark.py:145-211 (gamma+1, gamma+1, False),
    tests/StreamingLLM/selfspec_benchmark.py:159-238 (gamma, gamma, True).  All tensors updated in place
    (numpy-style); returns dict(accept_nums, bonus, terminal, next_double, double_buffer, cachelens_update)."""
    B = tokens_buffer.shape[0]
    draft = tokens_buffer[:, 1:gamma + 1]
    flag = target_tokens[:, :gamma] == draft
    eot = (draft == eot_1) | (draft == eot_2)
    acc = torch.cumprod((flag & ~eot).int(), dim=1).bool()
    accept_nums = acc.sum(dim=1) + 1
    terminal = bool((eot & acc).any())            # never true: acc excludes eot (kept bug-for-bug)
    cachelens -= gamma + 1
    last_page_len -= gamma + 1
    for b in range(B):
        a = int(accept_nums[b])
        c = int(cachelens[b])
        output[b, c:c + a] = tokens_buffer[b, :a]
    cachelens += accept_nums.to(cachelens.dtype)
    last_page_len += accept_nums.to(last_page_len.dtype)
    if draft_cachelens is not None:
        adv = torch.clamp(accept_nums, max=draft_cap).to(draft_cachelens.dtype)
        draft_cachelens += adv - draft_rollback
        draft_last_page_len += adv - draft_rollback
    bonus = target_tokens.gather(1, (accept_nums - 1).view(-1, 1)).view(-1)
    if bool((bonus == eot_1).any()) or bool((bonus == eot_2).any()):
        terminal = True
    num_nodes += accept_nums
    if int(num_nodes.max()) >= max_nodes:
        terminal = True
    res = dict(accept_nums=accept_nums.clone(), bonus=bonus.clone(), next_double=False, double_buffer=None,
               cachelens_update=None)
    if not terminal:
        tokens_buffer[:, 0] = bonus
        if use_double and int(accept_nums.max()) == gamma + 1:
            m = accept_nums == gamma + 1
            db = torch.zeros(B, 2, dtype=torch.long)
            db[:, 0] = torch.where(m, tokens_buffer[:, -1], bonus)
            db[:, 1] = torch.where(m, bonus, torch.zeros_like(bonus))
            res.update(next_double=True, double_buffer=db, cachelens_update=torch.where(m, 2, 1))
    else:
        for b in range(B):
            output[b, int(num_nodes[b])] = bonus[b]
        num_nodes += 1
    res["terminal"] = terminal
    return res


# --------------------------------------------------------------------------- the model
class RefModel:
    """Functional restatement of Transformer/TransformerBlock/Attention/FeedForward
    (Engine/SnapKV/model.py:114-455 and the StreamingLLM twins)."""

    def __init__(self, cfg: RefConfig, sd, max_pos=1 << 15, group=None, rank=0, world=1, vocab_shard=None):
        self.cfg = cfg
        self.sd = sd
        self.group, self.rank, self.world = group, rank, world
        self.vocab_shard = vocab_shard if vocab_shard is not None else sd["output.weight"].shape[0]
        llama31 = cfg.high_freq_factor is not None and cfg.low_freq_factor is not None
        self.rope_table = fr.rope_table(max_pos, cfg.head_dim, cfg.rope_base, cfg.scaling_factor,
                                        cfg.low_freq_factor if llama31 else None,
                                        cfg.high_freq_factor if llama31 else None,
                                        cfg.original_max_position_embeddings if llama31 else None)

    # -- helpers
    def _all_reduce(self, y):
        """Engine/SnapKV/model.py:334-335: sum all-reduce of the bf16 partial (gloo on CPU)."""
        if self.group is not None:
            import torch.distributed as dist
            dist.all_reduce(y, group=self.group)
        return y

    def qkv(self, x, i):
        c = self.cfg
        p = f"layers.{i}.attention."
        B, n, _ = x.shape
        kv = c.n_local_heads * c.head_dim
        y = linear(x, self.sd[p + "wqkv.weight"], self.sd.get(p + "wqkv.bias"), self.sd.get(p + "wqkv.scales"))
        q, k, v = y.split([c.n_head * c.head_dim, kv, kv], dim=-1)
        return (q.reshape(B * n, c.n_head, c.head_dim), k.reshape(B * n, c.n_local_heads, c.head_dim),
                v.reshape(B * n, c.n_local_heads, c.head_dim))

    def rope(self, q, k, indptr, offsets):
        return fr.apply_rope(q, k, indptr, offsets, self.rope_table)

    def attn(self, q, cache, qo_indptr, tab):
        c = self.cfg
        return fr.batch_prefill_paged(q, cache, qo_indptr, tab["indices"], tab["indptr"], tab["last"], c.n_head,
                                      c.n_local_heads, c.head_dim, causal=True)

    def block(self, x, i, attn_fn):
        """TransformerBlock.* (:260-278): h = x + attn(norm(x)); out = h + ffn(norm(h))."""
        p = f"layers.{i}."
        B, n, _ = x.shape
        y = attn_fn(rmsnorm(x, self.sd[p + "attention_norm.weight"], self.cfg.norm_eps), i)
        y = self._all_reduce(linear(y.reshape(B, n, -1), self.sd[p + "attention.wo.weight"], None,
                                    self.sd.get(p + "attention.wo.scales")))
        h = x + y
        f = feed_forward(rmsnorm(h, self.sd[p + "ffn_norm.weight"], self.cfg.norm_eps),
                         self.sd[p + "feed_forward.w1.weight"], self.sd[p + "feed_forward.w3.weight"],
                         self.sd[p + "feed_forward.w2.weight"], self.sd.get(p + "feed_forward.w1.scales"),
                         self.sd.get(p + "feed_forward.w3.scales"), self.sd.get(p + "feed_forward.w2.scales"))
        return h + self._all_reduce(f)

    def head(self, x, return_logits=False):
        """:175-188 final norm -> lm head -> argmax (TP: merge of per-rank maxima)."""
        x = rmsnorm(x, self.sd["norm.weight"], self.cfg.norm_eps)
        logits = linear(x, self.sd["output.weight"], None, self.sd.get("output.scales"))
        self.last_logits = logits          # kept for tie-aware token comparisons in tests
        if return_logits:
            return logits
        if self.group is not None:
            import torch.distributed as dist
            vals = torch.zeros(*logits.shape[:2], self.world, dtype=logits.dtype)
            idx = torch.zeros(*logits.shape[:2], self.world, dtype=torch.long)
            vals[..., self.rank], idx[..., self.rank] = torch.max(logits, dim=-1)
            idx[..., self.rank] += self.rank * logits.shape[-1]
            dist.all_reduce(vals, group=self.group)
            dist.all_reduce(idx, group=self.group)
            return tp_argmax_merge(vals, idx)
        return torch.argmax(logits, dim=-1)

    def run(self, idx, attn_fn, return_logits=False):
        x = F.embedding(idx, self.sd["tok_embeddings.weight"])
        for i in range(self.cfg.n_layer):
            x = self.block(x, i, attn_fn)
        return self.head(x, return_logits)


def _table(indices, indptr, last):
    return dict(indices=indices, indptr=indptr, last=last)


# --------------------------------------------------------------------------- engines
class RefEngine:
    """The four reference back-ends as one state machine.

    mode:
      "target"           Engine/SnapKV/backend.py LMBackend(dec_len)            (longspec target, baseline)
      "snapkv_self"      Engine/SnapKV/backend.py LMBackend(dec_len, draft_dec_len)   (self-spec)
      "snapkv_draft"     Engine/SnapKV/backend_draft.py LMBackend_Draft(draft_budget)
      "stream_draft"     Engine/StreamingLLM/backend_draft.py LMBackend_Draft
      "stream_self"      Engine/StreamingLLM/backend.py LMBackend
    Attribute names follow the reference (cachelens, paged_kv_last_page_len, draft_*), and like the
    reference they may be rebound from outside between calls.
    """

    def __init__(self, mode, cfg: RefConfig, sd, B, max_len=0, draft_budget=0, window=32, group=None, rank=0,
                 world=1, max_pos=1 << 15, kv_fp8=False):
        """kv_fp8: emulate the product's e4m3fn full-context cache (not a reference feature; specification =
        flashinfer_ref.quantize_fp8 / dequantize_cache_fp8 + calibrate_fp8_scales below).  The full cache then
        holds the EXACT float32 products byte * scale; compressed draft caches stay bf16."""
        assert mode in ("target", "snapkv_self", "snapkv_draft", "stream_draft", "stream_self")
        self.kv_fp8 = kv_fp8
        self.kv_scales = [None] * cfg.n_layer
        self.kv_scale_override = None     # [(k_scale, v_scale)] per layer: replaces the first-chunk calibration
        self.mode, self.cfg, self.B = mode, cfg, B
        self.model = RefModel(cfg, sd, max_pos=max_pos, group=group, rank=rank, world=world)
        self.budget, self.window = draft_budget, window
        KH, D = cfg.n_local_heads, cfg.head_dim
        self.has_full = mode != "stream_draft"
        self.has_draft = mode in ("snapkv_self", "snapkv_draft", "stream_self") and draft_budget != -1
        if mode == "stream_draft":
            self.ppr = draft_budget // PAGE + 1                      # StreamingLLM/backend_draft.py:27-28
            self.has_draft = False
        else:
            npages = B * max_len // PAGE                            # SnapKV/backend.py:32-35
            if npages * PAGE < B * max_len:
                npages += B
            self.ppr = npages // B
        self.caches = [torch.zeros(B * self.ppr, 2, PAGE, KH, D, dtype=torch.float32 if kv_fp8 else BF16)
                       for _ in range(cfg.n_layer)]
        if self.has_draft:
            self.dppr = draft_budget // PAGE + 1
            self.draft_caches = [torch.zeros(B * self.dppr, 2, PAGE, KH, D, dtype=BF16) for _ in range(cfg.n_layer)]
        self.snap_idx = None
        self.topk_replay, self._replay_pos = None, 0   # fixture replay of torch.topk's tie order (tests only)
        self.clear_kv()

    # ----- page-table state (clear_kv of the respective backend)
    def clear_kv(self):
        B = self.B
        for c in self.caches:
            c.zero_()
        self.kv_scales = [None] * self.cfg.n_layer      # re-calibrated by every encode (first prefill chunk)
        self.cachelens = torch.zeros(B, dtype=torch.int32)
        self.qo_indptr = torch.arange(B + 1, dtype=torch.int32)
        self.paged_kv_indptr = torch.arange(B + 1, dtype=torch.int32)
        self.paged_kv_indices = torch.zeros(B * self.ppr, dtype=torch.int32)
        self.paged_kv_last_page_len = torch.zeros(B, dtype=torch.int32)
        self.num_pages_per_request = torch.zeros(B, dtype=torch.int32)
        if self.has_draft:
            for c in self.draft_caches:
                c.zero_()
            if self.mode == "snapkv_self":
                self.draft_cachelens = torch.zeros(B, dtype=torch.int32)
            if self.mode == "stream_self":
                self.draft_cachelens = torch.zeros(B, dtype=torch.int32)
                self.draft_num_pages_per_request = torch.zeros(B, dtype=torch.int32)
                self.draft_paged_kv_indptr = torch.arange(B + 1, dtype=torch.int32) * self.dppr
                self.draft_paged_kv_indices = torch.zeros(B * self.dppr, dtype=torch.int32)
                self.draft_paged_kv_last_page_len = torch.zeros(B, dtype=torch.int32)
            else:  # SnapKV: Engine/SnapKV/backend.py:87-90 -- last_page_len starts at ONE
                self.draft_paged_kv_indptr = torch.arange(B + 1, dtype=torch.int32) * self.dppr
                self.draft_paged_kv_indices = torch.arange(B * self.dppr, dtype=torch.int32)
                self.draft_paged_kv_last_page_len = torch.ones(B, dtype=torch.int32)

    def _grow_pages(self, prefix=""):
        """pre_encode (Engine/SnapKV/backend.py:270-274): one more page per request, contiguous arange."""
        npr = getattr(self, prefix + "num_pages_per_request")
        ppr = self.dppr if prefix else self.ppr
        npr += 1
        setattr(self, prefix + "paged_kv_indices",
                torch.cat([torch.arange(i * ppr, i * ppr + int(npr[i]), dtype=torch.int32) for i in range(self.B)]))
        ip = getattr(self, prefix + "paged_kv_indptr")
        ip[1:] = torch.cumsum(npr, dim=0, dtype=torch.int32)

    def _full_pages(self, prefix=""):
        """Streaming pre_encode overflow branch (StreamingLLM/backend_draft.py:174-179)."""
        npr = getattr(self, prefix + "num_pages_per_request")
        ppr = self.dppr if prefix else self.ppr
        npr.fill_(ppr)
        setattr(self, prefix + "paged_kv_indices",
                torch.cat([torch.arange(i * ppr, (i + 1) * ppr, dtype=torch.int32) for i in range(self.B)]))
        ip = getattr(self, prefix + "paged_kv_indptr")
        ip[1:] = torch.cumsum(npr, dim=0, dtype=torch.int32)

    def _tab(self, prefix=""):
        return _table(getattr(self, prefix + "paged_kv_indices"), getattr(self, prefix + "paged_kv_indptr"),
                      getattr(self, prefix + "paged_kv_last_page_len"))

    # ----- attention variants
    def _attn_std(self, n, offsets, caches, tab, caches2=None, tab2=None, snap=False, debug=None):
        """Attention.forward / verify / draft_forward / prefill of Engine/SnapKV/model.py:322-387:
        rope(offsets) -> append -> attention [-> also append to a second cache] [-> gen_draft_kv]."""
        m = self.model
        qo = self.qo_indptr * n

        def fn(xn, i):
            q, k, v = m.qkv(xn, i)
            if self.kv_fp8 and caches is self.caches and self.kv_scales[i] is None:
                self.kv_scales[i] = (self.kv_scale_override[i] if self.kv_scale_override is not None
                                     else calibrate_fp8_scales(k, v))
            q, k = m.rope(q, k, qo, offsets)
            if self.kv_fp8 and caches is self.caches:
                ks, vs = self.kv_scales[i]
                k8 = fr.quantize_fp8(k, ks).float() * ks.view(1, -1, 1)      # exact byte * scale products
                v8 = fr.quantize_fp8(v, vs).float() * vs.view(1, -1, 1)
                fr.append_paged_kv_cache(k8, v8, qo, caches[i], tab["indices"], tab["indptr"], tab["last"])
            else:
                fr.append_paged_kv_cache(k, v, qo, caches[i], tab["indices"], tab["indptr"], tab["last"])
            if caches2 is not None:
                fr.append_paged_kv_cache(k, v, qo, caches2[i], tab2["indices"], tab2["indptr"], tab2["last"])
            y = m.attn(q, caches[i], qo, tab)
            if snap:
                self._gen_draft_kv(q, i, n, int(offsets[0]) + n)
            return y
        return fn

    def _gen_draft_kv(self, q, i, n, ctx_len):
        """Caller Engine/SnapKV/model.py:381-382 + gen_draft_kv :389-439."""
        c = self.cfg
        g = c.n_head // c.n_local_heads
        tab = self._tab()
        dtab = self._tab("draft_")
        if i == 0:
            self.snap_idx, self.snap_scores = [], []
        li, ls = [], []
        nk, nv = [], []
        for b in range(self.B):
            k, v = fr.gather_request_kv(self.caches[i], tab["indices"], tab["indptr"], tab["last"], b)
            ov = None
            if self.topk_replay is not None:      # [B, KH, topk] of this layer, recorded from the reference
                ov = torch.as_tensor(self.topk_replay[self._replay_pos])[b]
            idx, kk, vv, sc = snapkv_select(q[b * n:(b + 1) * n], k[:ctx_len], v[:ctx_len], g, self.window, self.budget,
                                            idx=ov)
            li.append(idx)
            ls.append(sc)
            nk.append(kk)
            nv.append(vv)
        self.snap_idx.append(torch.stack(li))
        self.snap_scores.append(torch.stack(ls))
        if self.topk_replay is not None:
            self._replay_pos += 1
        ip = (torch.arange(self.B + 1) * self.budget).to(torch.int32)
        fr.append_paged_kv_cache(torch.cat(nk), torch.cat(nv), ip, self.draft_caches[i], dtab["indices"],
                                 dtab["indptr"], dtab["last"])

    def _attn_stream_prefill(self, n, ctx, caches, tab, is_last, kv_len):
        """Attention.prefill of Engine/StreamingLLM/model_draft.py:309-326 (== draft_prefill of
        StreamingLLM/model.py:402-419): q rotated with the cache-relative offset, K/V through
        streaming_prefill_kv, attention on the rotated copy."""
        m = self.model
        B = self.B
        qo = self.qo_indptr * n

        def fn(xn, i):
            q, k, v = m.qkv(xn, i)
            off = ctx if ctx + n <= kv_len else kv_len - n
            q, _ = m.rope(q, k, qo, torch.full((B,), off, dtype=torch.int32))
            rot = streaming_prefill_kv(caches[i], k, v, B, ctx, n, kv_len, tab, m.rope, is_last)
            return m.attn(q, rot, qo, tab)
        return fn

    # ----- public API (names as in the reference back-ends)
    @torch.no_grad()
    def encode(self, input_ids):
        """SnapKV/backend.py:232-268, backend_draft.py:176-209, StreamingLLM/backend_draft.py:127-153,
        StreamingLLM/backend.py:190-211."""
        self.clear_kv()
        S = input_ids.shape[1]
        tokens = None
        is_last = False
        for st in range(0, S, CHUNK):
            ids = input_ids[:, st:st + CHUNK]
            n = ids.shape[1]
            if n != CHUNK:
                is_last = True
            if self.mode == "stream_draft":
                ctx = int(self.cachelens[0])
                if ctx + n <= self.budget:
                    self._grow_pages()
                    self.paged_kv_last_page_len = torch.full((self.B,), n, dtype=torch.int32)
                else:
                    self._full_pages()
                    self.paged_kv_last_page_len = torch.full((self.B,), self.budget % PAGE, dtype=torch.int32)
                fn = self._attn_stream_prefill(n, ctx, self.caches, self._tab(), is_last, self.budget)
                tokens = self.model.run(ids, fn)
                self.cachelens += n
                if int(self.cachelens[0]) >= self.budget:
                    self.cachelens.fill_(self.budget)
                continue
            self._grow_pages()
            self.paged_kv_last_page_len = torch.full((self.B,), n, dtype=torch.int32)
            snap = self.mode in ("snapkv_self", "snapkv_draft") and self.has_draft and is_last
            fn = self._attn_std(n, self.cachelens, self.caches, self._tab(), snap=snap)
            tokens = self.model.run(ids, fn)
            self.cachelens += n
        if self.mode == "snapkv_self":
            self.draft_cachelens.copy_(self.cachelens)
        return tokens

    @torch.no_grad()
    def draft_encode(self, input_ids):
        """StreamingLLM/backend.py:234-258 (self-spec: second pass filling the streaming draft cache)."""
        assert self.mode == "stream_self"
        S = input_ids.shape[1]
        tokens = None
        is_last = False
        for st in range(0, S, CHUNK):
            ids = input_ids[:, st:st + CHUNK]
            n = ids.shape[1]
            if n != CHUNK:
                is_last = True
            ctx = int(self.draft_cachelens[0])
            if ctx + n <= self.budget:
                self._grow_pages("draft_")
                self.draft_paged_kv_last_page_len = torch.full((self.B,), n, dtype=torch.int32)
            else:
                self._full_pages("draft_")
                self.draft_paged_kv_last_page_len = torch.full((self.B,), self.budget % PAGE, dtype=torch.int32)
            fn = self._attn_stream_prefill(n, ctx, self.draft_caches, self._tab("draft_"), is_last, self.budget)
            tokens = self.model.run(ids, fn)
            self.draft_cachelens += n
            if int(self.draft_cachelens[0]) >= self.budget:
                self.draft_cachelens.fill_(self.budget)
        return tokens

    @torch.no_grad()
    def inference(self, input_ids, cachelen_update=None, return_logits=False):
        """target: SnapKV/backend.py:129-159.  drafts: backend_draft.py:113-173 (cachelen_update path)."""
        n = input_ids.shape[1]
        if self.mode == "snapkv_draft" and self.has_draft:
            self.draft_paged_kv_last_page_len += n
            fn = self._attn_std(n, self.cachelens, self.draft_caches, self._tab("draft_"))
            lp = "draft_paged_kv_last_page_len"
        else:
            self.paged_kv_last_page_len += n
            fn = self._attn_std(n, self.cachelens, self.caches, self._tab())
            lp = "paged_kv_last_page_len"
        out = self.model.run(input_ids, fn, return_logits)
        if cachelen_update is None:
            self.cachelens += n
        else:
            cu = cachelen_update.to(torch.int32).flatten()
            self.cachelens += cu
            setattr(self, lp, getattr(self, lp) - n + cu)
        return out

    @torch.no_grad()
    def verify(self, input_ids, return_logits=False):
        """SnapKV/backend.py:163-197 (+ Attention.verify model.py:338-353); StreamingLLM/backend.py:120-150."""
        n = input_ids.shape[1]
        self.paged_kv_last_page_len += n
        if self.mode == "snapkv_self":
            self.draft_paged_kv_last_page_len += 1
            self.draft_cachelens += 1
            fn = self._attn_std(n, self.cachelens, self.caches, self._tab(), self.draft_caches, self._tab("draft_"))
        else:
            fn = self._attn_std(n, self.cachelens, self.caches, self._tab())
        out = self.model.run(input_ids, fn, return_logits)
        self.cachelens += n
        return out

    @torch.no_grad()
    def speculate(self, input_ids, cachelen_update=None):
        """SnapKV/backend.py:200-229; StreamingLLM/backend.py:152-188."""
        n = input_ids.shape[1]
        self.draft_paged_kv_last_page_len += n
        fn = self._attn_std(n, self.draft_cachelens, self.draft_caches, self._tab("draft_"))
        out = self.model.run(input_ids, fn)
        if cachelen_update is None:
            self.draft_cachelens += n
        else:
            cu = cachelen_update.to(torch.int32).flatten()
            self.draft_cachelens += cu
            self.draft_paged_kv_last_page_len = self.draft_paged_kv_last_page_len - n + cu
        return out

    # draft back-ends of the longspec harness expose the draft table under the plain names
    def draft_view(self):
        return self
