"""The tiny GQA configs the golden fixtures were generated with (mirrors oracle/gen_golden.py:TINY)."""
import json
import os

import numpy as np
import torch

from oracle.magicdec_ref import RefConfig, init_state_dict

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

TINY = {
    "tinytgt": (RefConfig(n_layer=2, n_head=8, n_local_heads=2, dim=512, intermediate_size=1024, vocab_size=2048,
                          rope_base=500000.0, scaling_factor=8, high_freq_factor=4, low_freq_factor=1,
                          original_max_position_embeddings=8192), 11, 0.1),
    "tinydrf": (RefConfig(n_layer=1, n_head=8, n_local_heads=2, dim=512, intermediate_size=1024, vocab_size=2048,
                          rope_base=10000.0), 12, 0.1),
    # other families of the reference's zoo (oracle/gen_golden.py holds the same numbers):
    # Qwen2.5-style: qkv bias, g = 5 (padded MFMA M tile, mis-aligned SnapKV mask), eps 1e-6, plain RoPE theta 1e6
    "tinyqwen": (RefConfig(n_layer=2, n_head=10, n_local_heads=2, dim=640, intermediate_size=1280, vocab_size=2048,
                           rope_base=1000000.0, norm_eps=1e-6, qkv_bias=True), 21, 0.1),
    # Llama-3.1-70B-style: g = 8 -> two MFMA M tiles per (request, kv head) in the verify kernel, D = 128
    "tiny70b": (RefConfig(n_layer=2, n_head=16, n_local_heads=2, dim=2048, intermediate_size=2048, vocab_size=2048,
                          rope_base=500000.0, scaling_factor=8, high_freq_factor=4, low_freq_factor=1,
                          original_max_position_embeddings=8192), 22, 0.1),
    # four kv heads (g = 4, D = 64): shards evenly over TP4 (target) and a TP2 draft sub-group
    "tinykh4": (RefConfig(n_layer=2, n_head=16, n_local_heads=4, dim=1024, intermediate_size=1024, vocab_size=2048,
                          rope_base=500000.0, scaling_factor=8, high_freq_factor=4, low_freq_factor=1,
                          original_max_position_embeddings=8192), 24, 0.1),
}
B, S, MAX_LEN, GAMMA, BUDGET, EOT_1, EOT_2 = 2, 416, 512, 3, 129, 2, 0


def tiny(name):
    cfg, seed, wo = TINY[name]
    return cfg, init_state_dict(cfg, seed, wo_scale=wo)


def config_kwargs(cfg):
    """RefConfig -> kwargs of magicdec_amd.Engine.model_core.transformer_configs (the reference's ModelArgs names)."""
    return dict(block_size=4096, n_layer=cfg.n_layer, n_head=cfg.n_head, n_local_heads=cfg.n_local_heads, dim=cfg.dim,
                intermediate_size=cfg.intermediate_size, vocab_size=cfg.vocab_size, rope_base=cfg.rope_base,
                norm_eps=cfg.norm_eps, scaling_factor=cfg.scaling_factor, high_freq_factor=cfg.high_freq_factor,
                low_freq_factor=cfg.low_freq_factor,
                original_max_position_embeddings=cfg.original_max_position_embeddings, qkv_bias=cfg.qkv_bias)


def register_tiny(model_core):
    for name in TINY:
        model_core.transformer_configs[name] = config_kwargs(TINY[name][0])


def synthetic_batches(n_seq=12, vocab=2048, prefix=S, seed=123):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(4, vocab, (n_seq, prefix), generator=g)
    ids[:, 0] = 1
    return [ids[i:i + B] for i in range(0, n_seq, B)]


def load_json(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def from_bits(a):
    return torch.from_numpy(a.astype(np.int16)).view(torch.bfloat16)


def benchflag_inputs(ncols, call_index, vocab=2048, batch=B):
    """Token ids of call `call_index` of a benchflag_* fixture's program (oracle/gen_golden.py:benchflag_inputs)."""
    g = torch.Generator().manual_seed(1000 + call_index)
    ids = torch.randint(4, vocab, (batch, ncols), generator=g)
    if ncols > 8:
        ids[:, 0] = 1
    return ids


def peaked_pair(emb_gain=16.0, peak=12.0, draft_miss_every=4, seed=77):
    """A target / draft pair whose next-token distributions are PEAKED, as a trained model's are (random-init heads
    give ~2048 nearly tied Gaussian logits, where any rounding difference legitimately flips an argmax): the head is
    tied to the embedding through a fixed permutation pi of the non-special token ids, `output[j] = c * emb[pi(j)]`,
    and the embedding dominates the residual stream (gain `emb_gain` over the layers' contributions), so that the logit
    of the one token j with pi(j) == current token stands at ~`peak` while the other 2047 stay within ~+-2 -- a top-2
    gap of hundreds of bf16 ulps, with the attention / MLP branches (all kernels still run at full strength) as the
    perturbation.  The draft shares embedding and layers' shapes but mispredicts every `draft_miss_every`-th token id
    (its permutation differs there), so rejections and rollbacks occur at a known rate.
    Returns ((cfg_t, sd_t), (cfg_d, sd_d))."""
    cfg_t, sd_t = tiny("tinytgt")
    cfg_d, sd_d = tiny("tinydrf")
    g = torch.Generator().manual_seed(seed)
    V, dim = cfg_t.vocab_size, cfg_t.dim
    assert (cfg_d.vocab_size, cfg_d.dim) == (V, dim)
    emb = (torch.randn(V, dim, generator=g) * 0.02 * emb_gain)
    perm = torch.arange(V)
    perm[4:] = 4 + torch.randperm(V - 4, generator=g)           # ids 0..3 (BOS / EOT ids of the tests) stay fixed
    c = peak / (dim * 0.02 * emb_gain)                          # logit of the tied row: |e|^2 / rms(e) * c = peak
    head_t = emb[perm] * c
    perm_d = perm.clone()
    miss = torch.arange(4, V, draft_miss_every)
    perm_d[miss] = perm[torch.roll(miss, 1)]                    # the draft's tied row is another token's there
    head_d = emb[perm_d] * c
    sd_t, sd_d = dict(sd_t), dict(sd_d)
    for sd, head in ((sd_t, head_t), (sd_d, head_d)):
        sd["tok_embeddings.weight"] = emb.to(torch.bfloat16)
        sd["output.weight"] = head.to(torch.bfloat16)
    return (cfg_t, sd_t), (cfg_d, sd_d)
