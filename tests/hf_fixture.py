"""Seeded tiny HuggingFace-layout checkpoints (inputs of the checkpoint-conversion fixture).

Shared by oracle/gen_golden.py (which runs the REAL reference converter on them and records key lists + tensor
hashes into tests/golden/convert_hf.json) and tests/test_host_cpu.py (which runs OUR converter on the same files)."""
import hashlib
import json
import os

import torch

# case name -> (checkpoint directory name == config name, tied lm head, sharded safetensors with an index file)
CASES = {
    "llama_sharded": ("tinytgt", False, True),
    "qwen_bias_tied_single": ("tinyqwen", True, False),
}


def hf_state_dict(cfg, seed, tied):
    """HF parameter names / shapes of a Llama- or Qwen2-style model with the given (reference-style) config."""
    g = torch.Generator().manual_seed(seed)
    D = cfg.dim // cfg.n_head
    kv = cfg.n_local_heads * D

    def w(*shape):
        return (torch.randn(*shape, generator=g) * 0.05).to(torch.bfloat16)

    sd = {"model.embed_tokens.weight": w(cfg.vocab_size, cfg.dim), "model.norm.weight": w(cfg.dim)}
    if not tied:
        sd["lm_head.weight"] = w(cfg.vocab_size, cfg.dim)
    for i in range(cfg.n_layer):
        p = f"model.layers.{i}."
        sd[p + "self_attn.q_proj.weight"] = w(cfg.dim, cfg.dim)
        sd[p + "self_attn.k_proj.weight"] = w(kv, cfg.dim)
        sd[p + "self_attn.v_proj.weight"] = w(kv, cfg.dim)
        sd[p + "self_attn.o_proj.weight"] = w(cfg.dim, cfg.dim)
        if cfg.qkv_bias:
            sd[p + "self_attn.q_proj.bias"] = w(cfg.dim)
            sd[p + "self_attn.k_proj.bias"] = w(kv)
            sd[p + "self_attn.v_proj.bias"] = w(kv)
        sd[p + "mlp.gate_proj.weight"] = w(cfg.intermediate_size, cfg.dim)
        sd[p + "mlp.up_proj.weight"] = w(cfg.intermediate_size, cfg.dim)
        sd[p + "mlp.down_proj.weight"] = w(cfg.dim, cfg.intermediate_size)
        sd[p + "input_layernorm.weight"] = w(cfg.dim)
        sd[p + "post_attention_layernorm.weight"] = w(cfg.dim)
        sd[p + "self_attn.rotary_emb.inv_freq"] = torch.arange(D // 2, dtype=torch.float32)   # dropped by converters
    return sd


def write_hf_checkpoint(root, case, cfg, seed=77):
    """<root>/<config name>/ with model.safetensors (or two shards + model.safetensors.index.json)."""
    from safetensors.torch import save_file
    name, tied, sharded = CASES[case]
    d = os.path.join(root, name)
    os.makedirs(d, exist_ok=True)
    sd = hf_state_dict(cfg, seed, tied)
    if not sharded:
        save_file(sd, os.path.join(d, "model.safetensors"))
        return d
    keys = list(sd)
    halves = [keys[:len(keys) // 2], keys[len(keys) // 2:]]
    weight_map = {}
    for i, ks in enumerate(halves):
        fn = f"model-0000{i + 1}-of-00002.safetensors"
        save_file({k: sd[k] for k in ks}, os.path.join(d, fn))
        weight_map.update({k: fn for k in ks})
    with open(os.path.join(d, "model.safetensors.index.json"), "w") as f:
        json.dump({"metadata": {}, "weight_map": weight_map}, f)
    return d


def describe(state_dict):
    """{key: [shape, dtype, sha256 of the raw bytes]} of a converted model.pth."""
    out = {}
    for k, t in state_dict.items():
        raw = t.contiguous().view(torch.uint8).numpy().tobytes() if t.dtype != torch.bfloat16 else \
            t.contiguous().view(torch.int16).numpy().tobytes()
        out[k] = [list(t.shape), str(t.dtype), hashlib.sha256(raw).hexdigest()]
    return out
