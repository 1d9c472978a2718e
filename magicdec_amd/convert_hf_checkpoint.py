"""HF checkpoint -> `model.pth` in the layout the Engine loads (row "next" 8f-3 of SURVEY.md; the reference's
convert_hf_checkpoint.py:79-163 produces the same file, so either tool's output loads here and in the reference).

Layout facts that are binding for the kernels:
  * q and k projection rows are permuted from HF's half-split RoPE order to the INTERLEAVED order the RoPE kernel
    rotates ((x[2i], x[2i+1]) pairs): per head, row r of the first half and row r of the second half become rows
    2r and 2r+1;
  * wq, wk, wv are fused row-wise into `wqkv = [q; k; v]` (biases likewise for Qwen);
  * a tied lm head is materialised as `output.weight = embed_tokens.weight`.

    python -m magicdec_amd.convert_hf_checkpoint --checkpoint_dir checkpoints/meta-llama/Meta-Llama-3.1-8B
"""
from __future__ import annotations

import argparse
import json
import re
from pathlib import Path

import torch

from .Engine.model_core import ModelArgs

_LAYER = re.compile(r"^model\.layers\.(\d+)\.(.+)$")
_PER_LAYER = {
    "self_attn.q_proj.weight": "attention.wq.weight", "self_attn.k_proj.weight": "attention.wk.weight",
    "self_attn.v_proj.weight": "attention.wv.weight", "self_attn.o_proj.weight": "attention.wo.weight",
    "self_attn.q_proj.bias": "attention.wq.bias", "self_attn.k_proj.bias": "attention.wk.bias",
    "self_attn.v_proj.bias": "attention.wv.bias",
    "mlp.gate_proj.weight": "feed_forward.w1.weight", "mlp.up_proj.weight": "feed_forward.w3.weight",
    "mlp.down_proj.weight": "feed_forward.w2.weight", "input_layernorm.weight": "attention_norm.weight",
    "post_attention_layernorm.weight": "ffn_norm.weight",
}
_TOP = {"model.embed_tokens.weight": "tok_embeddings.weight", "model.norm.weight": "norm.weight",
        "lm_head.weight": "output.weight"}


def _iter_hf_tensors(ckpt_dir: Path):
    """Yields (name, tensor) from sharded/unsharded safetensors or pytorch_model*.bin files, one shard at a time."""
    files = []
    for index in ("model.safetensors.index.json", "pytorch_model.bin.index.json"):
        if (ckpt_dir / index).is_file():
            files = sorted({ckpt_dir / f for f in json.loads((ckpt_dir / index).read_text())["weight_map"].values()})
            break
    if not files:
        files = [f for f in (ckpt_dir / "model.safetensors", ckpt_dir / "pytorch_model.bin") if f.is_file()]
    if not files:
        raise FileNotFoundError(f"no HF weights (safetensors / bin, sharded or not) under {ckpt_dir}")
    for f in files:
        if f.suffix == ".safetensors":
            from safetensors import safe_open
            with safe_open(str(f), framework="pt", device="cpu") as sf:
                for k in sf.keys():
                    yield k, sf.get_tensor(k)
        else:
            for k, v in torch.load(str(f), map_location="cpu", mmap=True, weights_only=True).items():
                yield k, v


def interleave_rope_rows(w: torch.Tensor, n_head: int, head_dim: int) -> torch.Tensor:
    """[n_head*head_dim, ...] rows in HF half-split order -> interleaved-pair order (weights and biases)."""
    rest = w.shape[1:]
    return w.reshape(n_head, 2, head_dim // 2, *rest).transpose(1, 2).reshape(n_head * head_dim, *rest)


@torch.inference_mode()
def convert_hf_checkpoint(checkpoint_dir: Path, model_name: str | None = None, out_file: Path | None = None) -> Path:
    checkpoint_dir = Path(checkpoint_dir)
    cfg = ModelArgs.from_name(model_name or checkpoint_dir.name)
    out: dict[str, torch.Tensor] = {}
    for name, t in _iter_hf_tensors(checkpoint_dir):
        m = _LAYER.match(name)
        if m:
            idx, rest = m.group(1), m.group(2)
            if rest.endswith("rotary_emb.inv_freq"):
                continue
            if rest not in _PER_LAYER:
                raise KeyError(f"unexpected tensor {name}")
            out[f"layers.{idx}.{_PER_LAYER[rest]}"] = t
        elif name in _TOP:
            out[_TOP[name]] = t
        else:
            raise KeyError(f"unexpected tensor {name}")
    if "output.weight" not in out:
        out["output.weight"] = out["tok_embeddings.weight"]
        print("tied lm head: output.weight = tok_embeddings.weight")
    for i in range(cfg.n_layer):
        p = f"layers.{i}.attention."
        for kind in ("weight", "bias"):
            if p + "wq." + kind not in out:
                continue
            q = interleave_rope_rows(out.pop(p + "wq." + kind), cfg.n_head, cfg.head_dim)
            k = interleave_rope_rows(out.pop(p + "wk." + kind), cfg.n_local_heads, cfg.head_dim)
            out[p + "wqkv." + kind] = torch.cat([q, k, out.pop(p + "wv." + kind)])
    out_file = Path(out_file) if out_file else checkpoint_dir / "model.pth"
    torch.save(out, out_file)
    print(f"saved {len(out)} tensors to {out_file}")
    return out_file


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="Convert a HuggingFace checkpoint to model.pth")
    ap.add_argument("--checkpoint_dir", type=Path, required=True)
    ap.add_argument("--model_name", type=str, default=None)
    a = ap.parse_args()
    convert_hf_checkpoint(a.checkpoint_dir, a.model_name)
