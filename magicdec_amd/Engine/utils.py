"""Loaders and seeding of the decode path (Engine/utils.py:189-277 of the reference, same names).

`model.pth` files written by the reference's convert_hf_checkpoint.py load unchanged (same parameter names,
fused wqkv = [q;k;v], interleaved-RoPE row order).  When the checkpoint file does not exist the model is
initialised with seeded normal(0, 0.02) bf16 weights of the architecture named by the checkpoint's parent
directory -- benchmarks on a box without network access use this (throughput is weight-value independent).
"""
import random
from pathlib import Path

import numpy as np
import torch


TUNED_GEMMS = Path(__file__).resolve().parent.parent / "tuned" / "gemm_gfx950.csv"


def enable_tuned_gemms(path=None):
    """Select the hipBLASLt / rocBLAS solutions recorded by tools/tune_gemms.py (PyTorch TunableOp, look-up only --
    no tuning happens in the serving path).  hipBLASLt's default heuristic picks un-split tiles for the skinny
    (M = 64..256) small-N projections of the decode step; the tuned table is 20-35 % faster on those.  Returns True
    when the table was loaded (validators = library versions must match, else PyTorch ignores the file)."""
    path = Path(path) if path is not None else TUNED_GEMMS
    if not path.exists() or not torch.cuda.is_available():
        return False
    import tempfile

    import torch.cuda.tunable as tunable
    tunable.enable(True)
    tunable.tuning_enable(False)
    tunable.set_filename(str(Path(tempfile.gettempdir()) / "magicdec_tunableop_unused.csv"))   # never rewrite the table
    return bool(tunable.read_file(str(path)))


def setup_seed(seed):
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


def _random_init_(model, seed, device, dtype, std=0.02):
    """Deterministic per-parameter init (each tensor from its own generator stream: identical on every rank)."""
    for j, (name, p) in enumerate(sorted(model.named_parameters())):
        g = torch.Generator(device="cpu").manual_seed(seed * 100003 + j)
        if p.dim() == 1 and "norm" in name:
            t = torch.ones(p.shape, dtype=dtype)
        elif p.numel() >= (1 << 20) and torch.device(device).type == "cuda":      # Philox on the GPU: same bits on every rank
            gg = torch.Generator(device=device).manual_seed(seed * 100003 + j)
            t = (torch.randn(p.shape, generator=gg, device=device, dtype=torch.float32) * std).to(dtype)
        else:
            t = (torch.randn(p.shape, generator=g) * std).to(dtype)
        parent, leaf = model, name.split(".")
        for part in leaf[:-1]:
            parent = getattr(parent, part)
        setattr(parent, leaf[-1], torch.nn.Parameter(t, requires_grad=False))
    import os
    mode = os.environ.get("MAGICDEC_SYNTH_WEIGHTS", "random")
    if mode.startswith("peaked"):
        _peak_(model, seed, dtype, *[float(x) for x in mode.split(":")[1:3]])
    elif mode != "random":
        raise ValueError(f"MAGICDEC_SYNTH_WEIGHTS must be 'random' or 'peaked[:emb_rms[:peak]]', got {mode!r}")


def _peak_(model, seed, dtype, emb_rms=40.0, peak=12.0):
    """Seeded weights with PEAKED next-token distributions (MAGICDEC_SYNTH_WEIGHTS=peaked; bench.py --weights peaked):
    the construction of tests/golden_cfg.py:peaked_pair at any size.  Random-init heads give `vocab` nearly tied
    Gaussian logits, so a draft never agrees with its target and measured acceptance is ~0 whatever the kernels do.
    Here the embedding dominates the residual stream (`emb_rms` per element against the ~1.5 sqrt(n_layer) the random
    layers add) and the head is tied to it through a fixed permutation pi of the token ids that depends on (seed,
    vocab) only -- `output[j] = c emb[pi(j)]`, logit ~`peak` for the one j with pi(j) == current token, the others
    ~N(0, peak^2 / dim) -- so that every model of a run with the same vocabulary (target, draft, self-speculation's
    sparse-cache draft) predicts through the same map and acceptance is decided by what the kernels compute: the
    attention / FFN branches of all layers still run at full strength as the perturbation."""
    emb = model.tok_embeddings.weight
    V, dim = emb.shape
    g = torch.Generator(device="cpu").manual_seed(seed * 7919 + V)
    perm = torch.arange(V)
    perm[4:] = 4 + torch.randperm(V - 4, generator=g)          # ids 0..3 (BOS / EOT ids of the tests) stay fixed
    e = emb.data.float() * (emb_rms / 0.02)
    c = peak / (dim * emb_rms)                                  # tied row: c |e|^2 / rms(h) ~ peak
    model.tok_embeddings.weight = torch.nn.Parameter(e.to(dtype), requires_grad=False)
    model.output.weight = torch.nn.Parameter((e[perm.to(e.device)] * c).to(dtype), requires_grad=False)


def _load(transformer_cls, checkpoint_path, device, precision, use_tp, rank_group, group, seed=1234):
    checkpoint_path = Path(checkpoint_path)
    with torch.device("meta"):
        model = transformer_cls.from_name(checkpoint_path.parent.name)
    int8 = "int8" in str(checkpoint_path)          # Engine/utils.py:201-205 of the reference
    if int8 and checkpoint_path.exists():
        print("Using int8 weight-only quantization!")
        from .quantize import WeightOnlyInt8QuantHandler
        model = WeightOnlyInt8QuantHandler(model).convert_for_runtime()
    if checkpoint_path.exists():
        checkpoint = torch.load(str(checkpoint_path), mmap=True, weights_only=True)
        if "model" in checkpoint and "stories" in str(checkpoint_path):
            checkpoint = checkpoint["model"]
        model.load_state_dict(checkpoint, assign=True)
    else:
        print(f"[magicdec_amd] {checkpoint_path} not found: seeded random weights for '{checkpoint_path.parent.name}'")
        _random_init_(model, seed, device, precision)
        if int8:                                   # quantise the seeded weights the way the reference's quantize.py does
            print("Using int8 weight-only quantization!")
            from .quantize import WeightOnlyInt8QuantHandler
            handler = WeightOnlyInt8QuantHandler(model)
            sd = handler.create_quantized_state_dict()
            model = handler.convert_for_runtime()
            model.load_state_dict(sd, assign=True)
    if use_tp:
        from .tp import apply_tp
        print("Applying tensor parallel to model ...")
        apply_tp(model, rank_group, group=group)
    model = model.to(device=device, dtype=precision)
    return model.eval()


def load_model_snapKV(checkpoint_path, device, precision, use_tp, rank_group=None, group=None):
    from .SnapKV.model import Transformer
    return _load(Transformer, checkpoint_path, device, precision, use_tp, rank_group, group)


def load_model_draft_snapKV(checkpoint_path, device, precision, use_tp, rank_group=None, group=None):
    from .SnapKV.model_draft import Transformer
    return _load(Transformer, checkpoint_path, device, precision, use_tp, rank_group, group)


def load_model_streamingLLM(checkpoint_path, device, precision, use_tp, rank_group=None, group=None):
    from .StreamingLLM.model import Transformer
    return _load(Transformer, checkpoint_path, device, precision, use_tp, rank_group, group)


def load_model_draft_streamingLLM(checkpoint_path, device, precision, use_tp, rank_group=None, group=None):
    from .StreamingLLM.model_draft import Transformer
    return _load(Transformer, checkpoint_path, device, precision, use_tp, rank_group, group)
