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


def _random_init_(model, seed, device, dtype, std=0.02, is_draft=False):
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
        emb_rms, peak, miss = parse_peaked(mode)
        _peak_(model, seed, dtype, emb_rms, peak, miss if is_draft else 0.0)
    elif mode != "random":
        raise ValueError(f"MAGICDEC_SYNTH_WEIGHTS must be 'random' or 'peaked[:emb_rms[:peak]][:miss=f]', got {mode!r}")


def parse_peaked(mode):
    """'peaked[:emb_rms[:peak]][:miss=f]' -> (emb_rms, peak, miss)."""
    emb_rms, peak, miss = 40.0, 12.0, 0.0
    pos = []
    for part in mode.split(":")[1:]:
        if part.startswith("miss="):
            miss = float(part[5:])
        elif part:
            pos.append(float(part))
    if len(pos) > 2 or not 0.0 <= miss < 1.0:
        raise ValueError(f"MAGICDEC_SYNTH_WEIGHTS must be 'random' or 'peaked[:emb_rms[:peak]][:miss=f]', got {mode!r}")
    if pos:
        emb_rms = pos[0]
    if len(pos) > 1:
        peak = pos[1]
    return emb_rms, peak, miss


def _peak_perm(seed, V, miss):
    """pi of _peak_: ids 0..3 fixed, the rest a seeded permutation that depends on (seed, V) only; with `miss` = f a
    seeded fraction f of the OUTPUT rows (ids >= 4) is rotated among themselves -- row j of the miss set answers to the
    token whose true successor is the PREVIOUS member of the set, i.e. the model confidently predicts a wrong token
    whenever the true next token lies in the set (the `miss_every` construction of tests/test_gpu_engine._peaked_wide
    with a random instead of a strided set)."""
    g = torch.Generator(device="cpu").manual_seed(seed * 7919 + V)
    perm = torch.arange(V)
    perm[4:] = 4 + torch.randperm(V - 4, generator=g)          # ids 0..3 (BOS / EOT ids of the tests) stay fixed
    if miss > 0.0:
        gm = torch.Generator(device="cpu").manual_seed(seed * 104729 + V + 17)
        sel = torch.nonzero(torch.rand(V - 4, generator=gm) < miss).view(-1) + 4
        if sel.numel() >= 2:
            perm = perm.clone()
            perm[sel] = perm[torch.roll(sel, 1)]
    return perm


def _peak_(model, seed, dtype, emb_rms=40.0, peak=12.0, miss=0.0):
    """Seeded weights with PEAKED next-token distributions (MAGICDEC_SYNTH_WEIGHTS=peaked; bench.py --weights peaked):
    the construction of tests/golden_cfg.py:peaked_pair at any size.  Random-init heads give `vocab` nearly tied
    Gaussian logits, so a draft never agrees with its target and measured acceptance is ~0 whatever the kernels do.
    Here the embedding dominates the residual stream (`emb_rms` per element against the ~1.5 sqrt(n_layer) the random
    layers add) and the head is tied to it through a fixed permutation pi of the token ids that depends on (seed,
    vocab) only -- `output[j] = c emb[pi(j)]`, logit ~`peak` for the one j with pi(j) == current token, the others
    ~N(0, peak^2 / dim) -- so that every model of a run with the same vocabulary (target, draft, self-speculation's
    sparse-cache draft) predicts through the same map and acceptance is decided by what the kernels compute: the
    attention / FFN branches of all layers still run at full strength as the perturbation.
    `miss` (draft models only; 'peaked:...:miss=f'): the head mispredicts whenever the true next token lies in a seeded
    fraction f of the vocabulary (_peak_perm), so a draft step is rejected with probability ~f -- a draft of KNOWN
    acceptance rate alpha ~ 1 - f, measured by the accept kernel's own decisions."""
    emb = model.tok_embeddings.weight
    V, dim = emb.shape
    e = (emb.data.float() * (emb_rms / 0.02)).to(dtype)
    model.tok_embeddings.weight = torch.nn.Parameter(e, requires_grad=False)
    model._peak_params = (seed, emb_rms, peak)
    model.output.weight = torch.nn.Parameter(_peaked_head_rows(e, seed, emb_rms, peak, miss, 0, V).to(dtype),
                                             requires_grad=False)


def _peaked_head_rows(e, seed, emb_rms, peak, miss, row0, rows):
    V, dim = e.shape
    perm = _peak_perm(seed, V, miss)[row0:row0 + rows].to(e.device)
    c = peak / (dim * emb_rms)                                  # tied row: c |e|^2 / rms(h) ~ peak
    return e[perm].float() * c


def repeak_head_(model, miss):
    """Rewrite the lm head of a model built by _peak_ for another `miss` fraction IN PLACE (same storage: captured
    hipGraphs and the streaming-layout copy keep their addresses) -- bench.py sweeps a draft's acceptance rate without
    reloading or re-prefilling anything (the SnapKV / StreamingLLM draft cache does not depend on the head).  Under
    tensor parallelism the head is vocab-chunked (Engine/tp.py): this rank's rows are regenerated."""
    seed, emb_rms, peak = model._peak_params
    e = model.tok_embeddings.weight.data
    w = model.output.weight
    rows = w.shape[0]
    rank = getattr(model, "rank", None) or 0
    world = getattr(model, "world_size", None) or 1
    row0 = rank * rows if world > 1 else 0
    new = _peaked_head_rows(e, seed, emb_rms, peak, miss, row0, rows).to(w.dtype)
    released = id(w) in getattr(model, "_released", ())      # row-major tensor released after prefill: one resident copy
    if not released:
        w.data.copy_(new)
    pk = getattr(model, "_packed", {}).get(id(w))
    if pk is not None:
        from .. import ops
        pk.data.copy_(ops.PackedWeight(new).data)
    elif released:
        raise RuntimeError("repeak_head_: the head's row-major tensor is released and it has no streaming copy")


def _load(transformer_cls, checkpoint_path, device, precision, use_tp, rank_group, group, seed=1234, is_draft=False):
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
        _random_init_(model, seed, device, precision, is_draft=is_draft)
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
    return _load(Transformer, checkpoint_path, device, precision, use_tp, rank_group, group, is_draft=True)


def load_model_streamingLLM(checkpoint_path, device, precision, use_tp, rank_group=None, group=None):
    from .StreamingLLM.model import Transformer
    return _load(Transformer, checkpoint_path, device, precision, use_tp, rank_group, group)


def load_model_draft_streamingLLM(checkpoint_path, device, precision, use_tp, rank_group=None, group=None):
    from .StreamingLLM.model_draft import Transformer
    return _load(Transformer, checkpoint_path, device, precision, use_tp, rank_group, group, is_draft=True)
