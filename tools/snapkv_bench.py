"""Timing of md_snapkv_select (Attention.gen_draft_kv, Engine/SnapKV/model.py:389-439) at one layer of a BASELINE shape.
python tools/snapkv_bench.py [--B 64 --KH 8 --g 4 --D 64 --S 16032 --fp8 0 --hnd 1 --dist normal|engine|heavy|all]
Algorithmic bytes: the K half of the context read once, B * S * KH * D * sizeof.

--dist: the distribution of the UNSCALED scores q.k the select's softmax sees (round 4's number held
on the first one only).  normal: q ~ 0.3 N(0,1), k ~ N(0,1) -> scores ~ N(0, 2.4^2) (round 4's microbenchmark);
engine: q, k ~ 0.02 sqrt(dim) N(0,1) = what layer 0 of bench.py's seeded-random 1B draft produces (RMSNorm'd
activations through normal(0, 0.02) projections, dim 2048) -> scores ~ N(0, 6.5^2), 1.4 % beyond |s| = 16;
heavy: scores ~ N(0, 30^2), |s| up to ~130 as a trained checkpoint's unscaled q.k produce."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magicdec_amd import ops                           # noqa: E402

ap = argparse.ArgumentParser()
for k, v in dict(B=64, KH=8, g=4, D=64, S=16032, W=32, budget=257, fp8=0, hnd=1, iters=5).items():
    ap.add_argument(f"--{k}", type=int, default=v)
ap.add_argument("--dist", default="all", choices=["normal", "engine", "heavy", "all"])
a = ap.parse_args()
DISTS = {"normal": (0.3, 1.0), "engine": (0.02 * 2048 ** 0.5, 0.02 * 2048 ** 0.5), "heavy": (1.94, 1.94)}
dev = "cuda"
H = a.KH * a.g
npg = (a.S + 127) // 128
dppr = a.budget // 128 + 1
indices = torch.arange(a.B * npg, dtype=torch.int32, device=dev)
indptr = (torch.arange(a.B + 1, dtype=torch.int32) * npg).to(dev)
dind = torch.arange(a.B * dppr, dtype=torch.int32, device=dev)
dptr = (torch.arange(a.B + 1, dtype=torch.int32) * dppr).to(dev)
dlast = torch.ones(a.B, dtype=torch.int32, device=dev)
dcache = torch.zeros(a.B * dppr, 2, 128, a.KH, a.D, dtype=torch.bfloat16, device=dev)
ws = ops.AttnWorkspace(dev)
layout = "HND" if a.hnd else "NHD"
nbytes = a.B * a.S * a.KH * a.D * (1 if a.fp8 else 2)

for dist in (list(DISTS) if a.dist == "all" else [a.dist]):
    qs, ks = DISTS[dist]
    gen = torch.Generator(device=dev).manual_seed(0)
    cache = torch.randn(a.B * npg, 2, 128, a.KH, a.D, device=dev, generator=gen, dtype=torch.float32) * ks
    scales = None
    if a.fp8:
        cache = (cache * 64).clamp_(-448, 448).to(torch.float8_e4m3fn)
        scales = (torch.full((a.KH,), 1 / 64.0, device=dev), torch.full((a.KH,), 1 / 32.0, device=dev))
    else:
        cache = cache.to(torch.bfloat16)
    if a.hnd:
        cache = cache.permute(0, 1, 3, 2, 4).contiguous()
    q = (torch.randn(a.B * a.W, H, a.D, device=dev, generator=gen, dtype=torch.float32) * qs).to(torch.bfloat16)

    def run():
        return ops.snapkv_select(q, cache, indices, indptr, a.S, a.W, a.budget, 5, dcache, dind, dptr, dlast, ws,
                                 kv_scales=scales, kv_layout=layout)

    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    # what the softmax sees: unscaled scores of request 0, kv head 0 against the first 2048 keys
    k0 = (cache[:16, 0, 0] if a.hnd else cache[:16, 0, :, 0]).reshape(-1, a.D).float()
    if a.fp8:
        k0 = k0 * scales[0][0]
    sc = q[:a.W, 0].float() @ k0.T
    print(f"md_snapkv_select dist={dist:6s} (scores: std {sc.std().item():.1f}, max |s| {sc.abs().max().item():.0f}, "
          f"{100 * (sc.abs() >= 16).float().mean().item():.2f} % beyond 16) B={a.B} KH={a.KH} g={a.g} D={a.D} S={a.S} "
          f"fp8={a.fp8} {layout}: {ms:.3f} ms per layer = {nbytes / ms / 1e9:.3f} TB/s of K bytes ({nbytes / 1e9:.2f} GB) = "
          f"{100 * nbytes / ms / 1e9 / 8:.1f} % of 8 TB/s", flush=True)
    del cache, q
