"""Same-box A/B of the two KV page layouts for the verify attention launch (md_paged_attn, one layer, Infinity-Cache
cold): NHD = the reference's flashinfer layout [pages, 2, 128, KH, D], HND = [pages, 2, KH, 128, D].  One process, so
the torch import and the cache fill are paid once.

    python tools/layout_ab.py [--B 64 --S 16075 --H 32 --D 128 --n 4 --iters 20 --KH 8,2]

Prints one line per (KH, dtype, layout): ms per launch, GB/s of algorithmic bytes, fraction of the 8 TB/s HBM peak, and
checks that the two layouts give bit-identical outputs."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magicdec_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
for k, v in dict(B=64, S=16075, H=32, D=128, n=4, iters=20, layers=2).items():
    ap.add_argument(f"--{k}", type=int, default=v)
ap.add_argument("--KH", default="8,2")
a = ap.parse_args()
dev = "cuda"
mp = (a.S + 127) // 128
g = torch.Generator(device=dev).manual_seed(0)
indices = torch.arange(a.B * mp, dtype=torch.int32, device=dev)
indptr = torch.arange(a.B + 1, dtype=torch.int32, device=dev) * mp
last = torch.full((a.B,), a.S - (mp - 1) * 128, dtype=torch.int32, device=dev)
qo = torch.arange(a.B + 1, dtype=torch.int32, device=dev) * a.n
ws = ops.AttnWorkspace(dev)


def timed(q, caches, scales, layout):
    run = lambda i: ops.paged_attention(q, caches[i % len(caches)], qo, indices, indptr, last, a.n, mp, ws,
                                        kv_scales=scales, kv_layout=layout)
    for i in range(3):
        out = run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(a.iters):
        run(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters, run(0)


for KH in [int(x) for x in a.KH.split(",")]:
    Hq = a.H * KH // 8 if KH < 8 else a.H                 # keep g = H/KH of the 8-head model for the shard shapes
    q = torch.randn(a.B * a.n, Hq, a.D, device=dev, generator=g, dtype=torch.float32).to(torch.bfloat16)
    base = [torch.randn(a.B * mp, 2, 128, KH, a.D, device=dev, generator=g, dtype=torch.bfloat16) for _ in range(a.layers)]
    for fp8 in (False, True):
        nhd = [c.to(torch.float8_e4m3fn) for c in base] if fp8 else base
        scales = ((torch.full((KH,), 0.5, device=dev), torch.full((KH,), 0.25, device=dev)) if fp8 else None)
        nbytes = a.B * a.S * KH * a.D * 2 * (1 if fp8 else 2) + 2 * a.B * a.n * Hq * a.D * 2
        outs = {}
        for layout in ("NHD", "HND"):
            caches = [c.permute(0, 1, 3, 2, 4).contiguous() for c in nhd] if layout == "HND" else nhd
            ms, outs[layout] = timed(q, caches, scales, layout)
            print(f"md_paged_attn B={a.B} S={a.S} KH={KH} H={Hq} D={a.D} n={a.n} {'fp8 ' if fp8 else 'bf16'} {layout}: "
                  f"{ms:.4f} ms  {nbytes / ms / 1e6:8.1f} GB/s  {nbytes / ms / 1e6 / 80:5.2f}% of 8 TB/s", flush=True)
            del caches
        same = torch.equal(outs["NHD"].view(torch.int16), outs["HND"].view(torch.int16))
        print(f"    outputs bit-identical between layouts: {same}", flush=True)
        assert same
    del base
    torch.cuda.empty_cache()
