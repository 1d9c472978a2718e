"""Stand-alone timing of md_paged_attn at one layer of the north-star verify shape (used for rocprofv3 PMC passes
and kernel tuning A/B).  python tools/attn_bench.py [--B 64 --S 16036 --KH 8 --H 32 --D 128 --n 4 --iters 20]"""
import argparse, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magicdec_amd import ops

ap = argparse.ArgumentParser()
for k, v in dict(B=64, S=16036, KH=8, H=32, D=128, n=4, iters=20, layers=2, wgs=0, fp8=0, hnd=0, kt=0, nw=0, mfma32=-1, reps=1, zero=0, dwaves=0).items():
    ap.add_argument(f"--{k}", type=int, default=v)
ap.add_argument("--variants", default="", help="comma list of md_debug_set_prefill_mfma32 values timed in ONE process on the "
                "same tensors (x --reps), each checked against the first one's output")
a = ap.parse_args()
if a.wgs:
    import ctypes
    from magicdec_amd import _lib
    _lib.load().md_debug_set_attn_target_wgs(ctypes.c_int(a.wgs))
if a.dwaves:                                   # decode / verify kernel: wavefronts per workgroup (4 | 8 forced)
    import ctypes
    from magicdec_amd import _lib
    _lib.load().md_debug_set_attn_waves(ctypes.c_int(a.dwaves))
if a.kt or a.nw:                               # prefill kernel: keys per shared tile / waves per workgroup
    import ctypes
    from magicdec_amd import _lib
    _lib.load().md_debug_set_prefill_kt(ctypes.c_int(a.kt or 64), ctypes.c_int(a.nw))
if a.mfma32 != -1:                             # prefill kernel choice (0 = 16x16x32, 32 | 64 | 128 keys, 1000+ ping-pong)
    import ctypes
    from magicdec_amd import _lib
    _lib.load().md_debug_set_prefill_mfma32(ctypes.c_int(a.mfma32))
dev = "cuda"
mp = (a.S + 127) // 128
g = torch.Generator(device=dev).manual_seed(0)
caches = [torch.randn(a.B * mp, 2, 128, a.KH, a.D, device=dev, generator=g, dtype=torch.float32).to(torch.bfloat16)
          for _ in range(a.layers)]           # > 256 MiB each: defeats the Infinity Cache between launches
scales = None
if a.fp8:
    caches = [c.to(torch.float8_e4m3fn) for c in caches]
    scales = (torch.full((a.KH,), 0.5, device=dev), torch.full((a.KH,), 0.25, device=dev))
layout = "HND" if a.hnd else "NHD"
if a.hnd:                                     # same values, rows of one kv head contiguous inside a page
    caches = [c.permute(0, 1, 3, 2, 4).contiguous() for c in caches]
q = torch.randn(a.B * a.n, a.H, a.D, device=dev, generator=g, dtype=torch.float32).to(torch.bfloat16)
indices = torch.arange(a.B * mp, dtype=torch.int32, device=dev)
indptr = torch.arange(a.B + 1, dtype=torch.int32, device=dev) * mp
last = torch.full((a.B,), a.S - (mp - 1) * 128, dtype=torch.int32, device=dev)
qo = torch.arange(a.B + 1, dtype=torch.int32, device=dev) * a.n
if a.zero:                                     # DVFS check: same instruction stream, no toggling data
    for c in caches:
        c.zero_()
    q.zero_()
ws = ops.AttnWorkspace(dev)
nbytes = a.B * a.S * a.KH * a.D * 2 * (1 if a.fp8 else 2) + 2 * a.B * a.n * a.H * a.D * 2
flops = 4.0 * a.B * a.n * a.H * a.D * (a.S - a.n / 2.0)          # causal: row i of the chunk sees S - n + i + 1 keys


def run_once(tag):
    for i in range(3):
        out = ops.paged_attention(q, caches[i % a.layers], qo, indices, indptr, last, a.n, mp, ws, kv_scales=scales,
                                  kv_layout=layout)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(a.iters):
        ops.paged_attention(q, caches[i % a.layers], qo, indices, indptr, last, a.n, mp, ws, kv_scales=scales,
                            kv_layout=layout)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    if a.n >= 32:
        print(f"  prefill view (kt={a.kt or 64} nw={a.nw or 'auto'} mfma32={tag}): {flops / ms / 1e9:.1f} TFLOP/s = "
              f"{flops / ms / 1e9 / 25:.2f}% of 2.5 PFLOP/s dense bf16")
    print(f"md_paged_attn B={a.B} S={a.S} KH={a.KH} H={a.H} D={a.D} n={a.n}: {ms:.4f} ms  {nbytes / ms / 1e6:.1f} GB/s  "
          f"{nbytes / ms / 1e6 / 80:.2f}% of 8 TB/s  (alg bytes {nbytes}) wgs={a.wgs} fp8={a.fp8} layout={layout} "
          f"map={os.environ.get('MD_ATTN_MAP', '0')} decode_waves={a.dwaves or 'rule'}", flush=True)
    return out


if a.variants:
    import ctypes
    from magicdec_amd import _lib
    first = None
    for rep in range(a.reps):
        for v in [int(x) for x in a.variants.split(",")]:
            _lib.load().md_debug_set_prefill_mfma32(ctypes.c_int(v))
            out = run_once(v).float()
            if first is None:
                first = out
            else:
                d = (out - first).abs()
                print(f"    vs the first variant: max |diff| {d.max().item():.3e}  mean {d.mean().item():.3e}  "
                      f"nan {int(torch.isnan(out).sum())}", flush=True)
else:
    for rep in range(a.reps):
        run_once(a.mfma32)
