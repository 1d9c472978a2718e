"""A/B of md_linear (hand-written weight-streaming skinny GEMM) against hipBLASLt (torch F.linear, TunableOp table
loaded) on the decode / verify shapes, cycling through enough distinct weight copies to defeat the 256 MiB Infinity
Cache.  python tools/gemm_bench.py [--blocks 512] [--only 8B] -> one line per shape, both timings kept."""
import argparse
import ctypes
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magicdec_amd import _lib, ops                     # noqa: E402
from magicdec_amd.Engine.utils import enable_tuned_gemms   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--blocks", type=int, nargs="+", default=[256])
ap.add_argument("--only", default="")
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--waves", type=int, nargs="+", default=[0],
                help="md_linear wavefronts per workgroup: 0 = the balance rule (plan_of), 4 / 6 / 7 forced (bit-identical)")
a = ap.parse_args()
print('tuned GEMM table loaded:', enable_tuned_gemms())
dev = "cuda"
SHAPES = [("1B wqkv", 64, 3072, 2048, 0), ("1B wo", 64, 2048, 2048, 0), ("1B w13", 64, 16384, 2048, 1),
          ("1B w2", 64, 2048, 8192, 0), ("1B head", 64, 128256, 2048, 0),
          ("8B wqkv v", 256, 6144, 4096, 0), ("8B wo v", 256, 4096, 4096, 0), ("8B w13 v", 256, 28672, 4096, 1),
          ("8B w2 v", 256, 4096, 14336, 0), ("8B head v", 256, 128256, 4096, 0),
          ("8B wqkv ar", 64, 6144, 4096, 0), ("8B wo ar", 64, 4096, 4096, 0), ("8B w13 ar", 64, 28672, 4096, 1),
          ("8B w2 ar", 64, 4096, 14336, 0), ("8B head ar", 64, 128256, 4096, 0),
          ("8B wqkv c2", 32, 6144, 4096, 0), ("8B wo c2", 32, 4096, 4096, 0), ("8B w13 c2", 32, 28672, 4096, 1),
          ("8B w2 c2", 32, 4096, 14336, 0),
          ("8B wqkv c2v", 128, 6144, 4096, 0), ("8B w13 c2v", 128, 28672, 4096, 1), ("8B w2 c2v", 128, 4096, 14336, 0),
          ("8B/8 wqkv ar", 64, 768, 4096, 0), ("8B/8 wo ar", 64, 4096, 512, 0), ("8B/8 w13 ar", 64, 3584, 4096, 1),
          ("8B/8 w2 ar", 64, 4096, 1792, 0), ("1B/4 wqkv", 64, 768, 2048, 0), ("1B/4 wo", 64, 2048, 512, 0),
          ("1B/4 w13", 64, 4096, 2048, 1), ("1B/4 w2", 64, 2048, 2048, 0),
          ("8B/8 wqkv v", 256, 768, 4096, 0), ("8B/8 wo v", 256, 4096, 512, 0), ("8B/8 w13 v", 256, 3584, 4096, 1),
          ("8B/8 w2 v", 256, 4096, 1792, 0)]


def timeit(fn, n):
    """Device time per call with the calls captured into ONE hipGraph (what the engine does with a decode step): eager
    launching would measure the host (~10-15 us per call through ctypes + two launches) for the small shapes."""
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for i in range(n):
                fn(i)
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3


lib = _lib.load()
ws = ops.AttnWorkspace(dev)
print(f"{'shape':14s} {'M':>4s} {'N':>6s} {'K':>6s} | hipBLASLt us  TB/s |" + "".join(f" md_linear@{b}/nw{nw}: row-major us TB/s / packed us TB/s |" for b in a.blocks for nw in a.waves))
for name, M, N, K, swiglu in SHAPES:
    if a.only and a.only not in name:
        continue
    nbytes = N * K * 2
    ncopy = max(2, int(600e6 // nbytes) + 1)
    wlist = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(ncopy)]
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    I = N // 2

    def ref(i):
        h = F.linear(x, wlist[i % ncopy])
        return ops.silu_mul(h[:, :I], h[:, I:]) if swiglu else h
    t_ref = timeit(ref, a.iters)
    line = f"{name:14s} {M:4d} {N:6d} {K:6d} | {t_ref:9.1f} {nbytes / t_ref / 1e6:6.2f} |"
    plist = [ops.PackedWeight(w, swiglu=bool(swiglu)) for w in wlist]
    for b in a.blocks:
        for nw in a.waves:
            lib.md_debug_set_gemm_target_blocks(ctypes.c_int(b))
            lib.md_debug_set_gemm_waves(ctypes.c_int(nw))
            t = timeit(lambda i: ops.linear(x, wlist[i % ncopy], swiglu=bool(swiglu), workspace=ws), a.iters)
            tp = timeit(lambda i: ops.linear(x, plist[i % ncopy], swiglu=bool(swiglu), workspace=ws), a.iters)
            line += f" {t:7.1f} {nbytes / t / 1e6:5.2f} / packed {tp:7.1f} {nbytes / tp / 1e6:5.2f} |"
    lib.md_debug_set_gemm_waves(ctypes.c_int(0))
    del plist
    print(line, flush=True)
    del wlist
