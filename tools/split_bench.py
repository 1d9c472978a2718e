"""A/B of md_linear_fused_split_add_rmsnorm (csrc/tilegemm.hip: tile kernel with K also split over workgroups + ONE combine
launch that adds the slices, the residual and normalises) against what a step runs otherwise for an output projection
followed by the residual add + RMSNorm: the library GEMM + md_add_rmsnorm, md_linear_add_rmsnorm (weight-streaming split-K +
the same combine launch), md_linear_fused (residual epilogue) + md_rmsnorm.  Graph-captured, weights cycled through > 600 MB.

    python tools/split_bench.py [--iters 30]
"""
import argparse
import ctypes
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magicdec_amd import _lib, ops                          # noqa: E402
from magicdec_amd.Engine.utils import enable_tuned_gemms   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--only", default="")
a = ap.parse_args()
print("tuned GEMM table loaded:", enable_tuned_gemms())
dev = "cuda"
lib = _lib.load()
ws = ops.AttnWorkspace(dev)

# name, M, N, K
CASES = [("1B w2", 64, 2048, 8192), ("1B w2 two-token", 128, 2048, 8192), ("1B wo", 64, 2048, 2048),
         ("8B wo M32", 32, 4096, 4096), ("8B w2 M32", 32, 4096, 14336), ("8B wo M64", 64, 4096, 4096),
         ("8B w2 M64", 64, 4096, 14336), ("1B/4 w2", 64, 2048, 2048), ("8B/8 w2 M64", 64, 4096, 1792)]


def timeit(fn, n):
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
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * 3) * 1e3


print(f"{'case':18s} {'M':>4s} {'N':>6s} {'K':>6s} {'MB':>6s} | {'lib+addnorm':>11s} | {'skinny+comb':>11s} | {'fused+norm':>10s} | split S=auto / 2 / 4 / 8 / 16")
for name, M, N, K in CASES:
    if a.only and a.only not in name:
        continue
    nbytes = N * K * 2
    ncopy = max(2, int(600e6 // nbytes) + 1)
    wl = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(ncopy)]
    pk = [ops.PackedWeight(w) for w in wl]
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    r = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    nw = torch.ones(N, device=dev, dtype=torch.bfloat16)

    def lib_fn(i):
        return ops.add_rmsnorm(r, F.linear(x, wl[i % ncopy]), nw, 1e-5)

    def skinny_fn(i):
        return ops.linear_add_rmsnorm(x, pk[i % ncopy], r, nw, 1e-5, workspace=ws)

    def fused_fn(i):
        return ops.rmsnorm(ops.fused_linear(x, pk[i % ncopy], resid=r), nw, 1e-5)

    def split_fn(i):
        return ops.fused_split_linear_add_rmsnorm(x, pk[i % ncopy], r, nw, 1e-5, workspace=ws)

    t_l = timeit(lib_fn, a.iters)
    t_s = timeit(skinny_fn, a.iters) if ops.linear_add_rmsnorm_supported(M, N, K) else float("nan")
    t_f = timeit(fused_fn, a.iters) if ops.fused_linear_supported(M, N, K) else float("nan")
    ts = []
    for S in (0, 2, 4, 8, 16):
        if S and (K // 16) % (S * 8):
            ts.append(float("nan"))
            continue
        lib.md_debug_set_fused_split(ctypes.c_int(S))
        ts.append(timeit(split_fn, a.iters))
    lib.md_debug_set_fused_split(ctypes.c_int(0))
    print(f"{name:18s} {M:4d} {N:6d} {K:6d} {nbytes / 1e6:6.1f} | {t_l:11.1f} | {t_s:11.1f} | {t_f:10.1f} | "
          + " / ".join(f"{t:5.1f}" for t in ts), flush=True)
    del wl, pk
