"""A/B of md_linear's W prefetch ring depth (8 against 16 k-steps per wavefront) on its four-wave, <= 64-row form
(csrc/gemm.hip, launch(): DESIGN.md 3.3 -- a CU keeps ~64 KB in flight, four waves x 8 KiB request half of that).
Graph-captured, weights cycled through > 600 MB.      python tools/ring_bench.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magicdec_amd import _lib, ops                          # noqa: E402

dev = "cuda"
lib = _lib.load()
ws = ops.AttnWorkspace(dev)
# name, M, N, K, kind
CASES = [("8B w2 + add + norm", 32, 4096, 14336, "resid"), ("8B w2 + add + norm", 64, 4096, 14336, "resid"),
         ("1B w2 + add + norm", 64, 2048, 8192, "resid"), ("1B lm head", 64, 128256, 2048, "plain"),
         ("8B lm head", 64, 128256, 4096, "plain"), ("8B lm head", 32, 128256, 4096, "plain"),
         ("8B w1|w3 + SiLU*mul (7 waves: unaffected)", 32, 28672, 4096, "swiglu")]


def timeit(fn, n=20):
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


print(f"{'case':44s} {'M':>4s} {'N':>7s} {'K':>6s} {'MB':>7s} | ring 8 us | ring 16 us | bits equal")
for name, M, N, K, kind in CASES:
    nbytes = N * K * 2
    ncopy = max(2, int(600e6 // nbytes) + 1)
    wl = [ops.PackedWeight(torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02, swiglu=(kind == "swiglu"))
          for _ in range(ncopy)]
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    r = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    nw = torch.ones(N, device=dev, dtype=torch.bfloat16)

    def fn(i):
        if kind == "resid":
            return ops.linear_add_rmsnorm(x, wl[i % ncopy], r, nw, 1e-5, workspace=ws)[1]
        return ops.linear(x, wl[i % ncopy], swiglu=(kind == "swiglu"), workspace=ws)
    ts, outs = [], []
    for rd in (8, 16):
        lib.md_debug_set_gemm_ring(ctypes.c_int(rd))
        outs.append(fn(0).clone())
        ts.append(timeit(fn))
    lib.md_debug_set_gemm_ring(ctypes.c_int(0))
    same = torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    print(f"{name:44s} {M:4d} {N:7d} {K:6d} {nbytes / 1e6:7.1f} | {ts[0]:9.1f} | {ts[1]:10.1f} | {same}", flush=True)
    del wl
