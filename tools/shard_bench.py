"""Tensor-parallel shard linears of a BASELINE model at the row counts of its configuration: library GEMM against md_linear
and md_linear_fused (default rule / 1 x 1 / 2 x 2 tiles), graph-captured, weights cycled through > 600 MB.
    python tools/shard_bench.py [--model 70b|qwen32b|8b|1b] [--tp 8] [--rows 32,128]
(default: the 70B model's TP-8 shards at the rows of configs[3]: 32-row autoregressive steps, 128-row verify)"""
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
ap.add_argument("--model", default="70b")
ap.add_argument("--tp", type=int, default=8)
ap.add_argument("--rows", default="32,128")
a = ap.parse_args()
print("tuned GEMM table loaded:", enable_tuned_gemms())
dev = "cuda"
lib = _lib.load()
ws = ops.AttnWorkspace(dev)
# name: dim, heads, kv heads, head dim, FFN width
MODELS = {"70b": (8192, 64, 8, 128, 28672), "qwen32b": (5120, 40, 8, 128, 27648), "8b": (4096, 32, 8, 128, 14336),
          "1b": (2048, 32, 8, 64, 8192)}
dim, H, KH, D, I = MODELS[a.model]
h, kh, i = H // a.tp, max(KH // a.tp, 1), I // a.tp
SHAPES = [("wqkv", (h + 2 * kh) * D, dim, False), ("wo", dim, h * D, False), ("w13", 2 * i, dim, True), ("w2", dim, i, False)]
print(f"model {a.model} / {a.tp}: " + ", ".join(f"{n} {N}x{K}" for n, N, K, _ in SHAPES))


def timeit(fn, n=30):
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


print(f"{'linear':6s} {'M':>4s} {'N':>6s} {'K':>6s} {'MB':>6s} | {'lib':>7s} | {'skinny':>7s} | {'fused':>7s} | {'1x1':>7s} | {'2x2':>7s}")
for M in [int(r) for r in a.rows.split(",")]:
    for name, N, K, sw in SHAPES:
        nbytes = N * K * 2
        ncopy = max(2, int(600e6 // nbytes) + 1)
        wl = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(ncopy)]
        pk = [ops.PackedWeight(w, swiglu=sw) for w in wl]
        x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        I = N // 2

        def lib_fn(i):
            h = F.linear(x, wl[i % ncopy])
            return ops.silu_mul(h[:, :I], h[:, I:]) if sw else h

        def skinny_fn(i):
            return ops.linear(x, pk[i % ncopy], swiglu=sw, workspace=ws)

        def fused_fn(i):
            return ops.fused_linear(x, pk[i % ncopy], swiglu=sw)
        fused_ok = ops.fused_linear_supported(M, N, K)
        t = [timeit(lib_fn), timeit(skinny_fn) if ops.linear_supported(M, N, K, sw) else float("nan"),
             timeit(fused_fn) if fused_ok else float("nan")]
        for knob in (11, 22):
            lib.md_debug_set_fused_nw(ctypes.c_int(knob))
            t.append(timeit(fused_fn) if fused_ok else float("nan"))
        lib.md_debug_set_fused_nw(ctypes.c_int(0))
        print(f"{name:6s} {M:4d} {N:6d} {K:6d} {nbytes / 1e6:6.1f} | " + " | ".join(f"{v:7.1f}" for v in t), flush=True)
        del wl, pk
