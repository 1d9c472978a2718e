"""The 70B model's TP-8 shard linears at the row counts of configs[3] (B = 32: 32-row autoregressive / draft-free steps, 128-row
verify): library GEMM against md_linear and md_linear_fused (1 x 1 / 2 x 2 tiles), graph-captured, weights cycled.
    python tools/shard70b_bench.py"""
import ctypes
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magicdec_amd import _lib, ops                          # noqa: E402
from magicdec_amd.Engine.utils import enable_tuned_gemms   # noqa: E402

print("tuned GEMM table loaded:", enable_tuned_gemms())
dev = "cuda"
lib = _lib.load()
ws = ops.AttnWorkspace(dev)
# 70B / 8: dim 8192, 8 q heads + 1 kv head per rank (D = 128), FFN 28672 / 8 = 3584
SHAPES = [("wqkv", 1280, 8192, False), ("wo", 8192, 1024, False), ("w13", 7168, 8192, True), ("w2", 8192, 3584, False)]


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
for M in (32, 128):
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
        t = [timeit(lib_fn), timeit(skinny_fn) if ops.linear_supported(M, N, K, sw) else float("nan"), timeit(fused_fn)]
        for knob in (11, 22):
            lib.md_debug_set_fused_nw(ctypes.c_int(knob))
            t.append(timeit(fused_fn))
        lib.md_debug_set_fused_nw(ctypes.c_int(0))
        print(f"{name:6s} {M:4d} {N:6d} {K:6d} {nbytes / 1e6:6.1f} | " + " | ".join(f"{v:7.1f}" for v in t), flush=True)
        del wl, pk
