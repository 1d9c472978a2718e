"""A/B of md_linear_block (block-tile GEMM, csrc/blockgemm.hip) against hipBLASLt (F.linear + the kernel behind it) and
md_linear / md_linear_add_rmsnorm (weight-streaming skinny GEMM) on the M = 129..256 verify shapes; every timing is a
hipGraph of `iters` calls cycling through > 600 MB of distinct weight copies (defeats the 256 MiB Infinity Cache).
python tools/block_bench.py [--only 8B] [--blocks 256 192] [--wnt 1 0] -> one line per shape."""
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
ap.add_argument("--only", default="")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--blocks", type=int, nargs="+", default=[256])
ap.add_argument("--wnt", type=int, nargs="+", default=[1], help="bit 0: nt weights; bits 4..: dev flags (16 = no x traffic, 32 = no W traffic, 64 = rotate k)")
ap.add_argument("--check", type=int, default=1)
a = ap.parse_args()
print('tuned GEMM table loaded:', enable_tuned_gemms())
dev = "cuda"
# name, M, N, K, kind: 0 plain, 1 swiglu, 2 resid + norm
SHAPES = [("8B wqkv v", 256, 6144, 4096, 0), ("8B wo v", 256, 4096, 4096, 2), ("8B w13 v", 256, 28672, 4096, 1),
          ("8B w2 v", 256, 4096, 14336, 2), ("8B head v", 256, 128256, 4096, 0),
          ("8B wqkv c2v", 128, 6144, 4096, 0), ("8B wo c2v", 128, 4096, 4096, 2), ("8B w13 c2v", 128, 28672, 4096, 1),
          ("8B w2 c2v", 128, 4096, 14336, 2),
          ("8B/8 wqkv v", 256, 768, 4096, 0), ("8B/8 wo v", 256, 4096, 512, 2), ("8B/8 w13 v", 256, 3584, 4096, 1),
          ("8B/8 w2 v", 256, 4096, 1792, 2),
          ("70B/8 wqkv", 128, 1280, 8192, 0), ("70B/8 w13", 128, 7168, 8192, 1), ("70B/8 w2", 128, 8192, 3584, 2)]


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
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


lib = _lib.load()
ws = ops.AttnWorkspace(dev)
print(f"{'shape':12s} {'M':>4s} {'N':>6s} {'K':>6s} | lib(+epi) us | skinny us |" +
      "".join(f" block@{b}/nt{w} us  TB/s  TF/s |" for b in a.blocks for w in a.wnt))
for name, M, N, K, kind in SHAPES:
    if a.only and a.only not in name:
        continue
    nbytes = N * K * 2
    ncopy = max(2, int(600e6 // nbytes) + 1)
    wlist = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(ncopy)]
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    I = N // 2
    resid = torch.randn(M, N, device=dev, dtype=torch.bfloat16) if kind == 2 else None
    nw = torch.ones(N, device=dev, dtype=torch.bfloat16) if kind == 2 else None

    def ref(i):
        h = F.linear(x, wlist[i % ncopy])
        if kind == 1:
            return ops.silu_mul(h[:, :I], h[:, I:])
        if kind == 2:
            return ops.add_rmsnorm(resid, h, nw, 1e-5)
        return h
    t_ref = timeit(ref, a.iters)
    plist = [ops.PackedWeight(w, swiglu=(kind == 1)) for w in wlist]

    def skinny(i):
        if kind == 2:
            if ops.linear_add_rmsnorm_supported(M, N, K):
                return ops.linear_add_rmsnorm(x, plist[i % ncopy], resid, nw, 1e-5, workspace=ws)
            return ops.add_rmsnorm(resid, ops.linear(x, plist[i % ncopy], workspace=ws), nw, 1e-5)
        return ops.linear(x, plist[i % ncopy], swiglu=(kind == 1), workspace=ws)
    t_sk = timeit(skinny, a.iters)

    def block(i):
        if kind == 2:
            return ops.linear_block_add_rmsnorm(x, plist[i % ncopy], resid, nw, 1e-5, workspace=ws)
        return ops.linear_block(x, plist[i % ncopy], swiglu=(kind == 1), workspace=ws)
    line = f"{name:12s} {M:4d} {N:6d} {K:6d} | {t_ref:9.1f}    | {t_sk:8.1f}  |"
    for b in a.blocks:
        for w in a.wnt:
            lib.md_debug_set_block_gemm(ctypes.c_int(b), ctypes.c_int(w))
            t = timeit(block, a.iters)
            line += f" {t:12.1f} {nbytes / t / 1e6:5.2f} {2 * M * N * K / t / 1e6:6.0f} |"
    lib.md_debug_set_block_gemm(ctypes.c_int(0), ctypes.c_int(1))
    if a.check:
        r, y = ref(0), block(0)
        r, y = (r if isinstance(r, torch.Tensor) else r[1]), (y if isinstance(y, torch.Tensor) else y[1])
        d = (r.float() - y.float()).abs().max().item()
        line += f" max|lib - block| {d:.4f} (max|ref| {r.float().abs().max().item():.2f})"
    print(line, flush=True)
    del plist, wlist
