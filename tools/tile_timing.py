"""Where does a launch of the tile GEMM (md_linear_fused / md_linear_fused_split, csrc/tilegemm.hip) spend its time?
Every wavefront records the 100 MHz wall clock at six points (md_debug_set_tile_timing, dev builds); this prints, per case,
the phase boundaries relative to the first wavefront's entry: min / median / max over the wavefronts.  The activations are
REWRITTEN by another kernel before every launch (as in a real step, where the previous kernel produced them) and the weights
are cycled through > 600 MB.

    python tools/tile_timing.py [--only w13]
"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the instrumented build (make -C magicdec_amd/csrc TIMING=1): the product library records nothing
os.environ.setdefault("MAGICDEC_HIP_LIB", os.path.join(ROOT, "magicdec_amd", "libmagicdec_hip_timing.so"))
from magicdec_amd import _lib, ops                          # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--only", default="")
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
dev = "cuda"
lib = _lib.load()
ws = ops.AttnWorkspace(dev)
NAMES = ["entry", "loads issued", "x chunk 0 staged", "K slice consumed", "partials in LDS", "stores issued"]

# name, kind, M, N, K
CASES = [("1B w13 swiglu", "swiglu", 64, 16384, 2048), ("1B w13 swiglu+pro 1x1", "swiglu_pro11", 64, 16384, 2048),
         ("1B w13 swiglu+pro", "swiglu_pro", 64, 16384, 2048),
         ("1B w13 swiglu 2x2", "swiglu22", 64, 16384, 2048),
         ("1B wo resid", "resid", 64, 2048, 2048), ("1B wqkv plain", "plain", 64, 3072, 2048),
         ("1B w2 split", "split", 64, 2048, 8192), ("1B w2 fused resid", "resid", 64, 2048, 8192),
         ("8B wo M32 resid", "resid", 32, 4096, 4096)]

for name, kind, M, N, K in CASES:
    if a.only and a.only not in name:
        continue
    nbytes = N * K * 2
    ncopy = max(2, int(600e6 // nbytes) + 1)
    sw = kind.startswith("swiglu")
    wl = [ops.PackedWeight(torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02, swiglu=sw) for _ in range(ncopy)]
    src = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    x = torch.empty_like(src)
    r = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    nw = torch.ones(N if kind in ("resid", "split") else K, device=dev, dtype=torch.bfloat16)
    pro = ops.DeferredNorm(x, torch.rand(M, K // 32, device=dev) * 32.0, torch.ones(K, device=dev, dtype=torch.bfloat16), 1e-5)
    buf = torch.zeros(4096 * 16 * 6, dtype=torch.int64, device=dev)

    def run(i):
        if kind == "swiglu":
            lib.md_debug_set_fused_nw(ctypes.c_int(11))
            return ops.fused_linear(x, wl[i % ncopy], swiglu=True)
        if kind == "swiglu22":
            lib.md_debug_set_fused_nw(ctypes.c_int(22))
            return ops.fused_linear(x, wl[i % ncopy], swiglu=True)
        if kind == "swiglu_pro11":
            lib.md_debug_set_fused_nw(ctypes.c_int(11))
            return ops.fused_linear(x, wl[i % ncopy], swiglu=True, pro=pro)
        if kind == "swiglu_pro":
            return ops.fused_linear(x, wl[i % ncopy], swiglu=True, pro=pro)
        if kind == "resid":
            return ops.fused_linear(x, wl[i % ncopy], resid=r, want_ssq=True)
        if kind == "plain":
            return ops.fused_linear(x, wl[i % ncopy])
        return ops.fused_split_linear_add_rmsnorm(x, wl[i % ncopy], r, nw, 1e-5, workspace=ws)

    rows = []
    for rep in range(a.reps + 2):
        x.copy_(src)                                   # the activations were just written by another kernel
        buf.zero_()
        torch.cuda.synchronize()
        lib.md_debug_set_tile_timing(ctypes.c_void_p(buf.data_ptr()))
        run(rep)
        lib.md_debug_set_tile_timing(None)
        lib.md_debug_set_fused_nw(ctypes.c_int(0))
        torch.cuda.synchronize()
        if rep < 2:
            continue
        t = buf.view(-1, 6).cpu()
        t = t[t[:, 0] > 0].double()
        t0 = t[:, 0].min()
        rows.append(((t - t0) / 100.0))                # us
    t = torch.cat(rows)
    nwv = t.shape[0] // a.reps
    print(f"{name:20s} M={M} N={N} K={K} {nbytes / 1e6:.1f} MB  ({nwv} wavefronts, {a.reps} launches; us after the first entry)")
    for i, nm in enumerate(NAMES):
        c = t[:, i]
        print(f"    {nm:18s} min {c.min():6.2f}  p10 {c.quantile(0.1):6.2f}  median {c.median():6.2f}  p90 {c.quantile(0.9):6.2f}  max {c.max():6.2f}")
    del wl
