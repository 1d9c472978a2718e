"""Offline tuning of the decode-path GEMMs with PyTorch TunableOp (hipBLASLt / rocBLAS solution selection).

The decode GEMMs are skinny (M = B*n in {64,128,256}) and weight-streaming; hipBLASLt's default heuristic picks
un-split tiles for the small-N projections (wqkv/wo/w2), which leaves most CUs idle.  This script tunes every
(M, N, K) the two BASELINE model pairs hit at TP in {1,2,4,8} and writes magicdec_amd/tuned/gemm_gfx950.csv, which
Engine.utils.enable_tuned_gemms() loads (tuning itself is never run in the serving path).

    python tools/tune_gemms.py [--tp 1 2 4 8] [--out magicdec_amd/tuned/gemm_gfx950.csv]
"""
import argparse, os, sys, time
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magicdec_amd.Engine.model_core import ModelArgs

ap = argparse.ArgumentParser()
ap.add_argument("--tp", type=int, nargs="+", default=[1])
ap.add_argument("--models", nargs="+", default=["llama-3.1-8b", "llama-3.2-1b"])
ap.add_argument("--M", type=int, nargs="+", default=[64, 128, 256])
ap.add_argument("--out", default="magicdec_amd/tuned/gemm_gfx950.csv")
a = ap.parse_args()

import torch.cuda.tunable as tunable
tunable.enable(True)
tunable.tuning_enable(True)
tunable.set_max_tuning_duration(30)       # ms per solution
tunable.set_max_tuning_iterations(20)
tunable.set_rotating_buffer_size(1024)     # MiB: cycle operands through > 4x the 256 MiB Infinity Cache so that
                                           # solutions are ranked on HBM-streamed weights, as in the real layer loop
tunable.set_filename(a.out)
if os.path.exists(a.out):
    tunable.read_file(a.out)

shapes = set()
for name in a.models:
    c = ModelArgs.from_name(name)
    D = c.head_dim
    for tp in a.tp:
        if c.n_local_heads % tp:
            continue
        H, KH = c.n_head // tp, c.n_local_heads // tp
        I = c.intermediate_size // tp
        V = c.vocab_size // tp
        for (N, K) in [((H + 2 * KH) * D, c.dim), (c.dim, H * D), (2 * I, c.dim), (c.dim, I), (V, c.dim)]:
            for M in a.M:
                shapes.add((M, N, K))
shapes = sorted(shapes)
print(f"{len(shapes)} GEMM shapes")
t0 = time.time()
for i, (M, N, K) in enumerate(shapes):
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    F.linear(x, w)
    torch.cuda.synchronize()
    if i % 10 == 0:
        print(f"  {i}/{len(shapes)}  M={M} N={N} K={K}  elapsed {time.time() - t0:.0f}s", flush=True)
tunable.write_file() if hasattr(tunable, "write_file") else None   # (TunableOp also writes the file at exit)
print("wrote", a.out, "in", round(time.time() - t0), "s")
