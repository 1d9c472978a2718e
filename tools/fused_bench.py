"""A/B of md_linear_fused (csrc/tilegemm.hip: linear + consumer op in ONE launch) against what the step runs otherwise
(hipBLASLt or md_linear, followed by the separate rope+append / add / SiLU*mul launch), per linear of a decode-step
layer, graph-captured, weights cycled through > 600 MB to defeat the Infinity Cache.

    python tools/fused_bench.py [--only 1B/4] [--iters 30]

One line per (model shard, M, linear): `unfused` = library GEMM (TunableOp table loaded) + the small kernel behind it,
`skinny` = md_linear (+ the small kernel where it has no fused epilogue), `fused` = md_linear_fused.  The winner per shape
is what Engine/gemm_policy.py encodes."""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magicdec_amd import _lib, ops                          # noqa: E402
from magicdec_amd.Engine.utils import enable_tuned_gemms   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--only", default="")
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--tiles", type=int, default=0, help="1: also time md_linear_fused with 1 x 1 and 2 x 2 tiles forced (dev knob)")
ap.add_argument("--pro", type=int, default=0, help="1: also time the deferred-RMSNorm (PRO) form of the swiglu / qkv linears")
a = ap.parse_args()
print("tuned GEMM table loaded:", enable_tuned_gemms())
dev = "cuda"


def layer_shapes(dim, H, KH, D, I, tp):
    h, kh, i = H // tp, KH // tp, I // tp
    return [("wqkv", (h + 2 * kh) * D, dim, "qkv", (h, kh)), ("wo", dim, h * D, "resid", None),
            ("w13", 2 * i, dim, "swiglu", None), ("w2", dim, i, "resid", None)]


MODELS = {"1B": (2048, 32, 8, 64, 8192), "8B": (4096, 32, 8, 128, 14336)}
CASES = [("1B", 1, 64), ("1B", 1, 128), ("1B", 4, 64), ("1B", 4, 128), ("8B", 8, 256), ("8B", 8, 64), ("8B", 4, 256),
         ("8B", 2, 256), ("8B", 1, 64), ("8B", 1, 256), ("8B", 1, 128), ("8B", 1, 32)]


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


_lib.load()
ws = ops.AttnWorkspace(dev)
print(f"{'case':22s} {'M':>4s} {'N':>6s} {'K':>6s} {'MB':>6s} | {'unfused us':>10s} | {'skinny us':>9s} | {'fused us':>8s} | fused/unfused")
for model, tp, M in CASES:
    dim, H, KH, D, I = MODELS[model]
    for lname, N, K, kind, heads in layer_shapes(dim, H, KH, D, I, tp):
        tag = f"{model}/{tp} {lname}"
        if a.only and a.only not in f"{tag} M{M}":
            continue
        nbytes = N * K * 2
        ncopy = max(2, int(600e6 // nbytes) + 1)
        wl = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(ncopy)]
        pk = [ops.PackedWeight(w, swiglu=(kind == "swiglu")) for w in wl]
        x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        skinny_ok = ops.linear_supported(M, N, K, kind == "swiglu")
        if kind == "swiglu":
            Ih = N // 2

            def unfused(i):
                h = F.linear(x, wl[i % ncopy])
                return ops.silu_mul(h[:, :Ih], h[:, Ih:])

            def skinny(i):
                return ops.linear(x, pk[i % ncopy], swiglu=True, workspace=ws)

            def fused(i):
                return ops.fused_linear(x, pk[i % ncopy], swiglu=True)
        elif kind == "resid":
            r = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
            nw = torch.ones(N, device=dev, dtype=torch.bfloat16)

            def unfused(i):
                return ops.add_rmsnorm(r, F.linear(x, wl[i % ncopy]), nw, 1e-5)

            def skinny(i):
                return ops.add_rmsnorm(r, ops.linear(x, pk[i % ncopy], workspace=ws), nw, 1e-5)

            def fused(i):
                return ops.rmsnorm(ops.fused_linear(x, pk[i % ncopy], resid=r), nw, 1e-5)
        else:
            h, kh = heads
            n_rows = 4 if M == 256 else (2 if M == 128 else 1)
            B = M // n_rows
            pages = 3
            cache = torch.zeros(B * pages, 2, 128, kh, D, device=dev, dtype=torch.bfloat16)
            indices = torch.arange(B * pages, device=dev, dtype=torch.int32)
            indptr = torch.arange(B + 1, device=dev, dtype=torch.int32) * pages
            last = torch.full((B,), 5, device=dev, dtype=torch.int32)
            offs = torch.full((B,), 300, device=dev, dtype=torch.int32)
            ip = torch.arange(B + 1, device=dev, dtype=torch.int32) * n_rows
            tab = ops.RopeTable(1024, D, 500000.0, 8.0, 1, 4, 8192, device=dev)

            def split(qkv):
                return (qkv[:, :h * D].unflatten(1, (h, D)), qkv[:, h * D:(h + kh) * D].unflatten(1, (kh, D)),
                        qkv[:, (h + kh) * D:].unflatten(1, (kh, D)))

            def unfused(i):
                q, k, v = split(F.linear(x, wl[i % ncopy]))
                return ops.rope_append(q, k, v, ip, offs, tab, cache, indices, indptr, last, n_max=n_rows)

            def skinny(i):
                q, k, v = split(ops.linear(x, pk[i % ncopy], workspace=ws))
                return ops.rope_append(q, k, v, ip, offs, tab, cache, indices, indptr, last, n_max=n_rows)

            def fused(i):
                return ops.fused_qkv_rope_append(x, pk[i % ncopy], None, h, kh, D, n_rows, offs, tab, cache, indices,
                                                 indptr, last)
        fused_pro = None
        if a.pro and kind in ("swiglu", "qkv"):
            pro = ops.DeferredNorm(x, torch.rand(M, K // 32, device=dev) * 32.0, torch.ones(K, device=dev, dtype=torch.bfloat16), 1e-5)
            if kind == "swiglu":
                def fused_pro(i):
                    return ops.fused_linear(x, pk[i % ncopy], swiglu=True, pro=pro)
            else:
                def fused_pro(i):
                    return ops.fused_qkv_rope_append(x, pk[i % ncopy], None, h, kh, D, n_rows, offs, tab, cache, indices,
                                                     indptr, last, pro=pro)
        t_u = timeit(unfused, a.iters)
        t_s = timeit(skinny, a.iters) if skinny_ok else float("nan")
        t_f = timeit(fused, a.iters)
        extra = ""
        if fused_pro is not None:
            extra += f" | pro: {timeit(fused_pro, a.iters):6.1f}"
            if kind == "swiglu" and skinny_ok:                       # md_linear_normed: the same norm on the streaming kernel
                extra += (f" | skinny+norm launch: {timeit(lambda i: ops.linear(ops.rmsnorm(x, pro.weight, 1e-5), pk[i % ncopy], swiglu=True, workspace=ws), a.iters):6.1f}"
                          f" | skinny pro: {timeit(lambda i: ops.linear(x, pk[i % ncopy], swiglu=True, workspace=ws, pro=pro), a.iters):6.1f}")
        if a.tiles:
            import ctypes
            lib = _lib.load()
            for knob in (11, 22):
                lib.md_debug_set_fused_nw(ctypes.c_int(knob))
                extra += f" | tiles {knob}: {timeit(fused, a.iters):6.1f}"
            lib.md_debug_set_fused_nw(ctypes.c_int(0))
        print(f"{tag + ' ' + kind:22s} {M:4d} {N:6d} {K:6d} {nbytes / 1e6:6.1f} | {t_u:10.1f} | {t_s:9.1f} | {t_f:8.1f} | "
              f"{t_f / t_u:5.2f}" + extra, flush=True)
        del wl, pk
