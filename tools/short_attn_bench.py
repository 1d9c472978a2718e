"""In-graph duration of md_paged_attn on the SHORT caches of the draft steps (SnapKV budget 257, StreamingLLM 257), by
cache length: the intercept is the kernel's fixed cost (launch, page-table read, merge, store), the slope its tile loop.
    python tools/short_attn_bench.py [--B 64 --KH 8 --H 32 --D 64 --n 1]

Round 4 finding (profiles/r04_short_attn_bench.txt): 4.7 us up to 3 tiles, then + bytes / (5-7 TB/s): 9.7 us at the 257-row
cache of a draft step = ~4 us fixed + 33.7 MB of K/V at 6 TB/s, NOT a latency chain -- keeping three tiles of a wave in
flight (a third register set) and fetching the page ids ahead of the tile loads were both built, bit-identical, and
changed nothing (9.72-9.78 us), and were removed."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magicdec_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
for k, v in dict(B=64, KH=8, H=32, D=64, n=1, iters=40, hnd=0, dwaves=0).items():
    ap.add_argument(f"--{k}", type=int, default=v)
a = ap.parse_args()
dev = "cuda"
if a.dwaves:                                   # wavefronts per workgroup of the decode kernel forced (4 | 8)
    import ctypes
    from magicdec_amd import _lib
    _lib.load().md_debug_set_attn_waves(ctypes.c_int(a.dwaves))


def timeit(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3


ws = ops.AttnWorkspace(dev)
x = torch.randn(a.B * a.n, 2048, device=dev, dtype=torch.bfloat16)
nw = torch.ones(2048, device=dev, dtype=torch.bfloat16)
print(f"floor: md_rmsnorm on {a.B * a.n} x 2048 in the same graph harness: {timeit(lambda: ops.rmsnorm(x, nw, 1e-5), a.iters):.2f} us")
layout = "HND" if a.hnd else "NHD"
for S in (16, 33, 65, 129, 257, 385, 513, 1025, 2049):
    mp = (S + 127) // 128
    cache = torch.randn(a.B * mp, 2, 128, a.KH, a.D, device=dev, dtype=torch.float32).to(torch.bfloat16)
    if a.hnd:
        cache = cache.permute(0, 1, 3, 2, 4).contiguous()
    q = torch.randn(a.B * a.n, a.H, a.D, device=dev, dtype=torch.float32).to(torch.bfloat16)
    indices = torch.arange(a.B * mp, dtype=torch.int32, device=dev)
    indptr = torch.arange(a.B + 1, dtype=torch.int32, device=dev) * mp
    last = torch.full((a.B,), S - (mp - 1) * 128, dtype=torch.int32, device=dev)
    qo = torch.arange(a.B + 1, dtype=torch.int32, device=dev) * a.n
    t = timeit(lambda: ops.paged_attention(q, cache, qo, indices, indptr, last, a.n, mp, ws, kv_layout=layout), a.iters)
    nbytes = a.B * S * a.KH * a.D * 2 * 2
    print(f"B={a.B} KH={a.KH} H={a.H} D={a.D} n={a.n} waves={a.dwaves or 'rule'} S={S:5d} ({(S + 31) // 32:3d} tiles): {t:6.2f} us  "
          f"({nbytes / 1e6:6.1f} MB, {nbytes / t / 1e6:5.2f} TB/s)", flush=True)
