"""What ONE fused all-reduce + add + RMSNorm launch (md_allreduce_add_rmsnorm) costs a LONE rank -- the protocol's fixed
overhead before a byte crosses a link -- against the add + RMSNorm launch it replaces (RCCL's one-rank all-reduce is a
no-op), graph-captured like a decode step.  python tools/ar_bench.py [--iters 40]"""
import argparse
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magicdec_amd import ops                                    # noqa: E402
from magicdec_amd.Engine.oneshot import OneShotAllReduce       # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=40)
a = ap.parse_args()
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29688")
dist.init_process_group("gloo", rank=0, world_size=1)
dev = "cuda"
ar = OneShotAllReduce(dist.group.WORLD, max_bytes=4 << 20)


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
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * 5) * 1e3


for rows, dim in ((256, 4096), (64, 4096), (64, 2048), (128, 2048)):
    parts = [torch.randn(rows, dim, device=dev).to(torch.bfloat16) for _ in range(4)]
    resid = torch.randn(rows, dim, device=dev).to(torch.bfloat16)
    w = torch.ones(dim, device=dev, dtype=torch.bfloat16)
    t_norm = timeit(lambda i: ops.add_rmsnorm(resid, parts[i % 4], w, 1e-5), a.iters)
    t_fused = timeit(lambda i: ar.all_reduce_add_rmsnorm(parts[i % 4], resid, w, 1e-5), a.iters)
    t_plain = timeit(lambda i: ar.all_reduce_(parts[i % 4]), a.iters)
    h0, y0 = ops.add_rmsnorm(resid, parts[0], w, 1e-5)
    h1, y1 = ar.all_reduce_add_rmsnorm(parts[0], resid, w, 1e-5)
    torch.cuda.synchronize()
    same = torch.equal(h0, h1) and float((y0.float() - y1.float()).abs().max()) <= 2.0 ** -6 * float(y0.float().abs().max())
    print(f"[{rows:3d} x {dim}] add+rmsnorm launch {t_norm:5.2f} us | lone-rank fused all-reduce+add+rmsnorm {t_fused:5.2f} us "
          f"(+{t_fused - t_norm:4.2f}) | lone-rank plain all-reduce {t_plain:5.2f} us | outputs agree: {same}", flush=True)
assert ar.status() == 0
ar.close()
dist.destroy_process_group()
