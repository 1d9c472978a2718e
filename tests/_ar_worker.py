"""Worker of tests/test_gpu_allreduce.py: one of WORLD_SIZE processes that all use GPU 0 (the GPU box has a single
device; IPC handles work between processes on the same device, which exercises the whole md_ar_* path -- handle
export / exchange / mapping, the flag protocol, double buffering, graph replay -- everything except the xGMI hop).
Bootstrap transport is gloo (RCCL refuses two ranks on one GPU)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def expected(world, n, seed):
    """sum in rank order, fp32 accumulate, one rounding -- the kernel's definition (csrc/allreduce.hip)."""
    acc = torch.zeros(n, dtype=torch.float32)
    ins = []
    for r in range(world):
        g = torch.Generator().manual_seed(seed * 100 + r)
        x = (torch.randn(n, generator=g) * 3).to(torch.bfloat16)
        ins.append(x)
        acc += x.float()
    return ins, acc.to(torch.bfloat16)


def expected_fused(world, rows, dim, seed, eps):
    """h = bf16(x + bf16(sum in rank order, fp32)), y = bf16(bf16(h * rsqrt(mean(h^2) + eps)) * w): the kernel's
    definition of md_allreduce_add_rmsnorm = the reference's rounding points (Engine/SnapKV/model.py:464-469)."""
    ins, s = expected(world, rows * dim, seed)
    g = torch.Generator().manual_seed(seed * 100 + 77)
    x = torch.randn(rows, dim, generator=g).to(torch.bfloat16)
    w = (1 + 0.1 * torch.randn(dim, generator=g)).to(torch.bfloat16)
    h = x + s.view(rows, dim)
    hf = h.float()
    y = (hf * torch.rsqrt(hf.pow(2).mean(-1, keepdim=True) + eps)).to(torch.bfloat16) * w
    return [t.view(rows, dim) for t in ins], x, w, h, y


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    # the expected values are computed on the host between kernel launches: on the GPU boxes' 256-core hosts torch's
    # default intra-op pool (x 3 processes) can stall a small CPU op for seconds -- longer than the kernels' 2 s
    # bounded spin, which then (correctly) reports a missing peer (seen once: GPU call r03_call1)
    torch.set_num_threads(4)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from magicdec_amd.Engine import oneshot
    from magicdec_amd.Engine.oneshot import OneShotAllReduce
    ar = OneShotAllReduce(dist.group.WORLD, max_bytes=4 << 20)
    dev = "cuda:0"
    call = 0
    algos = (oneshot.ALGO_ONESHOT, oneshot.ALGO_TWOSHOT, oneshot.ALGO_AUTO)
    # eager calls of many sizes (1 vector ... the full 4 MiB buffer), in place; odd/even call counts hit both halves
    # sizes alternate (large, small, large, ...) without any host synchronisation between calls: a small call must
    # never disturb the half-buffer a slower peer is still reading for the previous large call
    pending = []
    for rep in range(3):
        for n in (2 * 1024 * 1024, 8, 256 * 4096, 4096, 64 * 2048, 64):
            call += 1
            ins, want = expected(world, n, call)
            pending.append((n, ins[rank].to(dev), want))
    torch.cuda.synchronize()
    dist.barrier()
    for i, (n, t, want) in enumerate(pending):            # 18 kernels queued back to back, algorithms interleaved
        ar.all_reduce_(t, algos[i % 3])
    torch.cuda.synchronize()
    for n, t, want in pending:
        assert torch.equal(t.cpu().view(torch.int16), want.view(torch.int16)), (n, "back-to-back mismatch")
    for n in (8, 4096, 64 * 2048, 256 * 4096 + 8, 2 * 1024 * 1024):
        for rep in range(3):
            call += 1
            ins, want = expected(world, n, call)
            t = ins[rank].to(dev)
            dist.barrier()
            ar.all_reduce_(t, algos[rep])
            got = t.cpu()
            assert torch.equal(got.view(torch.int16), want.view(torch.int16)), (n, rep, "eager mismatch")
    # fused all-reduce + residual add + RMSNorm: rows of the 8B verify step, a 1B draft step, ragged row counts; the
    # sum and h are integer-exact, y is compared with the sequence above (the in-row fp32 sum of squares runs in
    # another order on the GPU: a 1-ulp flip of bf16(h * rstd) becomes at most 2 ulps after the bf16 multiply by w;
    # > 99.9 % of the elements are bit-equal)
    for rows, dim in ((256, 4096), (64, 2048), (5, 512), (1, 8192), (67, 1024)):
        for algo in algos:
            call += 1
            ins, x, w, h_want, y_want = expected_fused(world, rows, dim, call, 1e-5)
            dist.barrier()         # host-side skew (the CPU reference above) must not eat into the kernels' spin bound
            h, y = ar.all_reduce_add_rmsnorm(ins[rank].contiguous().to(dev), x.to(dev), w.to(dev), 1e-5, algo)
            h, y = h.cpu(), y.cpu()
            if not torch.equal(h.view(torch.int16), h_want.view(torch.int16)):
                neq = h.view(torch.int16) != h_want.view(torch.int16)
                rws = sorted(set(torch.nonzero(neq)[:, 0].tolist()))
                raise AssertionError((rows, dim, algo, "fused h mismatch", "elements", int(neq.sum()), "rows", rws[:24],
                                      len(rws), "nan", int(torch.isnan(h.float()).sum()), "status", ar.status(),
                                      "got", h[rws[0], :4].tolist(), "want", h_want[rws[0], :4].tolist(),
                                      "x", x[rws[0], :4].tolist()))
            d = (y.view(torch.int16).int() - y_want.view(torch.int16).int()).abs()
            same_sign = (y.view(torch.int16).int() ^ y_want.view(torch.int16).int()) >= 0
            bad_rows = sorted(set(torch.nonzero((d > 2) & same_sign)[:, 0].tolist()))
            assert bool((d[same_sign] <= 2).all()) and float((d == 0).float().mean()) > 0.999, (
                rows, dim, algo, "fused y", "equal fraction", float((d == 0).float().mean()), "max d", int(d.max()),
                "rows with d > 2", bad_rows[:20], len(bad_rows), "y", y[bad_rows[0], :4].tolist() if bad_rows else None,
                "want", y_want[bad_rows[0], :4].tolist() if bad_rows else None)
    # a captured graph with three dependent all-reduces, replayed with fresh inputs (flags live in device memory)
    n = 256 * 4096
    static = [torch.zeros(n, dtype=torch.bfloat16, device=dev) for _ in range(3)]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i, t in enumerate(static):
            ar.all_reduce_(t, algos[i])    # warm-up outside capture
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    dist.barrier()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for i, t in enumerate(static):
            ar.all_reduce_(t, algos[i])
    for it in range(3):
        wants = []
        for k, t in enumerate(static):
            call += 1
            ins, want = expected(world, n, call)
            t.copy_(ins[rank])
            wants.append(want)
        g.replay()
        torch.cuda.synchronize()
        for k, t in enumerate(static):
            assert torch.equal(t.cpu().view(torch.int16), wants[k].view(torch.int16)), (it, k, "graph mismatch")
    # argument validation
    bad = torch.zeros(12, dtype=torch.bfloat16, device=dev)
    try:
        ar.all_reduce_(bad)
        raise SystemExit("numel % 8 != 0 was accepted")
    except ValueError:
        pass
    assert ar.status() == 0, "a kernel timed out waiting for its peer"
    ar.check()
    dist.barrier()
    ar.close()
    dist.destroy_process_group()
    print(f"rank {rank}: OK ({call} all-reduces)")


if __name__ == "__main__":
    main()
