"""Stand-alone GPU check of md_paged_attn / md_rope / md_append_paged_kv against the
oracle (oracle/flashinfer_ref.py).  Developer tool: prints per-case error
statistics so that one gpurun call gives enough information to debug a layout bug.
Usage on the GPU box:  python tools/gpu_attn_check.py > gpurun_out/attn_check.log
"""
import ctypes
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import flashinfer_ref as fr  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, "magicdec_amd", "libmagicdec_hip.so"))
P, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
lib.md_last_error_string.restype = ctypes.c_char_p
lib.md_paged_attn_workspace_bytes.restype = ctypes.c_size_t
lib.md_paged_attn_workspace_bytes.argtypes = [I] * 7
lib.md_paged_attn.argtypes = [P, L, P, P, P, P, P, P, I, I, I, I, I, I, I, ctypes.c_float, I, P, ctypes.c_size_t, P]
dev = "cuda"


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def run_case(name, B, n, H, KH, D, lens, page_size=128, causal=True, seed=0, scatter_pages=False):
    g = torch.Generator().manual_seed(seed)
    max_pages = max((l + page_size - 1) // page_size for l in lens)
    npages_tot = B * max_pages + 3
    cache = (torch.randn(npages_tot, 2, page_size, KH, D, generator=g) * 1.0).to(torch.bfloat16)
    q = (torch.randn(B * n, H, D, generator=g) * 1.0).to(torch.bfloat16)
    perm = torch.randperm(npages_tot, generator=g) if scatter_pages else torch.arange(npages_tot)
    indices, indptr, last = [], [0], []
    for b in range(B):
        np_b = (lens[b] + page_size - 1) // page_size
        indices += [int(perm[b * max_pages + i]) for i in range(np_b)]
        indptr.append(indptr[-1] + np_b)
        last.append(lens[b] - (np_b - 1) * page_size if np_b > 0 else 0)
    indices = torch.tensor(indices + [0], dtype=torch.int32)
    indptr = torch.tensor(indptr, dtype=torch.int32)
    last = torch.tensor(last, dtype=torch.int32)
    qo = torch.arange(B + 1, dtype=torch.int32) * n
    ref = fr.batch_prefill_paged(q, cache, qo, indices, indptr, last, H, KH, D, causal=causal).float()

    dq, dc = q.to(dev), cache.to(dev)
    dind, dptr_, dlast, dqo = indices.to(dev), indptr.to(dev), last.to(dev), qo.to(dev)
    out = torch.full_like(dq, float("nan"))
    wsb = lib.md_paged_attn_workspace_bytes(B, n, H, KH, D, max_pages, page_size)
    ws = torch.empty(max(wsb, 256), dtype=torch.uint8, device=dev)
    rc = lib.md_paged_attn(ptr(dq), H * D, ptr(dc), ptr(out), ptr(dqo), ptr(dind), ptr(dptr_), ptr(dlast), B, n, H,
                           KH, D, page_size, 1 if causal else 0, 1.0 / math.sqrt(D), max_pages, ptr(ws), wsb, None)
    torch.cuda.synchronize()
    if rc != 0:
        print(f"[{name}] rc={rc} err={lib.md_last_error_string()}")
        return False
    o = out.float().cpu()
    err = (o - ref).abs()
    nan = torch.isnan(o).sum().item()
    mx = err.max().item() if nan == 0 else float("nan")
    ok = nan == 0 and mx < 2e-2 * max(1.0, ref.abs().max().item())
    print(f"[{name}] B={B} n={n} H={H} KH={KH} D={D} lens={lens[:4]}.. wsb={wsb} nan={nan} max_err={mx:.4g} "
          f"ref_max={ref.abs().max().item():.3g} mean_err={err.nanmean().item():.3g} {'OK' if ok else 'FAIL'}")
    if not ok:
        # per (row, head) error map of the first request to localise layout bugs
        e = err[:n].amax(dim=-1)
        print("  err[row,head] first request:\n", e)
        ed = err[:n].amax(dim=(0, 1))
        print("  err by d:", ed)
    return ok


def main():
    print("device:", torch.cuda.get_device_name(0))
    allok = True
    # verify shape of Llama-3.1-8B (g=4, gamma+1=4): one M tile
    allok &= run_case("verify-small", 2, 4, 8, 2, 128, [300, 257])
    allok &= run_case("verify-1tile", 1, 4, 4, 1, 128, [20])
    allok &= run_case("verify-32", 1, 4, 4, 1, 128, [32])
    allok &= run_case("verify-33", 1, 4, 4, 1, 128, [33])
    allok &= run_case("verify-ragged", 3, 4, 32, 8, 128, [1000, 129, 640], scatter_pages=True)
    allok &= run_case("verify-long-split", 2, 4, 8, 2, 128, [8065 + 4, 7000])
    allok &= run_case("draft-1row", 4, 1, 32, 8, 64, [260, 258, 300, 257])
    allok &= run_case("draft-2row", 4, 2, 8, 2, 64, [260, 258, 300, 257])
    allok &= run_case("g8-qt2", 2, 4, 16, 2, 128, [500, 300])
    allok &= run_case("g5-qt2", 2, 4, 10, 2, 128, [500, 300])
    allok &= run_case("g1-mha", 2, 1, 12, 12, 64, [129, 200])
    allok &= run_case("prefill-chunk", 2, 128, 8, 2, 128, [384, 384])
    allok &= run_case("prefill-last", 2, 32, 8, 2, 128, [160, 160])
    allok &= run_case("prefill-d64", 2, 128, 8, 2, 64, [256, 256])
    allok &= run_case("noncausal", 2, 4, 8, 2, 128, [300, 257], causal=False)
    allok &= run_case("empty-req", 2, 4, 8, 2, 128, [0, 200])
    print("ALL OK" if allok else "SOME FAILED")

    # quick timing of the north-star verify shape on one layer (B=64, 16K, KH=8)
    B, n, H, KH, D, S = 64, 4, 32, 8, 128, 16032 + 4
    max_pages = (S + 127) // 128
    cache = torch.randn(B * max_pages, 2, 128, KH, D, device=dev, dtype=torch.bfloat16)
    q = torch.randn(B * n, H, D, device=dev, dtype=torch.bfloat16)
    indices = torch.arange(B * max_pages, dtype=torch.int32, device=dev)
    indptr = (torch.arange(B + 1, dtype=torch.int32) * max_pages).to(dev)
    last = torch.full((B,), S - (max_pages - 1) * 128, dtype=torch.int32, device=dev)
    qo = (torch.arange(B + 1, dtype=torch.int32) * n).to(dev)
    out = torch.empty_like(q)
    wsb = lib.md_paged_attn_workspace_bytes(B, n, H, KH, D, max_pages, 128)
    ws = torch.empty(max(wsb, 256), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def call():
        return lib.md_paged_attn(ptr(q), H * D, ptr(cache), ptr(out), ptr(qo), ptr(indices), ptr(indptr), ptr(last),
                                 B, n, H, KH, D, 128, 1, 1.0 / math.sqrt(D), max_pages, ptr(ws), wsb,
                                 ctypes.c_void_p(st))
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    iters = 20
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    bytes_ = B * S * KH * D * 2 * 2 + 2 * B * n * H * D * 2
    print(f"verify layer B=64 S=16K KH=8: {ms:.3f} ms  {bytes_ / ms / 1e6:.1f} GB/s  ({bytes_ / ms / 1e6 / 8000 * 100:.1f}% of 8 TB/s)")
    return 0 if allok else 1


if __name__ == "__main__":
    sys.exit(main())
