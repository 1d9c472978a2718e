"""Developer probe: achieved weight-streaming bandwidth of torch F.linear (hipBLASLt) on the decode GEMM shapes,
cycling through enough distinct weight copies to defeat the 256 MiB Infinity Cache."""
import torch, sys
import torch.nn.functional as F
dev = "cuda"
shapes = [("1B wqkv", 64, 3072, 2048), ("1B wo", 64, 2048, 2048), ("1B w13", 64, 16384, 2048), ("1B w2", 64, 2048, 8192),
          ("1B head", 64, 128256, 2048),
          ("8B wqkv", 256, 6144, 4096), ("8B wo", 256, 4096, 4096), ("8B w13", 256, 28672, 4096), ("8B w2", 256, 4096, 14336),
          ("8B head", 256, 128256, 4096),
          ("8B wqkv ar", 64, 6144, 4096), ("8B wo ar", 64, 4096, 4096), ("8B w13 ar", 64, 28672, 4096), ("8B w2 ar", 64, 4096, 14336)]
for name, M, N, K in shapes:
    nbytes = N * K * 2
    ncopy = max(2, int(600e6 // nbytes) + 1)
    ws = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) for _ in range(ncopy)]
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    for i in range(3):
        F.linear(x, ws[i % ncopy])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 40
    e0.record()
    for i in range(iters):
        F.linear(x, ws[i % ncopy])
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:12s} M={M:4d} N={N:6d} K={K:5d}  {ms*1e3:8.1f} us  {nbytes/ms/1e6:8.1f} GB/s  ({nbytes/1e6:.0f} MB)")
    del ws
