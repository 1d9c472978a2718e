import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import golden_cfg as gc
from magicdec_amd import ops
DEV="cuda"; BF=torch.bfloat16
z = np.load("tests/golden/stream_prefill.npz")
B, KH, D, budget, ppr = [int(x) for x in z["meta"]]
tab = ops.RopeTable(1024, D, 10000.0, 1.0, device=DEV)
cache = torch.zeros(B * ppr, 2, 128, KH, D, dtype=BF, device=DEV)
rot = torch.empty_like(cache)
bits=lambda t: t.contiguous().view(torch.int16)
for step in range(int(z["nsteps"][0])):
    ctx, n, is_last, npr, last = [int(x) for x in z[f"info{step}"]]
    k = gc.from_bits(z[f"k{step}"]).to(DEV); v = gc.from_bits(z[f"v{step}"]).to(DEV)
    if ctx + n <= budget:
        indices = torch.cat([torch.arange(i * ppr, i * ppr + npr, dtype=torch.int32) for i in range(B)]).to(DEV)
        indptr = (torch.arange(B + 1) * npr).to(torch.int32).to(DEV)
        ops.update_kv(k, v, (torch.arange(B + 1) * n).to(torch.int32).to(DEV), cache, indices, indptr, torch.full((B,), last, dtype=torch.int32, device=DEV))
        valid = ctx + n
    else:
        ops.streaming_shift_append(k, v, cache, n, budget, 16, ppr); valid = budget
    ol = (ctx + n > budget) and is_last
    dst = cache if ol else rot
    if not ol: rot.copy_(cache)
    ops.streaming_rotate(cache, dst, B, valid, ppr, tab)
    for name, mine, ref in (("cache", cache, z[f"cache{step}"]), ("rot", dst, z[f"rot{step}"])):
        r = gc.from_bits(ref); m = mine.cpu()
        ne = (bits(m) != bits(r))
        print(step, name, "ctx", ctx, "n", n, "valid", valid, "mismatch", int(ne.sum()))
        if ne.any():
            idx = torch.nonzero(ne)
            print("  first", idx[:5].tolist(), "pages", sorted(set(idx[:,0].tolist())), "kv", sorted(set(idx[:,1].tolist())), "slots", sorted(set(idx[:,2].tolist()))[:10], "...", sorted(set(idx[:,2].tolist()))[-5:])
            i0 = tuple(idx[0].tolist()); print("  mine", m[i0].item(), bits(m)[i0].item(), "ref", r[i0].item(), bits(r)[i0].item())
