#!/bin/bash
# Round 5, GPU call 9: verify attention with 4 vs 8 waves per workgroup at the shapes that launch <= 256 workgroups
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
{
for KH in 1 2 4; do for W in 4 8; do
  python tools/attn_bench.py --KH $KH --H $((KH*4)) --hnd 1 --dwaves $W --iters 40 --layers 4 --reps 2 2>&1 | grep md_paged_attn
done; done
# cfg2: B = 32, 8 K keys, KH = 8 (256 pairs, one split)
for W in 4 8; do python tools/attn_bench.py --B 32 --S 8068 --KH 8 --H 32 --hnd 1 --dwaves $W --iters 40 --layers 4 --reps 2 2>&1 | grep md_paged_attn; done
# TP1 cfg3 (512 workgroups: the rule keeps 4)
for W in 4 8; do python tools/attn_bench.py --KH 8 --H 32 --hnd 1 --dwaves $W --iters 30 --layers 2 --reps 2 2>&1 | grep md_paged_attn; done
# cfg4 TP8 shard: 70B g = 8 -> two M tiles, B = 32 x 32 K keys, KH = 1
for W in 4 8; do python tools/attn_bench.py --B 32 --S 32645 --KH 1 --H 8 --hnd 1 --dwaves $W --iters 30 --layers 4 --reps 2 2>&1 | grep md_paged_attn; done
# cfg5 TP8 shard: Qwen g = 5, fp8, B = 128 x 64 K keys, KH = 1
for W in 4 8; do python tools/attn_bench.py --B 128 --S 65444 --KH 1 --H 5 --hnd 1 --fp8 1 --dwaves $W --iters 20 --layers 2 --reps 2 2>&1 | grep md_paged_attn; done
} > $OUT/r05c9_attn_waves.txt
cut -c1-175 $OUT/r05c9_attn_waves.txt
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_hnd.py tests/test_gpu_fp8.py -q -p no:cacheprovider -k "attn or attention or paged or fullsize or full_size or hnd or fp8" 2>&1 | tail -3
