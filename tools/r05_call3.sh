#!/bin/bash
# Round 5, GPU call 3: split-KV target of the verify attention at the TP shard shapes (KH = 1, 2, 4 of cfg3; HND pages)
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for KH in 1 2 4; do
  for W in 256 384 512 768 1024; do
    python tools/attn_bench.py --KH $KH --H $((KH*4)) --hnd 1 --wgs $W --iters 40 --layers 4 --reps 2 2>&1 | grep md_paged_attn
  done
done > $OUT/r05c3_attn_split_sweep.txt
cat $OUT/r05c3_attn_split_sweep.txt | cut -c1-150
