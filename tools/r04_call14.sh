#!/bin/bash
# round 4, GPU call 15: prefill32p at 32-key tiles (no spills) vs the shipped kernels
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "prefill_kernel_variants" -p no:cacheprovider 2>&1 | tail -2
AB=$OUT/r04_prefill_ab_call15.txt; : > $AB
echo "# D=128, B=64, 128 query tokens x 32 heads, 16 032 keys, HND: 128 = shipped, 32 = 32-key tiles, 33 = pipelined across tiles at 32 keys, 65 = at 64 keys (40 spilled registers)" >> $AB
timeout 300 python tools/attn_bench.py --n 128 --B 64 --S 16032 --iters 10 --D 128 --hnd 1 --variants 128,33,32,65 --reps 2 2>&1 | grep -v amdgpu.ids | grep "prefill view\|nan [1-9]" >> $AB
cat $AB
