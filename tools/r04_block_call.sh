#!/bin/bash
# round 4: md_linear_block with the W stream queued in the loader waves' registers (inline-asm loads, counted vmcnt)
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
# 1 = shipped DMA rings | 257 = x ring 4 + W 6 stages in registers | 513 = x ring 3 + W reg 6 | 769 = x ring 4 + W reg 4
timeout 600 python tools/block_bench.py --only "8B w" --blocks 256 --wnt 1 257 513 769 --check 0 2>&1 | grep -v amdgpu.ids > $OUT/r04_block_ab_call28.txt
timeout 600 python tools/block_bench.py --only "8B w13 v" --blocks 256 --wnt 257 --check 1 2>&1 | grep -v "amdgpu.ids\|tuned" >> $OUT/r04_block_ab_call28.txt
timeout 600 python tools/block_bench.py --only "8B w2 v" --blocks 256 --wnt 257 --check 1 2>&1 | grep -v "amdgpu.ids\|tuned" >> $OUT/r04_block_ab_call28.txt
cat $OUT/r04_block_ab_call28.txt
