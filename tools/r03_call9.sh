#!/bin/bash
# prefill attention at D = 64: 32- vs 64-key tiles with the run-time causal mask (two workgroups per CU again)
set -u
OUT=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for S in 16032 8064 2048; do for kt in 32 64; do
  timeout 200 python tools/attn_bench.py --n 128 --B 64 --S $S --iters 10 --kt $kt --D 64 --hnd 0 2>&1 | grep -v amdgpu.ids
done; done > $OUT/r03_prefill_ab_d64.txt 2>&1
timeout 200 python tools/attn_bench.py --n 128 --B 64 --S 16032 --iters 10 --kt 32 --hnd 1 2>&1 | grep -v amdgpu.ids >> $OUT/r03_prefill_ab_d64.txt
cat $OUT/r03_prefill_ab_d64.txt
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_hnd.py -q -p no:cacheprovider -k "attn or prefill or attention" 2>&1 | tail -2
