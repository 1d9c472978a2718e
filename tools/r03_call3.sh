#!/bin/bash
# round 3, GPU call 3: fused-kernel fixes + NW=16 / temporal-W variants, prefill attention 64-key tiles A/B
set -u
OUT=$PWD/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_fused.py -q -p no:cacheprovider > $OUT/r03_fused_tests2.log 2>&1
echo "fused tests rc=$?"; tail -6 $OUT/r03_fused_tests2.log; grep -n "q differs" $OUT/r03_fused_tests2.log | cut -c1-900 | head -5
timeout 900 python tools/fused_bench.py > $OUT/r03_fused_ab2.txt 2>&1
echo "fused bench rc=$?"; tail -42 $OUT/r03_fused_ab2.txt
for kt in 32 64; do for nw in 8 4; do
  timeout 200 python tools/attn_bench.py --n 128 --B 64 --S 16032 --iters 10 --kt $kt --nw $nw --hnd 1 2>&1 | grep -v amdgpu.ids
  timeout 200 python tools/attn_bench.py --n 128 --B 64 --S 4128 --iters 10 --kt $kt --nw $nw --hnd 1 2>&1 | grep -v amdgpu.ids
done; done > $OUT/r03_prefill_ab.txt 2>&1
timeout 200 python tools/attn_bench.py --n 128 --B 64 --S 16032 --iters 10 --kt 64 --D 64 --hnd 0 >> $OUT/r03_prefill_ab.txt 2>&1
timeout 200 python tools/attn_bench.py --n 128 --B 64 --S 16032 --iters 10 --kt 32 --D 64 --hnd 0 >> $OUT/r03_prefill_ab.txt 2>&1
cat $OUT/r03_prefill_ab.txt
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_hnd.py tests/test_gpu_fp8.py -q -p no:cacheprovider > $OUT/r03_attn_tests.log 2>&1
echo "attention tests rc=$?"; tail -6 $OUT/r03_attn_tests.log
