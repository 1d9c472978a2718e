#!/bin/bash
# 32x32x16-MFMA prefill attention kernel: correctness (the attention parity tests with the knob on) and A/B
set -u
OUT=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for kt in 32 64; do
  MAGICDEC_PREFILL_MFMA32=$kt timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_hnd.py -q -p no:cacheprovider -k "attn or prefill or attention" 2>&1 | tail -3
done
{
for D in 128 64; do for S in 16032 4128; do
  timeout 200 python tools/attn_bench.py --n 128 --B 64 --S $S --iters 10 --D $D --hnd $([ $D = 128 ] && echo 1 || echo 0) 2>&1 | grep -v amdgpu.ids
  for kt in 32 64; do
    timeout 200 python tools/attn_bench.py --n 128 --B 64 --S $S --iters 10 --D $D --hnd $([ $D = 128 ] && echo 1 || echo 0) --mfma32 $kt 2>&1 | grep -v amdgpu.ids
  done
done; done
} > $OUT/r03_prefill_mfma32_ab.txt 2>&1
cat $OUT/r03_prefill_mfma32_ab.txt
