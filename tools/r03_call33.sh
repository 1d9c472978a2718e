#!/bin/bash
# NOTE: provenance only -- the kernel this script measures (prefill32x2_attn_kernel: two 32-row tiles per wave, 4 waves, 256 VGPRs +
# 233 AGPRs) was measured and NOT kept (profiles/r03_prefill_x2_rejected_call33.txt); knob values 264 / 328 no longer exist.
# GPU call 33: two 32-row query tiles per wave (4 waves, 512 registers): parity of the variant, same-process A/B
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "prefill_kernel_variants" -p no:cacheprovider 2>&1 | tail -3
AB=$OUT/r03_prefill_x2_ab.txt; : > $AB
echo "# D=128, 16 032 keys: 128 = shipped (8 waves x 32 rows, 128-key tiles), 64 = the same at 64 keys, 264 = 4 waves x 2 x 32 rows, 64-key tiles" >> $AB
timeout 300 python tools/attn_bench.py --n 128 --B 64 --S 16032 --iters 10 --D 128 --hnd 1 --variants 128,264,64 --reps 3 2>&1 | grep -v amdgpu.ids >> $AB
echo "# D=128, 4 128 keys" >> $AB
timeout 300 python tools/attn_bench.py --n 128 --B 64 --S 4128 --iters 10 --D 128 --hnd 1 --variants 128,264 --reps 2 2>&1 | grep -v amdgpu.ids >> $AB
echo "# D=64, 16 032 keys: 64 = shipped, 264 / 328 = two tiles per wave at 64 / 128 keys" >> $AB
timeout 300 python tools/attn_bench.py --n 128 --B 64 --S 16032 --iters 10 --D 64 --hnd 1 --variants 64,264,328 --reps 3 2>&1 | grep -v amdgpu.ids >> $AB
grep -E "prefill view|^#|nan [1-9]" $AB
