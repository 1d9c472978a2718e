#!/bin/bash
# GPU call 1 (round 4): first run of md_linear_block -- parity, then the same-process A/B against hipBLASLt / md_linear
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_blockgemm.py -q -x -p no:cacheprovider 2>&1 | tail -25 > $OUT/r04_blockgemm_tests.log
tail -5 $OUT/r04_blockgemm_tests.log
timeout 600 python tools/block_bench.py --blocks 256 192 --wnt 1 0 2>&1 | grep -v amdgpu.ids > $OUT/r04_block_ab_call1.txt
cat $OUT/r04_block_ab_call1.txt
