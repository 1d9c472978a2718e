#!/bin/bash
# round 4: md_linear_block integrated -- its tests, the engine lock-step tests, the final A/B table, and a cfg3 bench line
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_blockgemm.py tests/test_gpu_engine.py tests/test_gpu_gemm.py tests/test_gpu_fused.py -q -x -p no:cacheprovider 2>&1 | tail -4
timeout 600 python tools/block_bench.py --blocks 256 --wnt 1 2>&1 | grep -v amdgpu.ids > $OUT/r04_block_ab_final.txt
cat $OUT/r04_block_ab_final.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/r04_bench_call10.log 2>&1; tail -1 $OUT/r04_bench_call10.log | cut -c1-1500
MAGICDEC_BLOCK=0 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/r04_bench_call10_noblock.log 2>&1; tail -1 $OUT/r04_bench_call10_noblock.log | cut -c1-600
