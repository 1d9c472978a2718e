#!/bin/bash
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_gemm.py -q -p no:cacheprovider 2>&1 | tail -4
timeout 600 python tools/gemm_bench.py --waves 4 0 --only "8B w" > $OUT/r05c7_gemm_waves.txt 2>&1
grep -v amdgpu.ids $OUT/r05c7_gemm_waves.txt | cut -c1-200
