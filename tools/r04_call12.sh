#!/bin/bash
# round 4, GPU call 12: md_linear_fused with 2 x 2 tiles per workgroup -- parity (all fused tests + the bit-identity test), A/B
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fused.py -q -x -p no:cacheprovider 2>&1 | tail -3
timeout 600 python tools/fused_bench.py --only "1B/1" --tiles 1 2>&1 | grep -v amdgpu.ids > $OUT/r04_fused_tiles_ab.txt
timeout 600 python tools/fused_bench.py --only "8B/1 w" --tiles 1 2>&1 | grep -v "amdgpu.ids\|tuned" >> $OUT/r04_fused_tiles_ab.txt
cat $OUT/r04_fused_tiles_ab.txt
