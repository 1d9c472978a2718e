#!/bin/bash
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 500 python tools/fused_bench.py --only "8B/8" --tiles 1 > $OUT/r05c4_fused_tp8_tiles.txt 2>&1
cat $OUT/r05c4_fused_tp8_tiles.txt | cut -c1-200
