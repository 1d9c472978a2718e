#!/bin/bash
# Kernel sequence around the iteration boundary (tools/iter_boundary.py) of the cfg3 bench and of one TP-8 rank's compute.
# usage: tools/boundary_trace.sh <tag>       outputs: gpurun_out/<tag>_boundary_{cfg3,tp8}.txt
set -u
TAG=$1
OUT=$PWD/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export MAGICDEC_BENCH_LAYOUT_AB=0
for W in cfg3 tp8; do
  rm -rf /tmp/prof_b
  ARGS=""; [ $W = tp8 ] && ARGS="--emulate-tp 8"
  timeout 600 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_b -o bench -- \
      python3 bench.py --gpus 1 --steps 16 --warmup 4 --no-cpu-baseline --no-pmc $ARGS > $OUT/${TAG}_boundary_$W.log 2>&1
  DB=$(find /tmp/prof_b -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/iter_boundary.py $DB 16 3 > $OUT/${TAG}_boundary_$W.txt
  cat $OUT/${TAG}_boundary_$W.txt
done
