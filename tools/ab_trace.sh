#!/bin/bash
# In-trace A/B of a policy switch: the cfg3 bench (short) under rocprofv3 with each setting of ONE environment variable, and the
# per-iteration kernel breakdown of each.   usage: tools/ab_trace.sh <tag> <ENVVAR> <value> [<value> ...]
# outputs: gpurun_out/<tag>_<value>_{line.json,iter_breakdown.csv}
set -u
TAG=$1; VAR=$2; shift 2
OUT=$PWD/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export MAGICDEC_BENCH_LAYOUT_AB=0
for V in "$@"; do
  rm -rf /tmp/prof_ab
  env $VAR=$V timeout ${AB_TIMEOUT:-600} rocprofv3 --kernel-trace --stats --output-format csv rocpd -d /tmp/prof_ab -o bench -- \
      python3 bench.py --gpus 1 --steps 16 --warmup 4 --no-cpu-baseline --no-pmc ${BENCH_ARGS:-} > $OUT/${TAG}_${V}.log 2>&1
  grep '^{"metric"' $OUT/${TAG}_${V}.log > $OUT/${TAG}_${V}_line.json
  DB=$(find /tmp/prof_ab -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/iter_breakdown.py $DB $OUT/${TAG}_${V}_iter_breakdown.csv > /dev/null
  python3 -c "
import json; l=json.load(open('$OUT/${TAG}_${V}_line.json')); print('$VAR=$V', l['value'], l['ms_per_step'], l['autoregressive_ms_per_step'])"
  head -24 $OUT/${TAG}_${V}_iter_breakdown.csv | cut -c1-100,180-260
done
