#!/bin/bash
# rocprofv3 kernel trace + per-iteration breakdown of the driver bench command (cfg3, TP1) and of one TP8 rank's compute
# usage: tools/r03_prof.sh <tag>   -> gpurun_out/<tag>_*
set -u
TAG=${1:-r03}
OUT=$PWD/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_b /tmp/prof_t
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d /tmp/prof_b -o bench -- \
    python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/${TAG}_prof_bench.log 2>&1
echo "bench under rocprofv3 rc=$?"
KS=$(find /tmp/prof_b -name "*kernel_stats.csv" | head -1)
[ -n "$KS" ] && head -60 "$KS" > $OUT/${TAG}_bench_cfg3_kernel_stats.csv
DB=$(find /tmp/prof_b -name "*.db" | head -1)
[ -n "$DB" ] && python tools/iter_breakdown.py $DB $OUT/${TAG}_bench_cfg3_iter_breakdown.csv > /dev/null
grep '^{"metric"' $OUT/${TAG}_prof_bench.log > $OUT/${TAG}_bench_cfg3_under_rocprofv3.json
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d /tmp/prof_t -o bench -- \
    python3 bench.py --emulate-tp 8 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/${TAG}_prof_tp8.log 2>&1
echo "tp8 under rocprofv3 rc=$?"
DB=$(find /tmp/prof_t -name "*.db" | head -1)
[ -n "$DB" ] && python tools/iter_breakdown.py $DB $OUT/${TAG}_emulated_tp8_iter_breakdown.csv > /dev/null
head -34 $OUT/${TAG}_bench_cfg3_iter_breakdown.csv | cut -c1-190
head -40 $OUT/${TAG}_emulated_tp8_iter_breakdown.csv | cut -c1-190
