#!/bin/bash
# Produces the evidence committed under profiles/ (run on the GPU box via gpurun):
#   1. the default bench.py line (roofline with live PMC traffic, cpu_baseline)
#   2. rocprofv3 kernel trace + stats of the same command (kernel_stats csv + per-iteration breakdown)
#   3. cfg2 (self-speculation, StreamingLLM cache) and one TP8 rank's compute (--emulate-tp 8), with and without the
#      fused xGMI all-reduce + add + RMSNorm kernel
# usage: tools/profile_round.sh <tag>      outputs: gpurun_out/<tag>_*
set -u
TAG=${1:-r02}
OUT=$PWD/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_cfg3.log 2>&1
tail -1 $OUT/${TAG}_bench_cfg3.log > $OUT/${TAG}_bench_cfg3.json
rm -rf /tmp/prof_b
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d /tmp/prof_b -o bench -- \
    python bench.py --no-cpu-baseline --no-pmc --steps 24 --warmup 4 > $OUT/${TAG}_prof_bench.log 2>&1
echo "bench under rocprofv3 rc=$?"
KS=$(find /tmp/prof_b -name "*kernel_stats.csv" | head -1)
[ -n "$KS" ] && head -40 "$KS" > $OUT/${TAG}_bench_cfg3_kernel_stats.csv
DB=$(find /tmp/prof_b -name "*.db" | head -1)
[ -n "$DB" ] && python tools/iter_breakdown.py $DB $OUT/${TAG}_bench_cfg3_iter_breakdown.csv > /dev/null
python bench.py --workload cfg2 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/${TAG}_bench_cfg2.log 2>&1
MAGICDEC_GEMM=lib python bench.py --workload cfg2 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/${TAG}_bench_cfg2_lib.log 2>&1
python bench.py --emulate-tp 8 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/${TAG}_emulated_tp8.log 2>&1
MAGICDEC_ONESHOT_AR=1 python bench.py --emulate-tp 8 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/${TAG}_emulated_tp8_fused_ar.log 2>&1
for f in bench_cfg3 bench_cfg2 bench_cfg2_lib emulated_tp8 emulated_tp8_fused_ar; do echo "== $f"; tail -1 $OUT/${TAG}_$f.log | cut -c1-400; done
head -14 $OUT/${TAG}_bench_cfg3_iter_breakdown.csv | cut -c1-170
