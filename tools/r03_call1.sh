#!/bin/bash
# round 3, GPU call 1: (a) the whole -m gpu suite with HND as the Engine default, (b) rocprofv3 kernel trace of the
# exact driver bench command (HND), (c) per-iteration breakdown of one TP8 rank's compute, (d) whole-iteration graph A/B
set -u
OUT=$PWD/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
MAGICDEC_KV_LAYOUT=HND timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/r03_gpu_tests_hnd.log 2>&1
echo "HND suite rc=$?"; tail -5 $OUT/r03_gpu_tests_hnd.log
cp $OUT/parity_report.txt $OUT/r03_parity_report_hnd.txt 2>/dev/null
rm -rf /tmp/prof_b
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d /tmp/prof_b -o bench -- \
    python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/r03_prof_bench.log 2>&1
echo "bench under rocprofv3 rc=$?"
KS=$(find /tmp/prof_b -name "*kernel_stats.csv" | head -1)
[ -n "$KS" ] && head -60 "$KS" > $OUT/r03_bench_cfg3_kernel_stats.csv
DB=$(find /tmp/prof_b -name "*.db" | head -1)
[ -n "$DB" ] && python tools/iter_breakdown.py $DB $OUT/r03_bench_cfg3_iter_breakdown.csv > /dev/null
tail -1 $OUT/r03_prof_bench.log | cut -c1-600
rm -rf /tmp/prof_t
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d /tmp/prof_t -o bench -- \
    python3 bench.py --emulate-tp 8 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/r03_prof_tp8.log 2>&1
echo "tp8 under rocprofv3 rc=$?"
DB=$(find /tmp/prof_t -name "*.db" | head -1)
[ -n "$DB" ] && python tools/iter_breakdown.py $DB $OUT/r03_emulated_tp8_iter_breakdown_before.csv > /dev/null
timeout 300 python3 bench.py --emulate-tp 8 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/r03_tp8_stepgraphs.log 2>&1
MAGICDEC_ITER_GRAPH=1 timeout 300 python3 bench.py --emulate-tp 8 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/r03_tp8_itergraph.log 2>&1
for f in r03_tp8_stepgraphs r03_tp8_itergraph; do echo "== $f"; tail -1 $OUT/$f.log | cut -c1-300; done
head -30 $OUT/r03_emulated_tp8_iter_breakdown_before.csv | cut -c1-200
