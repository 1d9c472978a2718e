#!/bin/bash
# round 3, GPU call 2: md_linear_fused -- parity, per-linear A/B, the whole suite with it on, bench A/B
set -u
OUT=$PWD/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_fused.py -q -p no:cacheprovider > $OUT/r03_fused_tests.log 2>&1
echo "fused tests rc=$?"; tail -4 $OUT/r03_fused_tests.log
timeout 900 python tools/fused_bench.py > $OUT/r03_fused_ab.txt 2>&1
echo "fused bench rc=$?"; tail -45 $OUT/r03_fused_ab.txt
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/r03_gpu_tests_b.log 2>&1
echo "suite rc=$?"; tail -6 $OUT/r03_gpu_tests_b.log
cp $OUT/parity_report.txt $OUT/r03_parity_report_b.txt 2>/dev/null
timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/r03_bench_fused.log 2>&1
MAGICDEC_FUSED=0 timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/r03_bench_nofused.log 2>&1
timeout 300 python3 bench.py --emulate-tp 8 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/r03_tp8_fused.log 2>&1
MAGICDEC_FUSED=0 timeout 300 python3 bench.py --emulate-tp 8 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/r03_tp8_nofused.log 2>&1
for f in r03_bench_fused r03_bench_nofused r03_tp8_fused r03_tp8_nofused; do echo "== $f"; grep '^{"metric"' $OUT/$f.log | python3 -c "
import json,sys
l=json.loads(sys.stdin.read()); print(l['value'], l['ms_per_step'], l['autoregressive_ms_per_step'], l['speedup_vs_autoregressive'], l['roofline']['avg_launch_ms'], l['roofline']['frac'])" || tail -5 $OUT/$f.log; done
