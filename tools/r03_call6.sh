#!/bin/bash
# round 3, GPU call 6: deferred RMSNorm (norm fused into the consuming linear): parity, suite, bench A/B
set -u
OUT=$PWD/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_fused.py -q -p no:cacheprovider > $OUT/r03_fused_tests3.log 2>&1
echo "fused tests rc=$?"; tail -5 $OUT/r03_fused_tests3.log; grep "deferred norm" $OUT/r03_fused_tests3.log | head -12
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_gpu_fused.py > $OUT/r03_gpu_tests_d.log 2>&1
echo "suite rc=$?"; tail -4 $OUT/r03_gpu_tests_d.log
cp $OUT/parity_report.txt $OUT/r03_parity_report_d.txt 2>/dev/null
timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/r03_bench_d.log 2>&1
timeout 300 python3 bench.py --emulate-tp 8 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/r03_tp8_d.log 2>&1
for f in r03_bench_d r03_tp8_d; do echo "== $f"; grep '^{"metric"' $OUT/$f.log | python3 -c "
import json,sys
l=json.loads(sys.stdin.read()); print(l['value'], l['ms_per_step'], l['autoregressive_ms_per_step'], l['speedup_vs_autoregressive'], l['roofline']['avg_launch_ms'], l['roofline']['frac'], l['prefill_s'])" || tail -5 $OUT/$f.log; done
