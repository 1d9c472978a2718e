#!/bin/bash
# round 3, GPU call 7: md_linear_add_rmsnorm (split-K combine + residual add + RMSNorm in one launch): parity, suite, benches
set -u
OUT=$PWD/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/r03_gpu_tests_e.log 2>&1
echo "suite rc=$?"; tail -4 $OUT/r03_gpu_tests_e.log; grep -n "^FAILED\|^ERROR" $OUT/r03_gpu_tests_e.log | head
cp $OUT/parity_report.txt $OUT/r03_parity_report_e.txt 2>/dev/null
timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/r03_bench_e.log 2>&1
timeout 300 python3 bench.py --workload cfg2 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/r03_cfg2_e.log 2>&1
timeout 300 python3 bench.py --emulate-tp 8 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/r03_tp8_e.log 2>&1
for f in r03_bench_e r03_cfg2_e r03_tp8_e; do echo "== $f"; grep '^{"metric"' $OUT/$f.log | python3 -c "
import json,sys
l=json.loads(sys.stdin.read()); print(l['value'], l['ms_per_step'], l['autoregressive_ms_per_step'], l['speedup_vs_autoregressive'], l['roofline']['avg_launch_ms'], l['roofline']['frac'], l['prefill_s'])" || tail -5 $OUT/$f.log; done
