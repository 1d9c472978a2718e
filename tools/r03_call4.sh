#!/bin/bash
# round 3, GPU call 4: fused-kernel rope fix (no FMA contraction), measured policy, prefill KT by head dim: whole suite + benches
set -u
OUT=$PWD/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/r03_gpu_tests_c.log 2>&1
echo "suite rc=$?"; tail -8 $OUT/r03_gpu_tests_c.log; grep -n "q differs" $OUT/r03_gpu_tests_c.log | cut -c1-600 | head -3
cp $OUT/parity_report.txt $OUT/r03_parity_report_c.txt 2>/dev/null
timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/r03_bench_c.log 2>&1
timeout 300 python3 bench.py --emulate-tp 8 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/r03_tp8_c.log 2>&1
MAGICDEC_ONESHOT_AR=1 timeout 300 python3 bench.py --emulate-tp 8 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/r03_tp8_c_ar.log 2>&1
timeout 300 python3 bench.py --workload cfg2 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/r03_cfg2_c.log 2>&1
for f in r03_bench_c r03_tp8_c r03_tp8_c_ar r03_cfg2_c; do echo "== $f"; grep '^{"metric"' $OUT/$f.log | python3 -c "
import json,sys
l=json.loads(sys.stdin.read()); print(l['value'], l['ms_per_step'], l['autoregressive_ms_per_step'], l['speedup_vs_autoregressive'], l['roofline']['avg_launch_ms'], l['roofline']['frac'], l['prefill_s'])" || tail -5 $OUT/$f.log; done
