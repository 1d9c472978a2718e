#!/bin/bash
# Round 5, GPU call 8: where the tree stands -- cfg3 (driver command without the CPU leg), cfg2 peaked, one TP-8 rank in both modes
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/r05c8_bench_cfg3.log 2>&1
python3 bench.py --workload cfg2 --weights peaked --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/r05c8_bench_cfg2.log 2>&1
python3 bench.py --emulate-tp 8 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/r05c8_emulated_tp8.log 2>&1
MAGICDEC_ONESHOT_AR=1 python3 bench.py --emulate-tp 8 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/r05c8_emulated_tp8_fused_ar.log 2>&1
for f in bench_cfg3 bench_cfg2 emulated_tp8 emulated_tp8_fused_ar; do grep '^{"metric"' $OUT/r05c8_$f.log > $OUT/r05c8_$f.json; python3 -c "
import json
l=json.load(open('$OUT/r05c8_$f.json')); print('$f', l['value'], l['ms_per_step'], l['autoregressive_ms_per_step'], l['speedup_vs_autoregressive'], l['roofline']['frac'], l['measured_acceptance_run'])"; done
