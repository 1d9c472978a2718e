#!/bin/bash
# Round 5, GPU call 2: the whole GPU suite with the new yardsticks (no -x), the round's BASELINE of one TP-8 rank.
set -u
OUT=$PWD/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/r05c2_gpu_tests.log 2>&1
echo "suite rc=$?"; tail -3 $OUT/r05c2_gpu_tests.log
cp $OUT/parity_report.txt $OUT/r05c2_parity_report.txt 2>/dev/null
grep -h "lockstep\|snapkv\] magnitudes\|allreduce\]" $OUT/r05c2_parity_report.txt | cut -c1-360
python3 bench.py --emulate-tp 8 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/r05c2_emulated_tp8.log 2>&1
MAGICDEC_ONESHOT_AR=1 python3 bench.py --emulate-tp 8 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/r05c2_emulated_tp8_fused_ar.log 2>&1
for f in emulated_tp8 emulated_tp8_fused_ar; do grep '^{"metric"' $OUT/r05c2_$f.log > $OUT/r05c2_$f.json; python3 -c "
import json
l=json.load(open('$OUT/r05c2_$f.json')); print('$f', l['value'], l['ms_per_step'], l['autoregressive_ms_per_step'], l['speedup_vs_autoregressive'], l['roofline']['frac'])"; done
