#!/bin/bash
# round 4, final tree: the whole GPU suite, smoke(), the driver's bench command, cfg2
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/r04_final_gpu_tests.log 2>&1; echo "suite rc=$?"; tail -1 $OUT/r04_final_gpu_tests.log
cp $OUT/parity_report.txt $OUT/r04_final_parity_report.txt 2>/dev/null
python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/r04_final_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/r04_final_smoke.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r04_final_bench.log 2>&1; echo "bench rc=$?"
grep '^{"metric"' $OUT/r04_final_bench.log > $OUT/r04_final_bench.json
python3 bench.py --workload cfg2 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc 2>/dev/null | grep '^{"metric"' > $OUT/r04_final_bench_cfg2.json
for f in r04_final_bench r04_final_bench_cfg2; do python3 -c "
import json; l=json.load(open('$OUT/$f.json')); print('$f', l['value'], l['ms_per_step'], l['autoregressive_ms_per_step'], l['speedup_vs_autoregressive'], l['roofline']['frac'], l['prefill_s'])"; done
