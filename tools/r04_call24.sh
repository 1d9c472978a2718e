#!/bin/bash
# round 4, GPU call 24: cfg2 (Llama-3.1-8B self-speculation, StreamingLLM cache, B=32 x 8K) -- bench line + per-iteration kernel breakdown
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 bench.py --workload cfg2 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/r04_bench_cfg2_call24.log 2>&1
grep '^{"metric"' $OUT/r04_bench_cfg2_call24.log > $OUT/r04_bench_cfg2_call24.json
python3 -c "
import json; l=json.load(open('$OUT/r04_bench_cfg2_call24.json')); print('cfg2', l['value'], l['ms_per_step'], l.get('autoregressive_ms_per_step'), l['speedup_vs_autoregressive'], l['prefill_s'])"
rm -rf /tmp/prof_c2
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d /tmp/prof_c2 -o bench -- python3 bench.py --workload cfg2 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/r04_prof_cfg2.log 2>&1
DB=$(find /tmp/prof_c2 -name "*.db" | head -1)
[ -n "$DB" ] && python tools/iter_breakdown.py $DB $OUT/r04_bench_cfg2_iter_breakdown.csv > /dev/null
head -30 $OUT/r04_bench_cfg2_iter_breakdown.csv | cut -c1-200
