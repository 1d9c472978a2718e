#!/bin/bash
# Round 5, GPU call 13: one TP-8 rank's compute of configs[3] (70B + 1B StreamingLLM draft) and configs[4] (Qwen2.5-32B self-speculation,
# fp8 KV) on the final tree; cfg3 with an fp8 KV cache as a secondary line (NOT the headline: the reference's KV is bf16)
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python3 bench.py --workload cfg4 --emulate-tp 8 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/r05_emulated_cfg4_tp8.log 2>&1
timeout 1200 python3 bench.py --workload cfg5 --emulate-tp 8 --steps 12 --warmup 4 --no-cpu-baseline --no-pmc > $OUT/r05_emulated_cfg5_tp8.log 2>&1
timeout 600 python3 bench.py --gpus 1 --kv-dtype fp8 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/r05_bench_cfg3_fp8kv.log 2>&1
for f in emulated_cfg4_tp8 emulated_cfg5_tp8 bench_cfg3_fp8kv; do grep '^{"metric"' $OUT/r05_$f.log > $OUT/r05_$f.json; python3 -c "
import json
l=json.load(open('$OUT/r05_$f.json')); print('$f', l['value'], l['ms_per_step'], l['autoregressive_ms_per_step'], l['speedup_vs_autoregressive'], l['roofline']['frac'], l['roofline']['avg_launch_ms'], l['prefill_s'], l['roofline']['kernel'])" || tail -5 $OUT/r05_$f.log; done
