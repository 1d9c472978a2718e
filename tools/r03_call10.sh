#!/bin/bash
# one rank's compute of the TP8 configurations cfg4 (70B + 1B StreamingLLM draft, B=32 x 32K) and cfg5 (Qwen2.5-32B self-spec,
# SnapKV, fp8 KV, B=128 x 64K) with the round-3 kernels
set -u
OUT=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 700 python3 bench.py --workload cfg4 --emulate-tp 8 --steps 12 --warmup 4 --no-cpu-baseline --no-pmc > $OUT/r03_emulated_cfg4_tp8.log 2>&1
timeout 700 python3 bench.py --workload cfg5 --emulate-tp 8 --steps 12 --warmup 4 --no-cpu-baseline --no-pmc > $OUT/r03_emulated_cfg5_tp8.log 2>&1
for f in r03_emulated_cfg4_tp8 r03_emulated_cfg5_tp8; do echo "== $f"; grep '^{"metric"' $OUT/$f.log > $OUT/$f.json; python3 -c "
import json,sys
l=json.load(open('$OUT/$f.json')); print(l['config']['workload']); print(l['value'], l['ms_per_step'], l['autoregressive_ms_per_step'], l['speedup_vs_autoregressive'], l['roofline']['avg_launch_ms'], l['roofline']['frac'], l['prefill_s'], l['kv_cache_dtype'])" || tail -5 $OUT/$f.log; done
