#!/bin/bash
# Round 5, GPU call 1: the new parity yardsticks / SnapKV table / peer-loss test on hardware, SnapKV by score distribution,
# the split-K target of md_linear on cfg2's shapes, cfg2 with peaked weights, the SnapKV select inside the bench trace.
set -u
OUT=$PWD/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/r05c1_gpu_tests.log 2>&1
echo "suite rc=$?"; tail -3 $OUT/r05c1_gpu_tests.log
cp $OUT/parity_report.txt $OUT/r05c1_parity_report.txt 2>/dev/null
grep -h "lockstep\|snapkv\] magnitudes\|allreduce\]" $OUT/r05c1_parity_report.txt | cut -c1-330
timeout 300 python tools/snapkv_bench.py --dist all > $OUT/r05c1_snapkv_bench.txt 2>&1; cat $OUT/r05c1_snapkv_bench.txt | tail -4
timeout 300 python tools/snapkv_bench.py --dist all --B 32 --S 8065 --D 128 --KH 8 >> $OUT/r05c1_snapkv_bench.txt 2>&1; tail -3 $OUT/r05c1_snapkv_bench.txt
timeout 400 python tools/gemm_bench.py --blocks 224 256 512 --only c2 > $OUT/r05c1_gemm_blocks.txt 2>&1; cat $OUT/r05c1_gemm_blocks.txt | tail -8
timeout 600 python3 bench.py --workload cfg2 --weights peaked --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/r05c1_bench_cfg2_peaked.log 2>&1
grep '^{"metric"' $OUT/r05c1_bench_cfg2_peaked.log > $OUT/r05c1_bench_cfg2_peaked.json
python3 - <<PY
import json
l=json.load(open('$OUT/r05c1_bench_cfg2_peaked.json'))
print('cfg2 peaked', l['value'], l['ms_per_step'], l['autoregressive_ms_per_step'], l['speedup_vs_autoregressive'], l['measured_acceptance_run'], l['speedup_condition'])
PY
rm -rf /tmp/prof_b
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d /tmp/prof_b -o bench -- \
    python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/r05c1_prof_bench.log 2>&1
echo "bench under rocprofv3 rc=$?"
KS=$(find /tmp/prof_b -name "*kernel_stats.csv" | head -1)
[ -n "$KS" ] && head -70 "$KS" > $OUT/r05c1_bench_cfg3_kernel_stats.csv
DB=$(find /tmp/prof_b -name "*.db" | head -1)
[ -n "$DB" ] && python tools/iter_breakdown.py $DB $OUT/r05c1_bench_cfg3_iter_breakdown.csv > /dev/null
grep '^{"metric"' $OUT/r05c1_prof_bench.log > $OUT/r05c1_bench_cfg3_under_rocprofv3.json
grep -i "snapkv" $OUT/r05c1_bench_cfg3_kernel_stats.csv | cut -c1-200
python3 - <<PY
import json
l=json.load(open('$OUT/r05c1_bench_cfg3_under_rocprofv3.json'))
print('cfg3 (rocprof)', l['value'], l['ms_per_step'], l['autoregressive_ms_per_step'], l['speedup_vs_autoregressive'], l['roofline']['frac'], l['prefill_s'], l['speedup_condition'])
PY
