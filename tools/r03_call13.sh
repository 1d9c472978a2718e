#!/bin/bash
# tune hipBLASLt solutions for the PREFILL-sized GEMMs (M = B x 128 = 8192 and the 2048-row last chunk) of cfg3's models,
# then A/B the prefill time of the bench with the old and the extended table
set -u
OUT=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cp magicdec_amd/tuned/gemm_gfx950.csv $OUT/gemm_gfx950_r03.csv
timeout 900 python tools/tune_gemms.py --tp 1 --M 8192 2048 --out $OUT/gemm_gfx950_r03.csv > $OUT/r03_tune_prefill.log 2>&1
echo "tune rc=$?"; tail -3 $OUT/r03_tune_prefill.log; wc -l magicdec_amd/tuned/gemm_gfx950.csv $OUT/gemm_gfx950_r03.csv
timeout 300 python3 bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline --no-pmc > $OUT/r03_prefill_oldtable.log 2>&1
cp $OUT/gemm_gfx950_r03.csv magicdec_amd/tuned/gemm_gfx950.csv
timeout 300 python3 bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline --no-pmc > $OUT/r03_prefill_newtable.log 2>&1
for f in r03_prefill_oldtable r03_prefill_newtable; do echo "== $f"; grep '^{"metric"' $OUT/$f.log | python3 -c "
import json,sys
l=json.loads(sys.stdin.read()); print('prefill_s', l['prefill_s'], 'ms/iter', l['ms_per_step'])" || tail -3 $OUT/$f.log; done
