#!/bin/bash
# Produces the rocprofv3 evidence committed under profiles/ (run on the GPU box via gpurun):
#   1. kernel trace + stats of the default bench.py command (kernel_stats csv + per-iteration breakdown)
#   2. PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, --kernel-trace only) of the verify-attention shape
# usage: tools/profile_round.sh <tag>      outputs: gpurun_out/<tag>_*
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_b /tmp/prof_f /tmp/prof_w
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d /tmp/prof_b -o bench -- \
    python bench.py --no-cpu-baseline --steps 24 --warmup 4 > $OUT/${TAG}_prof_bench.log 2>&1
echo "bench under rocprofv3 rc=$?"
KS=$(find /tmp/prof_b -name "*kernel_stats.csv" | head -1)
[ -n "$KS" ] && head -40 "$KS" > $OUT/${TAG}_bench_cfg3_kernel_stats.csv
DB=$(find /tmp/prof_b -name "*.db" | head -1)
[ -n "$DB" ] && python tools/iter_breakdown.py $DB $OUT/${TAG}_bench_cfg3_iter_breakdown.csv > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
    d=/tmp/prof_$(echo $c | cut -c1 | tr A-Z a-z)
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o pmc -- \
        python tools/attn_bench.py --iters 6 > $OUT/${TAG}_pmc_$c.log 2>&1
    CC=$(find $d -name "*counter_collection.csv" | head -1)
    [ -n "$CC" ] && python - "$CC" $c <<'PY' >> $OUT/${TAG}_pmc_summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if r.get("Counter_Name") == sys.argv[2]:
        agg[r["Kernel_Name"][:90]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if "paged_attn" in k or "attn_merge" in k:
        print(f"{sys.argv[2]} kernel={k} launches={len(v)} mean={sum(v) / len(v):.2f} min={min(v):.2f} max={max(v):.2f}")
PY
done
python tools/attn_bench.py --iters 40 >> $OUT/${TAG}_pmc_summary.txt 2>&1
cat $OUT/${TAG}_pmc_summary.txt
tail -1 $OUT/${TAG}_prof_bench.log | cut -c1-300
head -12 $OUT/${TAG}_bench_cfg3_iter_breakdown.csv | cut -c1-170
