#!/bin/bash
# the TP-8 rank's kernel breakdown of the FINAL tree + the whole GPU suite + smoke
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_t
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d /tmp/prof_t -o bench -- \
    python3 bench.py --emulate-tp 8 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/r05_prof_tp8.log 2>&1
DB=$(find /tmp/prof_t -name "*.db" | head -1)
[ -n "$DB" ] && python tools/iter_breakdown.py $DB $OUT/r05_emulated_tp8_iter_breakdown.csv > /dev/null
sed -n 1,3p $OUT/r05_emulated_tp8_iter_breakdown.csv
awk -F, 'NR>3 && $2+0>0 {n+=$2} END{print "launches/iter", n}' $OUT/r05_emulated_tp8_iter_breakdown.csv
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/r05_gpu_tests.log 2>&1
echo "suite rc=$?"; tail -1 $OUT/r05_gpu_tests.log
cp $OUT/parity_report.txt $OUT/r05_parity_report.txt 2>/dev/null
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
