#!/bin/bash
# where does the GPU idle inside an iteration?  kernel trace of the bench + tools/iter_breakdown.py (gap table)
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_b
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d /tmp/prof_b -o bench -- \
    python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/r04_gaps_bench.log 2>&1
DB=$(find /tmp/prof_b -name "*.db" | head -1)
python tools/iter_breakdown.py $DB $OUT/r04_gaps_iter_breakdown.csv | grep "^#"
