#!/bin/bash
# round 4: SnapKV select with the exp table -- parity (every snapkv test incl. the reference fixtures and full-size ones), timing, per-kernel times
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -k "snapkv" -p no:cacheprovider 2>&1 | tail -6
T=$OUT/r04_snapkv_bench.txt; : > $T
timeout 300 python tools/snapkv_bench.py 2>&1 | grep -v amdgpu.ids >> $T
timeout 300 python tools/snapkv_bench.py --B 16 --KH 1 --g 5 --D 128 --S 65440 --fp8 1 2>&1 | grep -v amdgpu.ids >> $T
timeout 300 python tools/snapkv_bench.py --B 32 --KH 8 --g 4 --D 128 --S 8032 2>&1 | grep -v amdgpu.ids >> $T
cat $T
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_snap -o snap -- python $GRAFT_REPO_ROOT/tools/snapkv_bench.py > /dev/null 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob("/tmp/prof_snap/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:9]:
        if "snapkv" in r["Name"]: print(r["Name"][:80], r["Calls"], r["AverageNs"])
PY
