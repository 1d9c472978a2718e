#!/bin/bash
# HBM fetch bytes and L2 hit rate of the tile GEMM at the 1B w1|w3 shape ("x is re-read through L2, W comes
# from HBM once") -- rocprofv3 PMC passes over tools/fused_bench.py; separate passes, --kernel-trace only.
set -u
OUT=$PWD/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for C in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum"; do
  rm -rf /tmp/pmc_t
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_t -o pmc -- \
      python3 tools/fused_bench.py --only "1B/1 w13 M64" --pro 1 --iters 6 > /dev/null 2>&1
  F=$(find /tmp/pmc_t -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && python3 - "$F" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "tile_gemm" in k or "skinny_gemm" in k or "Cijk" in k:
        name = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
        acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (name, c), v in sorted(acc.items()):
    print(f"{name:72s} {c:22s} mean {sum(v)/len(v):14.1f} over {len(v)} dispatches")
PY
done
