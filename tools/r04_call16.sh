#!/bin/bash
# round 4, GPU call 16: xGMI all-reduce with write-through publish (multi-process tests on one GPU), then one TP8 rank's
# compute in emulation: TP4 draft sub-group vs the draft replicated on every rank, RCCL vs the fused xGMI kernel
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_allreduce.py -q -x -p no:cacheprovider 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_engine.py -q -x -k "tp2_on_one_gpu or rccl_graphs_one_rank" -p no:cacheprovider 2>&1 | tail -3
for tag in "tp4draft:4:0" "tp4draft_fused_ar:4:1" "replicated_draft:0:0" "replicated_draft_fused_ar:0:1"; do
  IFS=: read name dtp ar <<< "$tag"
  MAGICDEC_ONESHOT_AR=$ar timeout 900 python bench.py --emulate-tp 8 --draft-tp $dtp --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/r04_emulated_tp8_$name.log 2>&1
  tail -1 $OUT/r04_emulated_tp8_$name.log > $OUT/r04_emulated_tp8_$name.json
  python - <<PY
import json
d = json.load(open("$OUT/r04_emulated_tp8_$name.json"))
print("$name", "ms_per_step", d["ms_per_step"], "tok/s", d["value"], "AR ms", d.get("autoregressive", {}).get("ms_per_step"), "speedup", d.get("speedup_vs_autoregressive"))
PY
done
