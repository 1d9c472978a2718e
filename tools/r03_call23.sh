#!/bin/bash
# NOTE: provenance only -- ran at commits 7a76d02 / 40f68c7 / d4fa652, where the knob values >= 1000 selected the ping-pong
# kernel (prefill32p_attn_kernel, removed afterwards: docs/DESIGN_r1_r5_lab_notes.md 3.5); on later commits those values fall back to the rule.
# GPU call 23: the ping-pong prefill kernel (prefill32p_attn_kernel): parity of every variant, same-box A/B against the
# shipped 32x32 kernels, SQ counters of the old 16x16 kernel / shipped / ping-pong.
set -u
OUT=$PWD/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "prefill_kernel_variants or page_sizes or paged_attention" \
    -p no:cacheprovider > $OUT/r03_call23_tests.log 2>&1
echo "tests rc=$?"; tail -4 $OUT/r03_call23_tests.log
AB=$OUT/r03_prefill_pingpong_ab2.txt
: > $AB
echo "# D=128, B=64, 128 tokens x 32 heads vs 16 032 keys (HND pages), one process, 2 repetitions" >> $AB
timeout 300 python tools/attn_bench.py --n 128 --B 64 --S 16032 --iters 10 --D 128 --hnd 1 --variants 128,129,1064,2064,3064,4064 --reps 2 2>&1 | grep -v amdgpu.ids >> $AB
echo "# D=128, 4 128 keys" >> $AB
timeout 300 python tools/attn_bench.py --n 128 --B 64 --S 4128 --iters 10 --D 128 --hnd 1 --variants 128,1064,2064 --reps 2 2>&1 | grep -v amdgpu.ids >> $AB
echo "# D=64, 16 032 keys" >> $AB
timeout 300 python tools/attn_bench.py --n 128 --B 64 --S 16032 --iters 10 --D 64 --hnd 1 --variants 64,1064,2064,1128,2128 --reps 2 2>&1 | grep -v amdgpu.ids >> $AB
grep -E "prefill view|vs the first|^#" $AB
# SQ counters: which exist on this box, then one pass per kernel variant
rocprofv3 -L > $OUT/r03_call23_counters_avail.txt 2>&1
WANT="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"
HAVE=""
for c in $WANT; do grep -qw "$c" $OUT/r03_call23_counters_avail.txt && HAVE="$HAVE $c"; done
echo "counters available: $HAVE"
PM=$OUT/r03_prefill_pmc2.txt
: > $PM
for v in 128 1064; do
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
    G=""
    for c in $grp; do case " $HAVE " in *" $c "*) G="$G $c";; esac; done
    [ -z "$G" ] && continue
    rm -rf /tmp/pmc_p
    timeout 200 rocprofv3 --pmc $G --kernel-trace --output-format csv -d /tmp/pmc_p -o pmc -- \
        python tools/attn_bench.py --n 128 --B 64 --S 16032 --iters 3 --D 128 --hnd 1 --mfma32 $v > /tmp/pmc_p.log 2>&1
    F=$(find /tmp/pmc_p -name "*counter_collection.csv" | head -1)
    [ -n "$F" ] && python3 - "$F" "$v" >> $PM <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "prefill" in r["Kernel_Name"]:
        acc[(r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"knob={sys.argv[2]} {k} {c} mean_per_launch={sum(v)/len(v):.6g} n={len(v)}")
PY
  done
done
cat $PM
