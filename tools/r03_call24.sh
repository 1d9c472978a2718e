#!/bin/bash
# NOTE: provenance only -- ran at commits 7a76d02 / 40f68c7 / d4fa652, where the knob values >= 1000 selected the ping-pong
# kernel (prefill32p_attn_kernel, removed afterwards: docs/DESIGN_r1_r5_lab_notes.md 3.5); on later commits those values fall back to the rule.
# GPU call 24: where the prefill kernel's time goes -- timing ablations (results wrong by construction), zero-filled data
# (clock / power check), clocks and power sampled while the kernel runs.
set -u
OUT=$PWD/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
AB=$OUT/r03_prefill_ablation.txt
: > $AB
echo "# D=128, 16 032 keys: 128 = shipped, 130 = no softmax (P = bf16(S)), 131 = no P.V MFMAs, 1064 = ping-pong, 5064 = ping-pong without softmax" >> $AB
timeout 300 python tools/attn_bench.py --n 128 --B 64 --S 16032 --iters 10 --D 128 --hnd 1 --variants 128,130,131,1064,5064 --reps 2 2>&1 | grep -v amdgpu.ids >> $AB
echo "# zero-filled q / K / V, shipped kernel and ping-pong" >> $AB
timeout 300 python tools/attn_bench.py --n 128 --B 64 --S 16032 --iters 10 --D 128 --hnd 1 --variants 128,1064 --reps 2 --zero 1 2>&1 | grep -v amdgpu.ids >> $AB
echo "# clocks / power while the shipped kernel runs back to back (rocm-smi, 3 samples 1 s apart)" >> $AB
rocm-smi --showmaxpower --showclocks --showpower >> $AB 2>&1
timeout 120 python tools/attn_bench.py --n 128 --B 64 --S 16032 --iters 3000 --D 128 --hnd 1 > /tmp/long.log 2>&1 &
PID=$!
sleep 9
for i in 1 2 3; do rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|Power|fclk" >> $AB; sleep 1; done
wait $PID
cat /tmp/long.log | grep -v amdgpu.ids >> $AB
grep -E "prefill view|^#|sclk|Power|Max" $AB
