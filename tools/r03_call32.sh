#!/bin/bash
# GPU call 32: shader clock and socket power sampled (rocm-smi, every 2 s) while one kernel runs back to back:
# prefill attention on random data, on zero-filled data, and the verify (decode) attention kernel.
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
F=$OUT/r03_clock_power_samples.txt; : > $F
sample() {   # $1 = label, rest = attn_bench arguments
  label=$1; shift
  echo "# $label: python tools/attn_bench.py $*" >> $F
  timeout 200 python tools/attn_bench.py "$@" > /tmp/run.log 2>&1 &
  PID=$!
  for i in $(seq 1 14); do
    sleep 2
    kill -0 $PID 2>/dev/null || break
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Graphics Package Power" | tr '\n' ' ' >> $F
    echo >> $F
  done
  wait $PID
  grep -v amdgpu.ids /tmp/run.log | tail -2 >> $F
}
sample "idle-before" --n 4 --iters 1
sample "prefill attention, random data" --n 128 --B 64 --S 16032 --D 128 --hnd 1 --iters 6000
sample "prefill attention, zero-filled data" --n 128 --B 64 --S 16032 --D 128 --hnd 1 --iters 6000 --zero 1
sample "verify attention (4 query rows), random data" --n 4 --B 64 --S 16076 --D 128 --hnd 1 --iters 25000
cat $F
