#!/bin/bash
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 120 tools/probes/l2_prefetch_probe.bin > $OUT/r04_l2_prefetch_probe.txt 2>&1; cat $OUT/r04_l2_prefetch_probe.txt
