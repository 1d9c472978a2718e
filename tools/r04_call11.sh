#!/bin/bash
# round 4, GPU call 11: the new full-size shard tests (cfg4 / cfg5 kernels), then the whole GPU suite with the tightened flip gate
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -p no:cacheprovider 2>&1 | tail -15
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -5
cp $OUT/parity_report.txt $OUT/r04_parity_report_call11.txt 2>/dev/null
