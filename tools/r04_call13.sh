#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_engine.py -q -x -p no:cacheprovider 2>&1 | tail -3
