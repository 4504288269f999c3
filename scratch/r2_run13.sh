#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_overlap_gpu.py -x -q -m gpu -s > gpurun_out/t_ovl.log 2>&1; echo "rc=$?" >> gpurun_out/t_ovl.log
grep -E "wall times|passed|failed|rc=|Error|assert" gpurun_out/t_ovl.log | tail -12
