#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_scale_gpu.py -x -q -m gpu -k "more_passes" > gpurun_out/t_retry.log 2>&1; echo "rc=$?" >> gpurun_out/t_retry.log
tail -n 12 gpurun_out/t_retry.log
