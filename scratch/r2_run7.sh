#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_cli_gpu.py -x -q -m gpu -k "cli" > gpurun_out/t_cli.log 2>&1; echo "rc=$?" >> gpurun_out/t_cli.log
tail -n 5 gpurun_out/t_cli.log
timeout 200 bash scratch/cli_scale.sh 4000000 40000000 64 > gpurun_out/cli_scale_small.log 2>&1; echo "rc=$?" >> gpurun_out/cli_scale_small.log
cat gpurun_out/cli_scale_small.log
timeout 400 bash scratch/cli_scale.sh 32000000 320000000 64 > gpurun_out/cli_scale.log 2>&1; echo "rc=$?" >> gpurun_out/cli_scale.log
cat gpurun_out/cli_scale.log
rm -rf /dev/shm/rfx_cli_scale
