#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_scale_gpu.py tests/test_cli_gpu.py -x -q -k "not slice_properties" 2>&1 | tail -6
bash scratch/cli_w_sample.sh 2>&1 | grep -v "^\[load_fd" | tee gpurun_out/r5_cli_w_sample.txt | head -40
