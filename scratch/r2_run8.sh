#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_cli_gpu.py tests/test_gpu_parity.py tests/test_overlap_gpu.py -x -q -m gpu > gpurun_out/t_all.log 2>&1; echo "rc=$?" >> gpurun_out/t_all.log
tail -n 8 gpurun_out/t_all.log
timeout 300 bash scratch/cli_filter_scale.sh 1000000 10000000 32 > gpurun_out/cli_filter_small.log 2>&1; echo "rc=$?" >> gpurun_out/cli_filter_small.log; cat gpurun_out/cli_filter_small.log
timeout 300 bash scratch/cli_filter_scale.sh 16000000 160000000 64 > gpurun_out/cli_filter.log 2>&1; echo "rc=$?" >> gpurun_out/cli_filter.log; cat gpurun_out/cli_filter.log
timeout 400 bash scratch/cli_scale.sh 32000000 320000000 64 > gpurun_out/cli_scale.log 2>&1; echo "rc=$?" >> gpurun_out/cli_scale.log; cat gpurun_out/cli_scale.log
rm -rf /dev/shm/rfx_cli_scale /dev/shm/rfx_filter_scale
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/b_full.json 2> gpurun_out/b_full.err; echo "rc=$?" >> gpurun_out/b_full.err
tail -c 300 gpurun_out/b_full.err; cut -c1-300 gpurun_out/b_full.json
