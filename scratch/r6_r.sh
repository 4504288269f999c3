#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1200 python -m pytest tests/test_scale_gpu.py tests/test_gpu_parity.py -x -q -m gpu -k "tumor or shard_passes or capacity or dense" 2>&1 | tail -3
timeout 1200 python bench.py --workload tn --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end > gpurun_out/r6r_tn.log 2>gpurun_out/r6r_tn.err
tail -1 gpurun_out/r6r_tn.log > gpurun_out/r6r_tn.json
python scratch/r5_summ.py TN < gpurun_out/r6r_tn.json
