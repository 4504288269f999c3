#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_scale_gpu.py -x -q -k "fixed_capacity or run_map or replay or shard_passes or several_devices" 2>&1 | tail -15
B="python bench.py --inner --no-cpu-baseline --no-end-to-end"
timeout 600 $B --genome 1000000000 --passes 2 --steps 3 --warmup 2 2>gpurun_out/r5e_1g.err | tail -1 | tee gpurun_out/r5e_1g.json | python scratch/r5_summ.py "1g maps"
timeout 900 $B --steps 4 --warmup 3 2>gpurun_out/r5e_w.err | tail -1 | tee gpurun_out/r5e_w.json | python scratch/r5_summ.py "W maps"
