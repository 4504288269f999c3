#!/bin/bash
# leaf changes: parity tests, 1 Gb bench, W bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_scale_gpu.py -x -q -m gpu -k "not slice_properties" 2>&1 | tail -5
B="python bench.py --inner --no-cpu-baseline --no-end-to-end"
timeout 600 $B --genome 1000000000 --passes 2 --steps 3 --warmup 2 2>gpurun_out/r5g_1g.err | tail -1 | tee gpurun_out/r5g_1g.json | python scratch/r5_summ.py "1g"
[ -n "$W" ] && timeout 900 $B --steps 4 --warmup 3 2>gpurun_out/r5g_w.err | tail -1 | tee gpurun_out/r5g_w.json | python scratch/r5_summ.py "W"
