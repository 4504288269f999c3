#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1200 python -m pytest tests/test_scale_gpu.py -x -q -m gpu -k "filter" 2>&1 | tail -3
B="python bench.py --inner --no-cpu-baseline --no-end-to-end --no-check"
timeout 600 $B --genome 1000000000 --passes 2 --steps 2 --warmup 1 2>gpurun_out/r6g_1g.err | tail -1 | python scratch/r5_summ.py "1g" | head -1
timeout 900 $B --steps 3 --warmup 2 2>gpurun_out/r6g_w.err | tail -1 | python scratch/r5_summ.py "W" | head -1
