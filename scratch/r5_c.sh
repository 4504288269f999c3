#!/bin/bash
# quick loop: run-map tests, 1 Gb bench (2 passes, maps), SQ counters
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_scale_gpu.py -x -q -k "run_map or replay" 2>&1 | tail -5
B="python bench.py --inner --no-cpu-baseline --no-end-to-end"
timeout 600 $B --genome 1000000000 --passes 2 --steps 3 --warmup 2 2>gpurun_out/r5c_1g.err | tail -1 | tee gpurun_out/r5c_1g.json | python scratch/r5_summ.py "1g maps"
bash scratch/r5_sq.sh > gpurun_out/r5c_sq.log 2>&1
grep -A1 -E "^k_msp_(replay|part1|leaf)|^k_part2" gpurun_out/r5_sq.txt | cut -c1-330
