#!/bin/bash
# whole GPU suite + W bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
B="python bench.py --inner --no-cpu-baseline --no-end-to-end"
timeout 900 $B --steps 6 --warmup 3 2>gpurun_out/r5f_w.err | tail -1 | tee gpurun_out/r5f_w.json | python scratch/r5_summ.py "W maps"
