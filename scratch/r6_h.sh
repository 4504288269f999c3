#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1200 python -m pytest tests/test_scale_gpu.py -x -q -m gpu -k "filter" 2>&1 | tail -2
B="python bench.py --inner --no-cpu-baseline --no-end-to-end --no-check"
RFX_LIB=$GRAFT_REPO_ROOT/scratch/variants/librufus_fpnopush.so timeout 900 $B --steps 2 --warmup 2 2>gpurun_out/r6h_w1.err | tail -1 | python scratch/r5_summ.py "W nopush" | head -1
RFX_FILTER_PAIR_BITS=2 timeout 900 $B --steps 2 --warmup 2 2>gpurun_out/r6h_w2.err | tail -1 | python scratch/r5_summ.py "W bits=2" | head -1
