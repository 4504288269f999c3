#!/bin/bash
# round 6, first call: parity of the count paths with the flush split out of k_msp_leaf (k_surv_place), then 1 Gb and W timings.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_scale_gpu.py -x -q -m gpu 2>&1 | tail -8
B="python bench.py --inner --no-cpu-baseline --no-end-to-end"
timeout 600 $B --genome 1000000000 --passes 2 --steps 3 --warmup 2 2>gpurun_out/r6a_1g.err | tail -1 | tee gpurun_out/r6a_1g.json | python scratch/r5_summ.py "1g"
timeout 900 $B --steps 4 --warmup 3 2>gpurun_out/r6a_w.err | tail -1 | tee gpurun_out/r6a_w.json | python scratch/r5_summ.py "W"
tail -3 gpurun_out/r6a_*.err
