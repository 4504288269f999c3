#!/bin/bash
# configs[4] (tumor 60x / normal 30x, k = 31) at full size on one GPU
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
B="python bench.py --inner --no-cpu-baseline --no-end-to-end --workload tn"
RFX_WGS_TRACE= timeout 1200 $B --steps 3 --warmup 2 $* 2>gpurun_out/r5_tn.err | tail -1 | tee gpurun_out/r5_tn.json | python scratch/r5_summ.py "TN"
tail -n 3 gpurun_out/r5_tn.err
