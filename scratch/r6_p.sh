#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
RFX_STAGE_DEBUG=1 timeout 600 python bench.py --inner --workload tn --genome 500000000 --steps 1 --warmup 0 --no-check --no-cpu-baseline --no-end-to-end 2>gpurun_out/r6p_tn.err | tail -1 | python scratch/r5_summ.py "TN 0.5g" | head -2
grep "rfx stage" gpurun_out/r6p_tn.err | head -8
RFX_STAGE_DEBUG=1 timeout 600 python bench.py --inner --genome 500000000 --steps 1 --warmup 0 --no-check --no-cpu-baseline --no-end-to-end 2>gpurun_out/r6p_w.err | tail -1 | python scratch/r5_summ.py "W 0.5g" | head -2
grep "rfx stage" gpurun_out/r6p_w.err | head -4
