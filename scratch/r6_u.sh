#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
RFX_WGS_TRACE=1 timeout 900 python bench.py --inner --steps 2 --warmup 2 --no-check --no-cpu-baseline --no-end-to-end > gpurun_out/r6u.log 2>gpurun_out/r6u.err
grep "^\[wgs\]" gpurun_out/r6u.log | tail -28
tail -1 gpurun_out/r6u.log | python scratch/r5_summ.py "W traced" | head -3
