#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1200 python bench.py --workload tn --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end > gpurun_out/r6o_tn.log 2>gpurun_out/r6o_tn.err
tail -1 gpurun_out/r6o_tn.log > gpurun_out/r6o_tn.json
python scratch/r5_summ.py TN < gpurun_out/r6o_tn.json
tail -3 gpurun_out/r6o_tn.err
