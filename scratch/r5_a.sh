#!/bin/bash
# round 5, first call: the run-map tests, then maps on / off at 1 Gb (2 passes) and at W.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_scale_gpu.py -x -q -k "run_map or replay or hashed_once or shard_passes" 2>&1 | tail -15
B="python bench.py --inner --no-cpu-baseline --no-end-to-end"
for e in "" 1; do
  RFX_BENCH_NO_EARLY=$e timeout 600 $B --genome 1000000000 --passes 2 --steps 3 --warmup 2 2>gpurun_out/r5a_1g_$e.err | tail -1 | tee gpurun_out/r5a_1g_$e.json | python scratch/r5_summ.py "1g noearly=$e"
done
for e in "" 1; do
  RFX_BENCH_NO_EARLY=$e timeout 900 $B --steps 4 --warmup 3 2>gpurun_out/r5a_w_$e.err | tail -1 | tee gpurun_out/r5a_w_$e.json | python scratch/r5_summ.py "W noearly=$e"
done
tail -5 gpurun_out/r5a_*.err
