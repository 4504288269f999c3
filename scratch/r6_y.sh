#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_scale_gpu.py -x -q -m gpu -k "run_map or replay or hashed or shard or trio" 2>&1 | tail -3
B="python bench.py --inner --no-cpu-baseline --no-end-to-end"
for e in "" 1; do
  unset RFX_REPLAY_OLD; [ -n "$e" ] && export RFX_REPLAY_OLD=1
  timeout 600 $B --genome 1000000000 --passes 2 --steps 3 --warmup 2 2>gpurun_out/r6y_1g_$e.err | tail -1 | python scratch/r5_summ.py "1g old=$e"
done
unset RFX_REPLAY_OLD
timeout 900 $B --steps 4 --warmup 3 2>gpurun_out/r6y_w.err | tail -1 | python scratch/r5_summ.py "W"
