#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_scale_gpu.py -x -q -m gpu -k "run_map or replay or hashed or trio or full_size" 2>&1 | tail -3
B="python bench.py --inner --no-cpu-baseline --no-end-to-end"
for e in "" 1 "" 1; do
  unset RFX_NO_MAP_AHEAD; [ -n "$e" ] && export RFX_NO_MAP_AHEAD=1
  timeout 900 $B --steps 3 --warmup 3 2>gpurun_out/r6aa_$e.err | tail -1 | tee gpurun_out/r6aa_$e.json | python scratch/r5_summ.py "W noahead=$e" | cut -c1-250
  python3 -c "
import json; d=json.load(open('gpurun_out/r6aa_$e.json')); c=d['config']; print('   ahead', c.get('run_maps_hashed_ahead_on_the_second_stream_per_step'), 'count wall/sample', c.get('count_wall_ms_per_sample'))"
done
