#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1200 python -m pytest tests/test_scale_gpu.py -x -q -m gpu -k "filter" 2>&1 | tail -3
B="python bench.py --inner --no-cpu-baseline --no-end-to-end --no-check"
for v in "" fpnopush; do
  L=""; [ -n "$v" ] && L="$GRAFT_REPO_ROOT/scratch/variants/librufus_$v.so"
  RFX_LIB=$L timeout 600 $B --genome 1000000000 --passes 2 --steps 2 --warmup 1 2>gpurun_out/r6f_$v.err | tail -1 | python scratch/r5_summ.py "1g $v" | head -1
done
timeout 900 $B --steps 3 --warmup 2 2>gpurun_out/r6f_w.err | tail -1 | python scratch/r5_summ.py "W" | head -1
