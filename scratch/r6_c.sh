#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
B="python bench.py --inner --no-cpu-baseline --no-end-to-end"
for v in "" sp256 sp256n4 sp1024 ilp4 ilp2; do
  L=""; [ -n "$v" ] && L="$GRAFT_REPO_ROOT/scratch/variants/librufus_$v.so"
  RFX_LIB=$L timeout 600 $B --genome 1000000000 --passes 2 --steps 3 --warmup 2 2>gpurun_out/r6c_1g_$v.err | tail -1 | tee gpurun_out/r6c_1g_$v.json | python scratch/r5_summ.py "1g $v" | head -2
done
