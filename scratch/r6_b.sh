#!/bin/bash
# round 6: pipelined k_surv_place; leaf geometry variants now that the leaf needs 61 VGPRs
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
B="python bench.py --inner --no-cpu-baseline --no-end-to-end"
for v in "" b1024 b896; do
  L=""; [ -n "$v" ] && L="$GRAFT_REPO_ROOT/scratch/variants/librufus_$v.so"
  RFX_LIB=$L timeout 600 $B --genome 1000000000 --passes 2 --steps 3 --warmup 2 2>gpurun_out/r6b_1g_$v.err | tail -1 | tee gpurun_out/r6b_1g_$v.json | python scratch/r5_summ.py "1g $v"
done
timeout 900 $B --steps 4 --warmup 3 2>gpurun_out/r6b_w.err | tail -1 | tee gpurun_out/r6b_w.json | python scratch/r5_summ.py "W"
tail -n 3 gpurun_out/r6b_*.err
