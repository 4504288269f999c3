#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_scale_gpu.py tests/test_gpu_parity.py -x -q -m gpu -k "filter or trio or pulled or smoke" 2>&1 | tail -8
B="python bench.py --inner --no-cpu-baseline --no-end-to-end"
for e in "" 1; do
RFX_FILTER_NO_PAIR=$e timeout 600 $B --genome 1000000000 --passes 2 --steps 3 --warmup 2 2>gpurun_out/r6d_1g_$e.err | tail -1 | tee gpurun_out/r6d_1g_$e.json | python scratch/r5_summ.py "1g nopair=$e" | head -1
done
for b in 2 3; do
RFX_FILTER_PAIR_BITS=$b timeout 600 $B --genome 1000000000 --passes 2 --steps 3 --warmup 2 2>gpurun_out/r6d_1g_b$b.err | tail -1 | tee gpurun_out/r6d_1g_b$b.json | python scratch/r5_summ.py "1g bits=$b" | head -1
done
timeout 900 $B --steps 4 --warmup 3 2>gpurun_out/r6d_w.err | tail -1 | tee gpurun_out/r6d_w.json | python scratch/r5_summ.py "W" | head -1
