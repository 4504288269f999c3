#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export RFX_LIB=$GRAFT_REPO_ROOT/scratch/variants/librufus_nomul.so
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_scale_gpu.py -x -q -m gpu 2>&1 | tail -3
B="python bench.py --inner --no-cpu-baseline --no-end-to-end"
timeout 600 $B --genome 1000000000 --passes 2 --steps 3 --warmup 2 2>gpurun_out/r6v_1g.err | tail -1 | python scratch/r5_summ.py "1g nomul"
unset RFX_LIB
timeout 600 $B --genome 1000000000 --passes 2 --steps 3 --warmup 2 2>gpurun_out/r6v_1g0.err | tail -1 | python scratch/r5_summ.py "1g base"
