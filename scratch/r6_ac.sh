#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_scale_gpu.py tests/test_gpu_parity.py -x -q -m gpu -k "trio or filter or full_size or bench or pulled" 2>&1 | tail -3
B="python bench.py --inner --no-cpu-baseline --no-end-to-end"
timeout 900 $B --steps 4 --warmup 3 2>gpurun_out/r6ac.err | tail -1 | tee gpurun_out/r6ac.json | python scratch/r5_summ.py "W" | head -1
RFX_WGS_TRACE=1 timeout 900 $B --steps 1 --warmup 2 --no-check 2>/dev/null | grep "\[wgs\] filter\|hash list" | tail -2
