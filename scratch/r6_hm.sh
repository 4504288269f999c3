#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mask_only or filter" 2>&1 | tail -n 2
RFX_FUZZ_SEEDS=60000-61000 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "filter_matches_oracle" 2>&1 | tail -n 1
timeout 900 python scratch/r6_hm6.py 2>&1 | grep -v amdgpu.ids | tail -n 9 | cut -c1-135
timeout 900 python scratch/r6_hm5.py 30000000 300 31 2>&1 | grep -v amdgpu.ids | tail -n 9 | cut -c1-150
