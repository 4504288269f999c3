#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
bash scratch/r6_full.sh
export RFX_FUZZ_SEEDS=50000-52000
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "three_count_paths" 2>&1 | tail -n 2 | tee gpurun_out/r6_fuzz_k.txt
