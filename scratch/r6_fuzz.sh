#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export RFX_FUZZ_SEEDS=${RFX_FUZZ_SEEDS:-2000-8000}
timeout ${FUZZ_TIMEOUT:-3000} python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "random_configurations" 2>&1 | tail -n 6 | tee gpurun_out/r6_fuzz.txt
