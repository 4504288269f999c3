#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
echo "=== the new test on the build with the 64-bit atomics (must fail)"
RFX_LIB=$PWD/scratch/variants/librufus_fpx1.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "mask_only" 2>&1 | tail -n 6 | cut -c1-300
echo "=== on the fixed build"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "mask_only" 2>&1 | tail -n 3
