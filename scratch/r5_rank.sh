#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python scratch/rank_of_n.py ${GENOME:-3100000000} 2 4 8 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_rank_of_n.txt
