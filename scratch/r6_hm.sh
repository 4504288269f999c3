#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python scratch/r6_hm5.py 50000000 600 25 2>&1 | grep -v amdgpu.ids | tail -n 10
timeout 900 python scratch/r6_hm5.py 30000000 300 31 2>&1 | grep -v amdgpu.ids | tail -n 10
