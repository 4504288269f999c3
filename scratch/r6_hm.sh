#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python scratch/r6_hm.py 2000000 3000 25 2>&1 | grep -v amdgpu.ids | tail -n 5
timeout 900 python scratch/r6_hm.py 200000 30000 31 2>&1 | grep -v amdgpu.ids | tail -n 5
timeout 900 python scratch/r6_hm.py 20000000 900 27 2>&1 | grep -v amdgpu.ids | tail -n 5
