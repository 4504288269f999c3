#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python scratch/r6_hm6.py 2>&1 | grep -v amdgpu.ids | tail -n 12
