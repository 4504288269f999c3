#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python scratch/r6_hm7.py 2>&1 | grep -v amdgpu.ids | tail -n 9 | cut -c1-420
