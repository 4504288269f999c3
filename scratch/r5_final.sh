#!/bin/bash
# round 5, final build: whole GPU suite, smoke, then the profiles
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r5_final_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash scratch/make_profiles_r05.sh 2>&1 | tail -60
