#!/bin/bash
# the whole GPU suite + smoke, as the driver runs them
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r6_full_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a gpurun_out/r6_full_tests.txt
