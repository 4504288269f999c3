#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/exp11; mkdir -p $O
timeout 600 python -m pytest tests/test_cli_gpu.py -x -q -m gpu -k "query or spool" > $O/tests.log 2>&1; tail -3 $O/tests.log
BENCH=${BENCH:-0} EXTRA=${EXTRA:-1} bash scratch/make_profiles_r03.sh
