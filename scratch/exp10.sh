#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/exp10; mkdir -p $O
timeout 600 python -m pytest tests/test_cli_gpu.py -x -q -m gpu -k "query or goldens" > $O/tests.log 2>&1; tail -3 $O/tests.log
bash scratch/cli_w_trio.sh ${PAIRS:-310000000} ${GENOME:-3100000000}
