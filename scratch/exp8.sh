#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/exp8; mkdir -p $O
timeout 900 python -m pytest tests/test_cli_gpu.py tests/test_cli_host.py -x -q -m gpu -k "${K:-sam or spool or stranded}" > $O/tests.log 2>&1; tail -5 $O/tests.log
[ "${SCALE:-1}" = 1 ] && bash scratch/feeder_scale.sh ${PAIRS:-16000000} 2>&1 | tee $O/feeder_scale.txt
