#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/exp17; mkdir -p $O
timeout 900 python -m pytest tests/test_cli_gpu.py -x -q -m gpu -k "${K:-round3 or spool or query or sam}" > $O/tests.log 2>&1; tail -15 $O/tests.log
