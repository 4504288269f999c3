#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/exp9; mkdir -p $O
timeout 900 python -m pytest tests/test_cli_gpu.py tests/test_scale_gpu.py -x -q -m gpu -k "several_devices or goldens or lock_step or pieces_and_ragged" > $O/tests.log 2>&1; tail -5 $O/tests.log
