#!/bin/bash
# round 3, batch 22: agreed retry on a group (two ranks sharing the GPU), exchange checkpoints
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/exp22; mkdir -p $O
timeout 1500 python -m pytest tests/test_scale_gpu.py -x -q -m gpu -k "${TESTS:-two_ranks or several_devices}" > $O/tests.log 2>&1; tail -15 $O/tests.log
