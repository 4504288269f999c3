#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/exp25; mkdir -p $O
MODES=forced RFX_WGS_TRACE=host timeout 600 python scratch/exchange_big.py 1000000000 1 2>&1 | grep -E "wgs|forced" > $O/forced.txt
tail -14 $O/forced.txt
