#!/bin/bash
# round 3, batch 23: forced exchange at scale (one rank, RCCL self group): serial rounds (default on one rank) and with the
# next block partitioned during the exchange (RFX_WGS_OVERLAP=1, the default of a real group)
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/exp23; mkdir -p $O
timeout 1500 python -m pytest tests/test_scale_gpu.py -x -q -m gpu -k "two_ranks or several_devices" > $O/tests.log 2>&1; tail -3 $O/tests.log
{
echo "== 1 Gb, 1 pass (each kind twice: the first run also grows the arena and torch's cache)"
timeout 600 python scratch/exchange_big.py 1000000000 1 2>&1 | grep -E "forced|path ==|Error|error"
echo "== 1 Gb, 1 pass, RFX_WGS_OVERLAP=1"
RFX_WGS_OVERLAP=1 timeout 600 python scratch/exchange_big.py 1000000000 1 2>&1 | grep -E "forced|path ==|Error|error"
echo "== W, 3 passes"
timeout 900 python scratch/exchange_big.py 3100000000 3 2>&1 | grep -E "forced|path ==|Error|error"
} > $O/exchange_big.txt
cat $O/exchange_big.txt
