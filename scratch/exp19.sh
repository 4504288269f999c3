#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 python scratch/exchange_big.py 1000000000 1 2>&1 | grep "forced\|exchange path\|Error\|error" | tail -5
for p in ${PASSES:-3 4}; do
  echo "== W, $p passes"; timeout 600 python scratch/exchange_big.py 3100000000 $p 2>&1 | grep "forced\|exchange path\|Error\|error" | tail -5
done
