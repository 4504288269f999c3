#!/bin/bash
# round 3, batch 27: phase breakdown of k_msp_leaf (wave-0 cycles), 1 Gb slice, -DRFX_TIMING variant
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/exp27; mkdir -p $O
for v in ${VARIANTS:-timing}; do
  cp scratch/variants/librufus_$v.so rufus_amd/librufus_hip.so
  echo "== $v"; timeout 600 python scratch/timing_probe.py 1000000000 2>&1 | grep -v "^$" | tail -22
done > $O/probe.txt 2>&1
cat $O/probe.txt
