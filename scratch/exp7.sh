#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
G=${GENOME:-3100000000}
cp rufus_amd/librufus_hip.so /tmp/orig.so
for v in ${VARIANTS:-fq_noload fq_nofence}; do
cp scratch/variants/librufus_$v.so rufus_amd/librufus_hip.so
TAG=$v FQ_TIMING=1 timeout 300 python scratch/filter_bench.py $G 24567 2 2>&1 | grep "k_filter\|drain"
done
cp /tmp/orig.so rufus_amd/librufus_hip.so
