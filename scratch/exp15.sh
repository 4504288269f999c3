#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
cp rufus_amd/librufus_hip.so /tmp/orig.so
cp scratch/variants/librufus_tm.so rufus_amd/librufus_hip.so
timeout 600 python scratch/timing_probe.py 1000000000 2>&1 | tail -22
cp /tmp/orig.so rufus_amd/librufus_hip.so
