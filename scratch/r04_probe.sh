#!/bin/bash
# phase breakdown (timing_probe.py) of the variants named in $VARIANTS on a 1 Gb slice
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in $VARIANTS; do echo "== $v"; RFX_LIB=$PWD/scratch/variants/librufus_$v.so python scratch/timing_probe.py ${GENOME:-1000000000} 2>&1 | grep -v amdgpu.ids | tail -16; done
