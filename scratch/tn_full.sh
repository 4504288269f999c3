#!/bin/bash
# BASELINE configs[4] at full size on ONE GPU: tumor 60x / normal 30x, 3.1 Gb genome, k = 31
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/tn_full; mkdir -p $O
S=$(date +%s)
RFX_WGS_TRACE=1 timeout 1500 python bench.py --inner --workload tn --steps 2 --warmup 1 > $O/tn.log 2> $O/tn.err
echo "rc=$? wall $(( $(date +%s) - S )) s"
tail -1 $O/tn.log | cut -c1-1800
grep -c "out of device memory" $O/tn.log; grep "out of device memory" $O/tn.log | tail -2; tail -3 $O/tn.err
