#!/bin/bash
# round 3, batch 24: kernel stats of the forced-exchange route next to the local one (1 Gb, 1 pass)
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/exp24; mkdir -p $O
for m in forced local; do
  MODES=$m timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$m -o s -- python scratch/exchange_big.py 1000000000 1 > $O/$m.log 2>&1
  grep forced $O/$m.log
  f=$(find $O/$m -name "*kernel_stats.csv" | head -1); echo "== $m"; head -14 $f | cut -c1-160
done
