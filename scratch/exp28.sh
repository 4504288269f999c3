#!/bin/bash
# round 3, batch 28: kernel stats of one step (small kernels of K4 / compaction)
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/exp28; mkdir -p $O; rm -rf $O/stats
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python bench.py --inner --steps 1 --warmup 1 --no-check > $O/stats.log 2>&1
cp "$(find $O/stats -name 's_kernel_stats.csv' | head -1)" $O/kernel_stats.csv; rm -rf $O/stats
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/exp28/kernel_stats.csv')):
    n=r['Name'].replace('(anonymous namespace)::','').replace('void ','')[:34]
    if any(x in n for x in ('compact','flag_','fa_bounds','histo','surv_','scan')): print('%-36s %5s calls %9.2f ms total %8.3f avg'%(n,r['Calls'],int(r['TotalDurationNs'])/1e6,float(r['AverageNs'])/1e6))
PY
