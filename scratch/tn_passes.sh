#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
for p in 6 5; do
  timeout 600 python bench.py --inner --workload tn --passes $p --steps 1 --warmup 1 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('passes asked $p ->', d['config']['passes'], round(d['value']/1e6,1), 'M reads/s', round(d['ms_per_step']), 'ms/step peak', round(d['config']['hbm_peak_bytes']/1e9,1), 'GB mapped', round(d['config']['hbm_mapped_bytes']/1e9,1))"
done
