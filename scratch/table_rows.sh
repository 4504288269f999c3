#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
for args in "--genome 1000000000" "--genome 300000000" "--workload tn --genome 500000000"; do
  timeout 400 python bench.py --inner $args --steps 3 --warmup 1 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$args', round(d['value']/1e6,1), 'M reads/s', round(d['ms_per_step'],1), 'ms/step chain', round(d['roofline']['avg_launch_ms'],1), 'frac', round(d['roofline']['frac'],3), 'passes', d['config'].get('passes'))"
done
