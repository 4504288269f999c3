#!/bin/bash
# per-kernel ms of the count chain (bench line) for the variants in $VARIANTS ("main" = the in-tree build) on the 1 Gb slice
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in $VARIANTS; do
  if [ "$v" = main ]; then unset RFX_LIB; else export RFX_LIB=$PWD/scratch/variants/librufus_$v.so; fi
  python bench.py --inner --genome ${GENOME:-1000000000} --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-check ${BENCH_FLAGS:-} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$v', 'reads/s %.0fM' % (d['value']/1e6), 'chain %.1f ms' % r['avg_launch_ms'], 'frac %.3f' % r['frac'], {k:round(v,1) for k,v in r['avg_launch_ms_by_kernel'].items()}, 'filter', d.get('roofline_filter',{}).get('frac'))"
done
