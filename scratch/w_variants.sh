#!/bin/bash
# config W, bench.py --inner, for the library variants in $VARIANTS (scratch/variants/librufus_<v>.so; "main" = in-tree)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in $VARIANTS; do
  if [ "$v" = main ]; then unset RFX_LIB; else export RFX_LIB=$PWD/scratch/variants/librufus_$v.so; fi
  python bench.py --inner --steps 4 --warmup 3 --no-cpu-baseline --no-end-to-end --no-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$v', '%.1f M reads/s' % (d['value']/1e6), '%.1f ms' % d['ms_per_step'], 'chain %.1f' % r['avg_launch_ms'], {k:round(x,1) for k,x in r['avg_launch_ms_by_kernel'].items() if 'part' in k or 'surv' in k})"
done
