#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export RFX_BENCH_STEP_TIMES=1
S='s/\(ms: \[\).*, \([0-9.]*\]\)/\1... \2/'
for e in 1 2 3; do
  timeout 300 python bench.py --workload s1 --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end 2>&1 | grep -v amdgpu.ids | cut -c1-230 | sed -e "$S"
done
