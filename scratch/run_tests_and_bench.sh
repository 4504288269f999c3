#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/final; mkdir -p $O
S=$(date +%s)
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
echo "tests wall $(( $(date +%s) - S )) s"
bash scratch/run_default_bench.sh
timeout 300 python bench.py --workload s1 --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 > $O/r02_bench_s1.json; cut -c1-260 $O/r02_bench_s1.json
