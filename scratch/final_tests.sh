#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/final; mkdir -p $O
S=$(date +%s)
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/tests.log 2>&1; tail -4 $O/tests.log
echo "tests wall $(( $(date +%s) - S )) s"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > $O/bench_default.log 2> $O/bench_default.err; tail -1 $O/bench_default.log | cut -c1-400
