#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/tests; mkdir -p $O
S=$(date +%s)
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
echo "tests wall $(( $(date +%s) - S )) s"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
