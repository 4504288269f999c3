#!/bin/bash
# final checks of a build: the whole GPU suite, the driver's bench command, the tumor/normal line
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/exp12; mkdir -p $O
S=$(date +%s)
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/tests.log 2>&1; tail -4 $O/tests.log
echo "tests wall $(( $(date +%s) - S )) s"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
STATS=0 PMC=0 SQ=0 BENCH=1 EXTRA=0 bash scratch/make_profiles_r03.sh
timeout 900 python bench.py --workload tn --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/tn.log 2>$O/tn.err; tail -1 $O/tn.log > gpurun_out/profiles_r03/r03_bench_tn_full.json; cut -c1-200 gpurun_out/profiles_r03/r03_bench_tn_full.json; tail -2 $O/tn.err
