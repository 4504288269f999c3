#!/bin/bash
# round-3 baseline: gpu tests, default bench (3 steps), kernel stats of one step -> gpurun_out/r03a/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$R"
O=gpurun_out/${TAG:-r03a}; mkdir -p $O
S=$(date +%s)
if [ "${TESTS:-1}" = 1 ]; then
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
echo "tests wall $(( $(date +%s) - S )) s"
fi
S=$(date +%s)
timeout 900 python bench.py --steps ${STEPS:-3} --warmup 1 ${BENCH_FLAGS:-} > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log > $O/bench.json
echo "bench wall $(( $(date +%s) - S )) s"
if [ "${STATS:-1}" = 1 ]; then
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python bench.py --inner --steps 1 --warmup 1 --no-cpu-baseline > $O/stats.log 2>&1
cp "$(find $O/stats -name 's_kernel_stats.csv' | head -1)" $O/kernel_stats.csv
rm -rf $O/stats
head -20 $O/kernel_stats.csv | cut -c1-170
fi
cut -c1-1500 $O/bench.json; tail -5 $O/bench.err
