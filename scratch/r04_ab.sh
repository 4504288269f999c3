#!/bin/bash
# round-4 A/B helper: 1 Gb slice and (FULL=1) the W bench, gpu-side only, per-kernel ms from the bench line -> gpurun_out/$TAG/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$R"
O=gpurun_out/${TAG:-r04a}; mkdir -p $O
if [ "${TESTS:-0}" = 1 ]; then
  S=$(date +%s)
  timeout ${TEST_TIMEOUT:-1500} python -m pytest tests/ -x -q -m gpu ${TEST_ARGS:-} > $O/tests.log 2>&1; tail -5 $O/tests.log
  echo "tests wall $(( $(date +%s) - S )) s"
fi
if [ "${SLICE:-1}" = 1 ]; then
  timeout 600 python bench.py --inner --genome 1000000000 --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end ${BENCH_FLAGS:-} > $O/bench_1g.log 2> $O/bench_1g.err
  tail -1 $O/bench_1g.log > $O/bench_1g.json; cut -c1-2500 $O/bench_1g.json; tail -3 $O/bench_1g.err
fi
if [ "${FULL:-0}" = 1 ]; then
  timeout 900 python bench.py --inner --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --no-end-to-end ${BENCH_FLAGS:-} > $O/bench_w.log 2> $O/bench_w.err
  tail -1 $O/bench_w.log > $O/bench_w.json; cut -c1-2500 $O/bench_w.json; tail -3 $O/bench_w.err
fi
