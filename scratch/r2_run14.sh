#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python bench.py --end-to-end-only > gpurun_out/e2e.log 2> gpurun_out/e2e.err; echo "rc=$?" >> gpurun_out/e2e.err
tail -c 1500 gpurun_out/e2e.log; tail -n 5 gpurun_out/e2e.err
timeout 600 python bench.py --cpu-baseline-only > gpurun_out/cpu.log 2> gpurun_out/cpu.err; echo "rc=$?" >> gpurun_out/cpu.err
tail -c 1200 gpurun_out/cpu.log; tail -n 3 gpurun_out/cpu.err
