#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
RFX_WGS_TRACE=1 timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/trace_full.log 2> gpurun_out/trace_full.err; echo "rc=$?" >> gpurun_out/trace_full.err
grep "^\[wgs\]" gpurun_out/trace_full.log | tail -20
