#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_scale_gpu.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/t_all.log 2>&1; echo "all rc=$?" >> gpurun_out/t_all.log
RFX_WGS_TRACE=1 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/b_full.log 2> gpurun_out/b_full.err; echo "rc=$?" >> gpurun_out/b_full.err
tail -n 3 gpurun_out/t_all.log; tail -c 300 gpurun_out/b_*.err; grep "^\[wgs\]" gpurun_out/b_full.log | tail -8
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_1g -o p -- python $GRAFT_REPO_ROOT/bench.py --genome 1000000000 --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_1g.log 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/prof_1g | head
