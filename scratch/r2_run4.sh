#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python bench.py --workload s1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/b_s1.json 2> gpurun_out/b_s1.err; echo "rc=$?" >> gpurun_out/b_s1.err
RFX_NO_ARENA=1 timeout 300 python bench.py --workload s1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/b_s1_noarena.json 2> gpurun_out/b_s1_noarena.err
RFX_WGS_TRACE=1 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/b_full.log 2> gpurun_out/b_full.err; echo "rc=$?" >> gpurun_out/b_full.err
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1; echo "all rc=$?" >> gpurun_out/t_all.log
tail -n 3 gpurun_out/t_all.log; tail -c 300 gpurun_out/b_*.err; grep "^\[wgs\]" gpurun_out/b_full.log | tail -18
