#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_scale_gpu.py tests/test_gpu_parity.py -x -q -m gpu --deselect tests/test_scale_gpu.py::test_wgs_slice_properties > gpurun_out/t_all.log 2>&1; echo "rc=$?" >> gpurun_out/t_all.log
tail -n 4 gpurun_out/t_all.log
timeout 1500 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err; echo "rc=$?" >> gpurun_out/bench_default.err
tail -n 3 gpurun_out/bench_default.err; tail -1 gpurun_out/bench_default.log | cut -c1-400
