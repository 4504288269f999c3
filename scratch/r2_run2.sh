#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_scale_gpu.py::test_wgs_slice_properties > gpurun_out/t_all.log 2>&1; echo "all rc=$?" >> gpurun_out/t_all.log
timeout 900 python -m pytest tests/test_scale_gpu.py -x -q -m gpu -k wgs_slice > gpurun_out/t_scale.log 2>&1; echo "scale rc=$?" >> gpurun_out/t_scale.log
timeout 300 python bench.py --workload s1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/b_s1.json 2> gpurun_out/b_s1.err; echo "rc=$?" >> gpurun_out/b_s1.err
for g in 300000000 1000000000; do
  timeout 600 python bench.py --genome $g --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/b_$g.json 2> gpurun_out/b_$g.err; echo "rc=$?" >> gpurun_out/b_$g.err
done
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/b_full.json 2> gpurun_out/b_full.err; echo "rc=$?" >> gpurun_out/b_full.err
tail -n 3 gpurun_out/t_scale.log gpurun_out/t_all.log; tail -c 300 gpurun_out/b_*.err
