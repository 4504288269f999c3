#!/bin/bash
# first GPU contact of round 2: new scale tests, the whole GPU suite, benches at growing genome sizes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
nproc > gpurun_out/host.txt; free -g >> gpurun_out/host.txt; rocm-smi --showmeminfo vram >> gpurun_out/host.txt 2>&1
timeout 900 python -m pytest tests/test_scale_gpu.py -x -q -m gpu > gpurun_out/t_scale.log 2>&1; echo "scale rc=$?" >> gpurun_out/t_scale.log
timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_scale_gpu.py > gpurun_out/t_all.log 2>&1; echo "all rc=$?" >> gpurun_out/t_all.log
for g in 300000000 1000000000; do
  timeout 600 python bench.py --genome $g --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/b_$g.json 2> gpurun_out/b_$g.err; echo "rc=$?" >> gpurun_out/b_$g.err
done
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/b_full.json 2> gpurun_out/b_full.err; echo "rc=$?" >> gpurun_out/b_full.err
tail -3 gpurun_out/t_scale.log gpurun_out/t_all.log; tail -c 600 gpurun_out/b_*.err; cat gpurun_out/b_*.json | cut -c1-1500
