#!/bin/bash
# round 3: parity subset + 1 Gb slice bench after a kernel change
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/exp2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_scale_gpu.py -x -q -m gpu -k "${TESTS:-count or msp or wgs or table or trio}" > $O/tests.log 2>&1; tail -4 $O/tests.log
for env in "A=1" ${EXTRA_ENVS}; do
  echo "== $env"
  env $env timeout 300 python bench.py --genome ${GENOME:-1000000000} --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end 2>$O/bench.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['avg_launch_ms_by_kernel']; print(round(d['value']/1e6,1), round(d['roofline']['avg_launch_ms'],1), k, d['config'].get('mutant_kmers'))"
done
