#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_scale_gpu.py -x -q -m gpu -k "msp or k31 or tn or wide or 31 or range or three_count or two_rank or exchange" 2>&1 | tail -3
for h in "" 1; do
  if [ -n "$h" ]; then export RFX_MSP_WIDE_HALF=1; else unset RFX_MSP_WIDE_HALF; fi
  timeout 400 python bench.py --inner --workload tn --genome 500000000 --steps 3 --warmup 1 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['avg_launch_ms_by_kernel']; print('half=$h', round(d['value']/1e6,1), 'M reads/s chain', round(d['roofline']['avg_launch_ms'],1), {x:k[x] for x in ('k_msp_part1','k_msp_leaf','k_part2','k_part3')}, d['config']['mutant_kmers'])"
done
