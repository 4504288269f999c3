#!/bin/bash
# round 3, batch 26: k_histo_bins with four loads in flight -- histogram parity, then the bench
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/exp26; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_scale_gpu.py -x -q -m gpu -k "histo or trio_in_blocks or wgs or count" > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/full.log 2>$O/bench.err
tail -1 $O/full.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['avg_launch_ms_by_kernel']; print('full', round(d['value']/1e6,1), 'ms', round(d['ms_per_step'],1), 'chain', round(d['roofline']['avg_launch_ms'],1), {x:k.get(x) for x in ('k_surv_part2','k_surv_part3','k_surv_sort')}, d['config']['mutant_kmers'], d['config']['pulled_pairs'], d['config'].get('checked'))"
