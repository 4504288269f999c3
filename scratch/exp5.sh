#!/bin/bash
# round 3, batch 5: queue filter v2 (cross-chunk queue, 4 probes in flight) + self-check
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/exp5; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_scale_gpu.py -x -q -m gpu -k "filter or trio_in_blocks or smoke" > $O/tests.log 2>&1; tail -4 $O/tests.log
pr() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['avg_launch_ms_by_kernel']; print('$2', round(d['value']/1e6,1), 'chain', round(d['roofline']['avg_launch_ms'],1), 'filter', round(d['roofline_filter']['ms_per_step'],2), round(d['roofline_filter']['frac'],3), {x:k.get(x) for x in ('k_msp_part1','k_msp_leaf','k_part2','k_bin_hist','k_part3')}, d['config']['mutant_kmers'], d['config']['pulled_pairs'], d['config'].get('checks'))"; }
timeout 300 python bench.py --genome 1000000000 --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/g1.log 2>$O/bench.err; pr $O/g1.log "1Gb"; tail -3 $O/bench.err
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/full.log 2>$O/bench_full.err; pr $O/full.log "full"; tail -3 $O/bench_full.err
timeout 300 python bench.py --workload s1 --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 | cut -c1-200
