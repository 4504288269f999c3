#!/bin/bash
# round 3, batch 3: queue filter parity + A/B at 1 Gb; k_msp_part1 without the record path; phase timing
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/exp3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_scale_gpu.py -x -q -m gpu -k "filter or trio_in_blocks or smoke" > $O/tests.log 2>&1; tail -4 $O/tests.log
pr() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['avg_launch_ms_by_kernel']; print('$2', round(d['value']/1e6,1), 'chain', round(d['roofline']['avg_launch_ms'],1), 'filter', round(d['roofline_filter']['ms_per_step'],2), d['roofline_filter']['frac'], {x:k[x] for x in ('k_msp_part1','k_msp_leaf','k_part2','k_part3')}, d['config']['mutant_kmers'], d['config']['pulled_pairs'])"; }
for env in "A=1" "RFX_FILTER_OLD=1"; do
  env $env timeout 300 python bench.py --genome 1000000000 --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/f_$env.log 2>$O/bench.err; pr $O/f_$env.log "$env"
done
cp rufus_amd/librufus_hip.so /tmp/orig.so
cp scratch/variants/librufus_noclose.so rufus_amd/librufus_hip.so
timeout 300 python bench.py --genome 1000000000 --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/noclose.log 2>$O/bench.err; pr $O/noclose.log noclose
cp scratch/variants/librufus_tm.so rufus_amd/librufus_hip.so
timeout 600 python scratch/timing_probe.py 1000000000 > $O/timing.txt 2>&1; cat $O/timing.txt
cp /tmp/orig.so rufus_amd/librufus_hip.so
