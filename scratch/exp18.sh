#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/exp18; mkdir -p $O
pr() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['avg_launch_ms_by_kernel']; print('$2', round(d['value']/1e6,1), 'chain', round(d['roofline']['avg_launch_ms'],1), {x:k.get(x) for x in ('k_msp_part1','k_msp_leaf','k_part2','k_part3')}, d['config']['mutant_kmers'], d['config']['records_per_sample'][0])"; }
cp rufus_amd/librufus_hip.so /tmp/orig.so
for v in ${VARIANTS}; do
cp scratch/variants/librufus_$v.so rufus_amd/librufus_hip.so
timeout 300 python bench.py --genome 1000000000 --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-check > $O/g1_$v.log 2>$O/err; pr $O/g1_$v.log "1Gb $v"
done
cp /tmp/orig.so rufus_amd/librufus_hip.so
