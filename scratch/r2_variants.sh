#!/bin/bash
# runs bench at 1 Gb for each scratch/variants/librufus_*.so (and the in-tree build last)
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/variants; mkdir -p $O
cp rufus_amd/librufus_hip.so /tmp/orig.so
for v in ${VARIANTS:-$(ls scratch/variants/ | sed 's/librufus_//; s/\.so//')} orig; do
  if [ $v = orig ]; then cp /tmp/orig.so rufus_amd/librufus_hip.so; else cp scratch/variants/librufus_$v.so rufus_amd/librufus_hip.so; fi
  timeout 300 python bench.py --genome ${GENOME:-1000000000} --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/$v.log 2> $O/$v.err
  tail -1 $O/$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['avg_launch_ms_by_kernel']; print('$v', round(d['value']/1e6,1), round(d['roofline']['avg_launch_ms'],1), {x:k[x] for x in ('k_msp_part1','k_msp_leaf','k_part2','k_part3')}, d['config']['mutant_kmers'])"
done
