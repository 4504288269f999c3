#!/bin/bash
# round 3, experiment batch 1: where do k_msp_part1 / k_msp_leaf spend their time (1 Gb slice)
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/exp1; mkdir -p $O
cp rufus_amd/librufus_hip.so /tmp/orig.so
cp scratch/variants/librufus_tm.so rufus_amd/librufus_hip.so
timeout 600 python scratch/timing_probe.py 1000000000 > $O/timing.txt 2>&1
cp /tmp/orig.so rufus_amd/librufus_hip.so
VARIANTS="noclose" bash scratch/r2_variants.sh > $O/variants.txt 2>&1
for geo in "RFX_MSP_GEO=1 RFX_MSP_REFINE_BITS=21" "RFX_MSP_GEO=0 RFX_MSP_REFINE_BITS=21" "RFX_MSP_GEO=1 RFX_MSP_REFINE_BITS=20"; do
  echo "== $geo" >> $O/variants.txt
  env $geo timeout 300 python bench.py --genome 1000000000 --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end 2>$O/geo.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['avg_launch_ms_by_kernel']; print(round(d['value']/1e6,1), round(d['roofline']['avg_launch_ms'],1), k)" >> $O/variants.txt
done
cat $O/timing.txt $O/variants.txt
