#!/bin/bash
# round 3, batch 6: K5 alone at W -- prefetch build, lookups only (no push / drain), key count sweep
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/exp6; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_scale_gpu.py -x -q -m gpu -k "filter or trio_in_blocks" > $O/tests.log 2>&1; tail -2 $O/tests.log
G=${GENOME:-3100000000}
TAG=prefetch timeout 300 python scratch/filter_bench.py $G 24567 2 2>&1 | grep k_filter
TAG=prefetch-8k timeout 300 python scratch/filter_bench.py $G 8172 2 2>&1 | grep k_filter
TAG=one-bit RFX_FQ_ONE=1 timeout 300 python scratch/filter_bench.py $G 24567 2 2>&1 | grep k_filter
cp rufus_amd/librufus_hip.so /tmp/orig.so
cp scratch/variants/librufus_fq_nopush.so rufus_amd/librufus_hip.so
TAG=nopush timeout 300 python scratch/filter_bench.py $G 24567 2 2>&1 | grep k_filter
cp /tmp/orig.so rufus_amd/librufus_hip.so
