#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export RFX_FUZZ_SEEDS=20000-20600
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "three_count_paths or testrun or golden" 2>&1 | tail -n 3
unset RFX_FUZZ_SEEDS
bash scratch/r6_sweep2.sh 2>&1 | tail -n 12
timeout 600 python bench.py --inner --steps 1 --warmup 1 --genome 700000000 --k 27 2>/dev/null | tail -n 1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('k27', d['value']/1e6, r['frac'], r['avg_launch_ms_by_kernel'].get('k_msp_leaf'), d['config']['checks'].get('multiset_checksums'))"
timeout 600 python bench.py --inner --steps 2 --warmup 1 --no-check --genome 1000000000 --passes 2 2>/dev/null | tail -n 1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('k25 1Gb', d['value']/1e6, r['frac'], r['avg_launch_ms'])"
