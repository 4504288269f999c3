#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export RFX_FUZZ_SEEDS=40000-41500
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "three_count_paths or testrun or golden" 2>&1 | tail -n 2
unset RFX_FUZZ_SEEDS
bash scratch/r6_sweep5.sh 2>&1 | tail -n 8
O=gpurun_out/r06_selfcheck_sweep6.txt; : > $O
run() {
  echo "--- bench.py $*" >> $O
  timeout 900 python bench.py --inner --steps 1 --warmup 1 "$@" 2>gpurun_out/sweep.err | tail -n 1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; ch=c.get('checks',{}); r=d['roofline']['avg_launch_ms_by_kernel']
print('   %.0f M reads/s, %s passes, chain frac %.3f (leaf %.0f of %.0f ms); records_verified %s, order violations %s, mutant_in_controls %s, snv k-mers %s of %s, passes compared %s, checksums %s' % (d['value']/1e6, c.get('passes'), d['roofline']['frac'], r.get('k_msp_leaf',0), d['roofline']['avg_launch_ms'], ch.get('records_verified'), ch.get('order_pos_count_violations'), ch.get('mutant_in_controls'), ch.get('snv_kmers_found'), ch.get('snv_kmers_expected'), ch.get('passes_compared'), ch.get('multiset_checksums')))" >> $O 2>&1 || { echo "   FAILED: $(tail -n 3 gpurun_out/sweep.err | cut -c1-300)" >> $O; }
}
run --genome 300000000 --k 23
run --genome 700000000 --k 29 --passes 3
run --genome 700000000 --k 30
run --workload tn --genome 800000000 --k 29
run --genome 3100000000 --k 30
cat $O
