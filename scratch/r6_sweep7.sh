#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_selfcheck_sweep7.txt; : > $O
run() {
  echo "--- bench.py $*" >> $O
  timeout 900 python bench.py --inner --steps 1 --warmup 1 "$@" 2>gpurun_out/sweep.err | tail -n 1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; ch=c.get('checks',{}); r=d['roofline']['avg_launch_ms_by_kernel']
print('   %.0f M reads/s, %s passes, chain frac %.3f; chain ms by kernel %s; records_verified %s, order violations %s, mutant_in_subject %s, mutant_in_controls %s, snv k-mers %s of %s, passes compared %s, checksums %s' % (d['value']/1e6, c.get('passes'), d['roofline']['frac'], {k: round(v,1) for k,v in r.items() if v >= 1}, ch.get('records_verified'), ch.get('order_pos_count_violations'), ch.get('mutant_in_subject'), ch.get('mutant_in_controls'), ch.get('snv_kmers_found'), ch.get('snv_kmers_expected'), ch.get('passes_compared'), ch.get('multiset_checksums')))" >> $O 2>&1 || { echo "   FAILED: $(tail -n 3 gpurun_out/sweep.err | cut -c1-400)" >> $O; }
}
run --genome 2000000 --coverage 3000
run --genome 20000000 --coverage 900
run --genome 200000 --coverage 30000 --k 31
run --genome 50000000 --coverage 300 --passes 3
cat $O
