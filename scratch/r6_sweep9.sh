#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_selfcheck_sweep9.txt; : > $O
run() {
  echo "--- bench.py $*" >> $O
  timeout 900 python bench.py --inner --steps 1 --warmup 1 "$@" 2>gpurun_out/sweep.err | tail -n 1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; ch=c.get('checks',{})
print('   %.0f M reads/s, %s passes, chain frac %.3f, filter frac %.3f; records_verified %s, order violations %s, mutant_in_subject %s, mutant_in_controls %s, snv k-mers %s of %s, passes compared %s, checksums %s' % (d['value']/1e6, c.get('passes'), d['roofline']['frac'], d['roofline_filter']['frac'], ch.get('records_verified'), ch.get('order_pos_count_violations'), ch.get('mutant_in_subject'), ch.get('mutant_in_controls'), ch.get('snv_kmers_found'), ch.get('snv_kmers_expected'), ch.get('passes_compared'), ch.get('multiset_checksums')))" >> $O 2>&1 || { echo "   FAILED: $(tail -n 3 gpurun_out/sweep.err | cut -c1-400)" >> $O; }
}
run --genome 50000000 --coverage 300 --passes 3
run --genome 100000000 --coverage 150
run --workload tn --genome 100000000 --coverage 120
cat $O
