#!/bin/bash
# Parity at scale away from the headline geometry: the bench's self-check (checksums of S vs S + 1 shard passes over all records,
# (pos,key) order, planted-SNV k-mers found, none in the controls) on other genome sizes, k and pass counts.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_selfcheck_sweep.txt; : > $O
run() {
  echo "--- bench.py $*" >> $O
  timeout 600 python bench.py --inner --steps 1 --warmup 1 "$@" 2>/dev/null | tail -n 1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; ch=c.get('checks',{})
print('   %.0f M reads/s, %s passes, chain frac %.3f; records_verified %s, order violations %s, mutant_in_subject %s, mutant_in_controls %s, snv k-mers %s of %s, passes compared %s, checksums %s' % (d['value']/1e6, c.get('passes'), d['roofline']['frac'], ch.get('records_verified'), ch.get('order_pos_count_violations'), ch.get('mutant_in_subject'), ch.get('mutant_in_controls'), ch.get('snv_kmers_found'), ch.get('snv_kmers_expected'), ch.get('passes_compared'), ch.get('multiset_checksums')))" >> $O 2>&1 || echo "   FAILED" >> $O
}
run --genome 300000000 --k 23
run --genome 300000000 --k 24 --passes 3
run --genome 500000000 --k 25 --passes 1
run --genome 1000000000 --k 25 --passes 5
run --genome 700000000 --k 27
run --genome 700000000 --k 29 --passes 3
run --genome 1000000000 --k 31
run --genome 1500000000 --k 25 --coverage 20
run --workload tn --genome 800000000
run --workload tn --genome 800000000 --k 27 --passes 3
run --genome 2000000000 --k 25
cat $O
