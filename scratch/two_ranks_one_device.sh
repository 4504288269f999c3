#!/bin/bash
# The N-rank path of bench.py with a REAL second rank on the box's one GPU (bench.py --gpus 2 --one-device), at a size where the
# numbers mean something: 30x trio of a 1 Gb genome.  One rank first (the reference), then two ranks over RCCL, then over gloo.
# usage: two_ranks_one_device.sh [genome=1000000000]
cd "$GRAFT_REPO_ROOT" || exit 1
G=${1:-1000000000}
F="--genome $G --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end"
summ() { python3 -c "
import json,sys
d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); c=d['config']
print('$2: %.0f M reads/s, %.0f ms per step, n_gpus %d, mutant k-mers %s, pulled pairs %s, records %s, checks %s, dry run: %s' % (d['value']/1e6, d['ms_per_step'], d['n_gpus'], c.get('mutant_kmers'), c.get('pulled_pairs'), c.get('records_per_sample'), c.get('checks'), c.get('one_device_dry_run')))"; }
python bench.py --inner $F > /tmp/one.log 2>/tmp/one.err || tail -5 /tmp/one.err
summ /tmp/one.log "one rank"
for backend in nccl gloo; do
  RFX_BENCH_BACKEND=$backend HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --one-device $F > /tmp/two_$backend.log 2>/tmp/two_$backend.err
  if grep -q '^{' /tmp/two_$backend.log; then summ /tmp/two_$backend.log "two ranks on the one device over $backend"; else echo "two ranks over $backend: failed"; tail -3 /tmp/two_$backend.err; fi
done
