#!/bin/bash
# N ranks on the ONE device at a size whose exchanges exceed 1 GiB per message: the same results as one rank?
cd "$GRAFT_REPO_ROOT" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
F="--genome ${G:-600000000} --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-check"
summ() { python3 -c "
import json,sys
l=[x for x in sys.stdin.read().splitlines() if x.startswith('{')]
if not l: print('   NO LINE'); sys.exit()
d=json.loads(l[-1]); c=d['config']
print('   n_gpus %s value %.0f M reads/s; mutant_kmers %s pulled %s records %s; passes %s; exchange %s' % (d['n_gpus'], d['value']/1e6, c.get('mutant_kmers'), c.get('pulled_pairs'), c.get('records_per_sample'), c.get('passes'), str(c.get('exchange_bytes_per_rank'))[:120]))"; }
echo "--- 1 rank"; timeout 900 python bench.py --inner $F 2>/dev/null | summ
for n in 2 4; do
  for be in nccl gloo; do
    echo "--- $n ranks, one device, $be"
    RFX_BENCH_BACKEND=$be timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500+n)) bench.py --gpus $n --one-device $F > gpurun_out/ranks_$n.out 2> gpurun_out/ranks_$n.err
    rc=$?; cat gpurun_out/ranks_$n.out | summ; [ $rc = 0 ] && break; echo "   rc $rc: $(grep -v "^\[" gpurun_out/ranks_$n.err | tail -n 2 | cut -c1-250)"
  done
done
