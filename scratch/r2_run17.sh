#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r17; mkdir -p $O
S=$(date +%s)
timeout 1700 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.log 2> $O/bench.err
echo "wall $(( $(date +%s) - S )) s" >> $O/bench.err
tail -1 $O/bench.log > $O/r02_bench.json
cut -c1-1500 $O/r02_bench.json; tail -5 $O/bench.err
