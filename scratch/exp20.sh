#!/bin/bash
# round 3, batch 20: tiled k_flag_absent -- K4 parity both routes, then the bench (RFX_K4_TILE_MIN=huge: the untiled search)
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/exp20; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_scale_gpu.py -x -q -m gpu -k "merge or hash or subtract or trio or wgs or full_size" > $O/tests.log 2>&1; tail -4 $O/tests.log
pr() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', round(d['value']/1e6,1), 'ms', round(d['ms_per_step'],1), d['config']['mutant_kmers'], d['config']['pulled_pairs'], d['config'].get('checked'))"; }
for env in ${ENVS:-A=1}; do
  env $env timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/full_$env.log 2>$O/bench.err; pr $O/full_$env.log "full $env"
done
