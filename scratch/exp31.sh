#!/bin/bash
# round 3, last call: bench line of the final build (legs on) and the full-size tumor/normal line
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/exp31; mkdir -p $O
timeout 330 python3 bench.py --gpus 1 --steps 10 --warmup 2 > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log > $O/r03_bench.json
python3 -c "
import json; d=json.load(open('$O/r03_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline_filter']['frac'], d['config']['checked'], d['roofline']['avg_launch_ms_by_kernel']); print(json.dumps(d.get('cpu_baseline'))[:300]); print(json.dumps(d.get('end_to_end'))[:300])"
timeout 200 python bench.py --workload tn --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/tn.log 2>$O/tn.err; tail -1 $O/tn.log > $O/r03_bench_tn_full.json
python3 -c "
import json; d=json.load(open('$O/r03_bench_tn_full.json')); print('TN', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['config']['checked'], d['roofline']['avg_launch_ms_by_kernel'])"; tail -2 $O/tn.err
