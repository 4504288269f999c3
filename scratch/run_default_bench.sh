#!/bin/bash
# the driver's command; output -> gpurun_out/bench_default/
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/bench_default; mkdir -p $O
S=$(date +%s)
timeout 1700 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.log 2> $O/bench.err
echo "bench wall $(( $(date +%s) - S )) s rc=$?"
tail -1 $O/bench.log > $O/r02_bench.json
python3 - <<PY
import json
d=json.load(open("$O/r02_bench.json"))
print(d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
print(json.dumps(d.get("end_to_end"))[:900])
print(json.dumps(d.get("cpu_baseline"))[:300])
PY
tail -3 $O/bench.err
