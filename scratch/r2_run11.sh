#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_scale_gpu.py -x -q -m gpu -k "two_ranks or k31" > gpurun_out/t_k31.log 2>&1; echo "rc=$?" >> gpurun_out/t_k31.log
tail -n 8 gpurun_out/t_k31.log
timeout 300 python tests/rccl_selftest.py > gpurun_out/rccl.log 2>&1; tail -n 3 gpurun_out/rccl.log
timeout 600 python bench.py --workload tn --genome 500000000 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/b_tn_500m.json 2> gpurun_out/b_tn_500m.err; echo "rc=$?" >> gpurun_out/b_tn_500m.err
tail -c 300 gpurun_out/b_tn_500m.err
python - <<'PY'
import json
b=json.loads(open("gpurun_out/b_tn_500m.json").read()); r=b["roofline"]
print("tn 500M: %.1f M reads/s"%(b["value"]/1e6), "ms/step %.0f"%b["ms_per_step"], "frac %.3f"%r["frac"], r["avg_launch_ms_by_kernel"], b["config"]["passes"], b["config"]["mutant_kmers"], b["config"]["pulled_pairs"])
PY
