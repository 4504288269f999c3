#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_cli_gpu.py -x -q -m gpu 2>&1 | tail -3
for e in 0 1; do
  if [ $e = 1 ]; then export RFX_NO_THP_PIN=1; else unset RFX_NO_THP_PIN; fi
  timeout 900 python bench.py --end-to-end-only 2>gpurun_out/r6s_e2e_$e.err | tail -1 > gpurun_out/r6s_e2e_$e.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r6s_e2e_$e.json"))
print("no_thp=$e", d["stages_s"], "value %.2f M" % (d["value"]/1e6), "pj %.2f M" % (d["parallel_jelly"]["value"]/1e6), d["parallel_jelly"]["jellyfish count x 3_s"])
PY
done
