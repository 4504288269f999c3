#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for v in async sync async sync; do
  unset RFX_SYNC_OPEN; [ $v = sync ] && export RFX_SYNC_OPEN=1
  timeout 900 python bench.py --end-to-end-only 2>gpurun_out/r6x_e2e_$v.err | tail -1 > gpurun_out/r6x_e2e_$v.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r6x_e2e_$v.json"))
print("$v", d["stages_s"], "value %.2f M" % (d["value"]/1e6), "pj %.2f M" % (d["parallel_jelly"]["value"]/1e6), d["parallel_jelly"]["jellyfish count x 3_s"])
PY
done
