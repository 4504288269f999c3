#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for v in default mmap host; do
  unset RFX_TEXT_MMAP RFX_HOST_PARSE
  [ $v = mmap ] && export RFX_TEXT_MMAP=1
  [ $v = host ] && export RFX_HOST_PARSE=1
  timeout 900 python bench.py --end-to-end-only 2>gpurun_out/r6t_e2e_$v.err | tail -1 > gpurun_out/r6t_e2e_$v.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r6t_e2e_$v.json"))
print("$v", d["stages_s"], "value %.2f M" % (d["value"]/1e6), "pj %.2f M" % (d["parallel_jelly"]["value"]/1e6), d["parallel_jelly"]["jellyfish count x 3_s"])
PY
done
