#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_cli_gpu.py tests/test_text_gpu.py -x -q -m gpu 2>&1 | tail -5
for e in 0 1; do
  if [ $e = 1 ]; then export RFX_HOST_PARSE=1; else unset RFX_HOST_PARSE; fi
  timeout 900 python bench.py --end-to-end-only 2>gpurun_out/r6k_e2e_$e.err | tail -1 > gpurun_out/r6k_e2e_$e.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r6k_e2e_$e.json"))
print("host_parse=$e", json.dumps(d)[:1400])
PY
done
