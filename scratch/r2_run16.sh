#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
RFX_WGS_TRACE=1 timeout 900 python bench.py --passes 3 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/b_p3.log 2> gpurun_out/b_p3.err; echo "rc=$?" >> gpurun_out/b_p3.err
tail -n 3 gpurun_out/b_p3.err; grep "retry" gpurun_out/b_p3.log | head
python - <<'PY'
import json
l=[x for x in open("gpurun_out/b_p3.log") if x.startswith("{")]
if l:
    b=json.loads(l[-1]); r=b["roofline"]
    print("%.1f M reads/s"%(b["value"]/1e6), "chain %.0f"%r["avg_launch_ms"], "frac %.3f"%r["frac"], "passes", b["config"]["passes"], "peak %.0f mapped %.0f"%(b["config"]["hbm_peak_bytes"]/1e9, b["config"]["hbm_mapped_bytes"]/1e9), r["avg_launch_ms_by_kernel"])
PY
