#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 400 bash scratch/cli_scale.sh 32000000 320000000 64 > gpurun_out/cli_scale.log 2>&1; echo "rc=$?" >> gpurun_out/cli_scale.log; grep -v "^-rw" gpurun_out/cli_scale.log
rm -rf /dev/shm/rfx_cli_scale
for cfg in "base::" "geo1:RFX_MSP_GEO=1:" "bits23geo1:RFX_MSP_GEO=1:RFX_MSP_REFINE_BITS=23" "bits23:RFX_MSP_REFINE_BITS=23:"; do
  name=${cfg%%:*}; rest=${cfg#*:}; e1=${rest%%:*}; e2=${rest#*:}
  env $e1 $e2 timeout 300 python bench.py --genome 1000000000 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/b1g_$name.json 2> gpurun_out/b1g_$name.err
done
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/b_full.json 2> gpurun_out/b_full.err; echo "rc=$?" >> gpurun_out/b_full.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/b1g_*.json"))+["gpurun_out/b_full.json"]:
    try:
        b=json.loads(open(f).read()); r=b["roofline"]
        print(f, "%.1f M reads/s"%(b["value"]/1e6), "chain %.0f ms"%r["avg_launch_ms"], "leaf %.0f part1 %.0f part3 %.0f"%(r["avg_launch_ms_by_kernel"].get("k_msp_leaf",0), r["avg_launch_ms_by_kernel"].get("k_msp_part1",0), r["avg_launch_ms_by_kernel"].get("k_part3",0)), "passes", b["config"]["passes"], "peak %.0f GB"%(b["config"]["hbm_peak_bytes"]/1e9))
    except Exception as e:
        print(f, "failed", e)
PY
