#!/bin/bash
# round 3, batch 30: k_part3 as ONE launch per refinement chunk over the slices of all read blocks (k_part2<.., MULTI>,
# rfxk::part2_multi) against a launch per (chunk, block): A/B at W, the parity tests of the refinement on the new build,
# and its kernel stats when it wins
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/exp30; mkdir -p $O
cp rufus_amd/librufus_hip.so /tmp/keep.so
for v in base new; do
  cp scratch/variants/librufus_$v.so rufus_amd/librufus_hip.so
  timeout 240 python bench.py --inner --steps 2 --warmup 1 > $O/$v.log 2> $O/$v.err
  python - $O/$v.log $v <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]; b = r["avg_launch_ms_by_kernel"]
    print(sys.argv[2], "reads/s %.1f M" % (d["value"] / 1e6), "step %.0f ms" % d["ms_per_step"], "chain %.1f" % r["avg_launch_ms"],
          {k: b[k] for k in ("k_part2", "k_bin_hist", "k_part3", "k_msp_leaf")}, "launches", r["launches_by_kernel_per_chain"]["k_part3"],
          "checked", d["config"]["checked"], "viol", (d["config"]["checks"] or {}).get("order_pos_count_violations"),
          "mut", d["config"]["mutant_kmers"], d["config"]["pulled_pairs"], d["config"]["records_per_sample"])
except Exception as e:
    print(sys.argv[2], "FAILED", repr(e)); print(open(sys.argv[1].replace(".log", ".err")).read()[-800:])
PY
done
cp /tmp/keep.so rufus_amd/librufus_hip.so
S=$(date +%s)
timeout 200 python -m pytest tests/test_gpu_parity.py tests/test_scale_gpu.py -x -q -m gpu -k "two_level or msp_count or msp_bins or refines or three_count or shard_passes or p2l_dense or trio_in_blocks or table_counts or tumor_normal or synthetic_count or several_devices" > $O/tests.log 2>&1; tail -3 $O/tests.log
echo "tests wall $(( $(date +%s) - S )) s"
if python - <<'PY'
import json, sys
c = {v: json.loads(open(f"gpurun_out/exp30/{v}.log").read().strip().splitlines()[-1])["roofline"]["avg_launch_ms"] for v in ("base", "new")}
ok = "passed" in open("gpurun_out/exp30/tests.log").read().splitlines()[-1] and "failed" not in open("gpurun_out/exp30/tests.log").read().splitlines()[-1]
sys.exit(0 if ok and c["new"] < c["base"] - 4 else 1)
PY
then
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python bench.py --inner --steps 1 --warmup 1 --no-check > $O/stats.log 2>&1
  cp "$(find $O/stats -name 's_kernel_stats.csv' | head -1)" $O/r03_kernel_stats_wgs.csv; rm -rf $O/stats; echo "stats taken"
fi
