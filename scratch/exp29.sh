#!/bin/bash
# round 3, batch 29: k_bin_hist with four loads in flight + k_part2 fetching the next tile during the write-out,
# A/B against the build before (scratch/variants/librufus_{base,new}.so) at W, then the parity tests that go
# through these kernels on the new build
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/exp29; mkdir -p $O
cp rufus_amd/librufus_hip.so /tmp/keep.so
for i in 1 2; do for v in base new; do
  cp scratch/variants/librufus_$v.so rufus_amd/librufus_hip.so
  timeout 240 python bench.py --inner --steps 2 --warmup 1 > $O/$v.$i.log 2> $O/$v.$i.err
  python - $O/$v.$i.log $v <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]; b = r["avg_launch_ms_by_kernel"]
    print(sys.argv[2], "reads/s %.1f M" % (d["value"] / 1e6), "step %.0f ms" % d["ms_per_step"], "chain %.1f" % r["avg_launch_ms"],
          {k: b[k] for k in ("k_part2", "k_bin_hist", "k_part3", "k_surv_part2", "k_surv_part3")}, "checked", d["config"]["checked"],
          "viol", (d["config"]["checks"] or {}).get("order_pos_count_violations"), "mut", d["config"]["mutant_kmers"], d["config"]["pulled_pairs"])
except Exception as e:
    print(sys.argv[2], "FAILED", repr(e)); print(open(sys.argv[1].replace(".log", ".err")).read()[-600:])
PY
done; done
cp /tmp/keep.so rufus_amd/librufus_hip.so
S=$(date +%s)
timeout 330 python -m pytest tests/test_gpu_parity.py tests/test_scale_gpu.py -x -q -m gpu -k "two_level or msp_count or msp_bins or refines or three_count or shard_passes or p2l_dense or trio_in_blocks or table_counts or tumor_normal or synthetic_count" > $O/tests.log 2>&1; tail -3 $O/tests.log
echo "tests wall $(( $(date +%s) - S )) s"
