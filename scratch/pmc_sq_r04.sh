#!/bin/bash
# SQ counters of the count-chain kernels (two passes: 8 SQ slots each) -> profiles/r04_sq_counters_1g.txt
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/pmc_sq; rm -rf $O; mkdir -p $O
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/a -o s -- python bench.py --inner --genome 1000000000 --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-check > $O/a.log 2>&1
timeout 900 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_VMEM --kernel-trace --output-format csv -d $O/b -o s -- python bench.py --inner --genome 1000000000 --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-check > $O/b.log 2>&1
python - <<'PY'
import csv, collections, re, glob
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob('gpurun_out/pmc_sq/*/**/s_counter_collection.csv', recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"][:30]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "SQ_INSTS_LDS"):
            calls[(k, r["Counter_Name"])] += 1
out = ["# SQ counters, summed over all launches of `python bench.py --steps 1 --warmup 0` (30x trio on a 1 Gb genome, one step:",
       "# 18 read blocks, 3 samples).  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles.", ""]
for k in ("k_msp_part1", "k_part2", "k_msp_leaf", "k_surv_sort", "k_bin_hist", "k_surv_hist", "k_filter_q", "k_flag_absent_tiled"):
    if k in agg:
        out.append(k + "  launches=%d" % max(calls[(k, "SQ_WAVE_CYCLES")], 1))
        for c, v in sorted(agg[k].items()):
            out.append(f"    {c:24s} {v:.4g}")
open("gpurun_out/pmc_sq/r04_sq_counters_1g.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
