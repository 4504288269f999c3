#!/bin/bash
# SQ counters per kernel of the count chain on the 1 Gb slice (two passes, run maps on) -> gpurun_out/r6_sq.txt
# usage: bash scratch/r5_sq.sh [extra bench args]
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/pmc_sq; rm -rf $O; mkdir -p $O
export RFX_BENCH_MAP_BUDGET=${RFX_BENCH_MAP_BUDGET:-14e9}
ARGS="--inner --genome 1000000000 --passes 2 --steps 1 --warmup 0 --no-cpu-baseline --no-check $*"
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/a -o s -- python bench.py $ARGS > $O/a.log 2>&1
timeout 900 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_VMEM --kernel-trace --output-format csv -d $O/b -o s -- python bench.py $ARGS > $O/b.log 2>&1
python - <<'PY'
import csv, collections, re, glob
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob('gpurun_out/pmc_sq/*/**/s_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"][:30]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "SQ_INSTS_LDS"):
            calls[(k, r["Counter_Name"])] += 1
rows = 3 * 2e8 * 150 / 64
out = ["# SQ counters summed over all launches of one step of the 30x trio on a 1 Gb genome, 2 shard passes (%.3g wave-rows of 64 bases)" % rows,
       "# kernel launches  VALU+SALU per row  LDS per row  VMEM per row  wait_any/wave_cycles  lds_conflict/lds_active"]
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0)):
    a = agg[k]
    if a.get("SQ_WAVE_CYCLES", 0) < 1e8: continue
    out.append("%-16s %5d  %7.2f  %6.2f  %6.3f  %.3f  %.3f   busy %.4g" % (k, max(calls[(k, "SQ_WAVE_CYCLES")], 1),
        (a["SQ_INSTS_VALU"] + a["SQ_INSTS_SALU"]) / rows, a.get("SQ_INSTS_LDS", 0) / rows, a.get("SQ_INSTS_VMEM", 0) / rows,
        a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"], a.get("SQ_LDS_BANK_CONFLICT", 0) / max(a.get("SQ_LDS_IDX_ACTIVE", 1), 1), a["SQ_BUSY_CYCLES"]))
    out.append("      " + "  ".join(f"{c[3:]}={v:.3g}" for c, v in sorted(a.items())))
open("gpurun_out/r6_sq.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
tail -n 3 $O/a.log $O/b.log
