#!/bin/bash
# Regenerates profiles/r05_* on a GPU box (run from the repo root: gpurun -- 'bash scratch/make_profiles_r05.sh'):
# kernel stats and PMC traffic of the default workload (one step, no self-check), SQ counters on the 1 Gb slice,
# the default bench line with the driver's flags, the S1 and full-size tumor/normal lines.
set -u
# (runs without a warm-up step get the run-map pool from the start: bench.py sets it after the first step otherwise)
export RFX_BENCH_MAP_BUDGET=${RFX_BENCH_MAP_BUDGET:-39.8e9}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$R"
O=gpurun_out/profiles_r05; mkdir -p $O
if [ "${STATS:-1}" = 1 ]; then
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python bench.py --inner --steps 1 --warmup 1 --no-check > $O/stats.log 2>&1
cp "$(find $O/stats -name 's_kernel_stats.csv' | head -1)" $O/r05_kernel_stats_wgs.csv
rm -rf $O/stats
fi
if [ "${PMC:-1}" = 1 ]; then
timeout 1200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python bench.py --inner --steps 1 --warmup 0 --no-check > $O/f.log 2>&1
timeout 1200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python bench.py --inner --steps 1 --warmup 0 --no-check > $O/w.log 2>&1
F=$(find $O/pmc_fetch -name "f_counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "w_counter_collection.csv" | head -1)
python profiles/summarize_pmc.py "$F" "$W" $O/r05_pmc_wgs.json 3 3100000000 > $O/r05_pmc_summary_wgs.txt
rm -rf $O/pmc_fetch $O/pmc_write
cp $O/r05_pmc_wgs.json profiles/r05_pmc_wgs.json   # (the bench line below quotes it: it was taken on this very build)
head -40 $O/r05_pmc_summary_wgs.txt
fi
if [ "${SQ:-1}" = 1 ]; then
S=$O/sq; rm -rf $S; mkdir -p $S
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $S/a -o s -- python bench.py --inner --genome 1000000000 --passes 2 --steps 1 --warmup 0 --no-check > $S/a.log 2>&1
timeout 900 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_VMEM --kernel-trace --output-format csv -d $S/b -o s -- python bench.py --inner --genome 1000000000 --passes 2 --steps 1 --warmup 0 --no-check > $S/b.log 2>&1
python - <<'PY'
import csv, collections, re, glob
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob('gpurun_out/profiles_r05/sq/*/**/s_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"][:30]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "SQ_INSTS_LDS"):
            calls[(k, r["Counter_Name"])] += 1
out = ["# SQ counters, summed over all launches of `python bench.py --inner --genome 1000000000 --passes 2 --steps 1 --warmup 0 --no-check` (30x trio on a",
       "# 1 Gb genome, TWO shard passes with run maps, one step: 3 samples x 2e8 reads = 9e10 bases = 1.4e9 wave-rows of 64 bases).  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_*",
       "# count quad-cycles.", ""]
for k in ("k_msp_part1", "k_msp_replay", "k_part2", "k_msp_leaf", "k_surv_sort", "k_bin_hist", "k_surv_hist", "k_filter_q", "k_hits_mask", "k_flag_absent_tiled"):
    if k in agg:
        out.append(k + "  launches=%d" % max(calls[(k, "SQ_WAVE_CYCLES")], 1))
        for c, v in sorted(agg[k].items()):
            out.append(f"    {c:24s} {v:.4g}")
open("gpurun_out/profiles_r05/r05_sq_counters_1g.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out[:40]))
PY
rm -rf $S
fi
if [ "${BENCH:-1}" = 1 ]; then
unset RFX_BENCH_MAP_BUDGET
timeout 1700 python3 bench.py --gpus 1 --steps ${STEPS:-20} --warmup 5 > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log > $O/r05_bench.json
python3 -c "
import json; d=json.load(open('$O/r05_bench.json')); print(d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline_filter']['frac'], d['config'].get('checks')); print(json.dumps(d.get('end_to_end'))[:700])"
tail -2 $O/bench.err
fi
if [ "${EXTRA:-1}" = 1 ]; then
timeout 300 python bench.py --workload s1 --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end > $O/s1.log 2>/dev/null; tail -1 $O/s1.log > $O/r05_bench_s1.json; cut -c1-220 $O/r05_bench_s1.json
timeout 900 python bench.py --workload tn --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/tn.log 2>$O/tn.err; tail -1 $O/tn.log > $O/r05_bench_tn_full.json; cut -c1-220 $O/r05_bench_tn_full.json; tail -2 $O/tn.err
fi
