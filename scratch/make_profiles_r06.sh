#!/bin/bash
# Regenerates profiles/r06_* on a GPU box (gpurun -- 'bash scratch/make_profiles_r06.sh'); parts by env: STATS PMC TNPMC SQ BENCH EXTRA.
#   STATS  kernel statistics of the LAST step of `bench.py --inner --steps 1 --warmup 1` (profiles/summarize_trace.py)
#   PMC    HBM bytes per kernel, W (FETCH_SIZE / WRITE_SIZE in separate passes)        TNPMC  the same for --workload tn
#   SQ     SQ counters on the 1 Gb slice        BENCH  the driver's line        EXTRA  S1 and TN lines
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$R"
export RFX_COMMIT=${RFX_COMMIT:-$(cat scratch/.commit 2>/dev/null || echo unknown)}
O=gpurun_out/profiles_r06; mkdir -p $O
if [ "${STATS:-1}" = 1 ]; then
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python bench.py --inner --steps 1 --warmup 1 --no-check > $O/stats.log 2>&1
python profiles/summarize_trace.py "$(find $O/trace -name 't_kernel_trace.csv' | head -1)" $O/r06_kernel_stats_wgs.csv
python - <<'PY'
# idle gaps and outliers of the last step (what averages hide: VERDICT r5 #5; round 6 found the TN rerun this way)
import csv, glob, re, collections
f = glob.glob('gpurun_out/profiles_r06/trace/**/t_kernel_trace.csv', recursive=True)[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.search(r"(k_[a-z0-9_]+|__amd_rocclr_[A-Za-z]+)", r["Kernel_Name"]).group(1)) for r in csv.DictReader(open(f)) if re.search(r"(k_[a-z0-9_]+|__amd_rocclr_[A-Za-z]+)", r["Kernel_Name"]))
isf = lambda n: n.startswith("k_filter") or n in ("k_hits_mask", "k_mask_count")
ends = []
for i, r in enumerate(rows):
    if isf(r[2]):
        nxt = next((rows[j][2] for j in range(i + 1, len(rows)) if rows[j][2].startswith("k_")), None)
        if nxt is None or not isf(nxt): ends.append(i)
last = rows[ends[-2] + 1: ends[-1] + 1] if len(ends) > 1 else rows
busy = sum(e - s for s, e, _ in last); wall = last[-1][1] - last[0][0]
gaps = sorted(((last[i + 1][0] - last[i][1], last[i][2], last[i + 1][2]) for i in range(len(last) - 1)), reverse=True)
print("last step: %d launches, wall %.1f ms, kernels %.1f ms, idle %.1f ms" % (len(last), wall / 1e6, busy / 1e6, (wall - busy) / 1e6))
print("largest gaps (us, after, before):", [(round(g / 1e3), a, b) for g, a, b in gaps[:12]])
by = collections.defaultdict(list)
for s, e, n in last: by[n].append((e - s) / 1e3)
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:14]:
    v.sort(); print("%-22s n=%4d  sum %8.1f ms  median %8.1f us  max %8.1f us" % (n, len(v), sum(v) / 1e3, v[len(v) // 2], v[-1]))
PY
rm -rf $O/trace
fi
pmc() {  # $1 = bench args, $2 = out json, $3 = samples, $4 = out txt
  timeout 1500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python bench.py --inner --steps 1 --warmup 0 --no-check $1 > $O/f.log 2>&1
  timeout 1500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python bench.py --inner --steps 1 --warmup 0 --no-check $1 > $O/w.log 2>&1
  F=$(find $O/pmc_fetch -name "f_counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "w_counter_collection.csv" | head -1)
  python profiles/summarize_pmc.py "$F" "$W" $2 $3 3100000000 > $4
  rm -rf $O/pmc_fetch $O/pmc_write
  head -30 $4
}
if [ "${PMC:-1}" = 1 ]; then
export RFX_BENCH_MAP_BUDGET=${RFX_BENCH_MAP_BUDGET:-39.8e9}
pmc "" $O/r06_pmc_wgs.json 3 $O/r06_pmc_summary_wgs.txt
cp $O/r06_pmc_wgs.json profiles/r06_pmc_wgs.json   # (the bench line below quotes it: it was taken on this very build)
unset RFX_BENCH_MAP_BUDGET
fi
if [ "${TNPMC:-1}" = 1 ]; then
export RFX_BENCH_MAP_BUDGET=${RFX_TN_MAP_BUDGET:-59.7e9}
pmc "--workload tn --passes 4" $O/r06_pmc_tn.json 2 $O/r06_pmc_summary_tn.txt
cp $O/r06_pmc_tn.json profiles/r06_pmc_tn.json
unset RFX_BENCH_MAP_BUDGET
fi
if [ "${SQ:-1}" = 1 ]; then
bash scratch/r6_sq.sh > $O/sq.log 2>&1; cp gpurun_out/r6_sq.txt $O/r06_sq_counters_1g.txt; head -30 $O/r06_sq_counters_1g.txt
fi
if [ "${BENCH:-1}" = 1 ]; then
timeout 1700 python3 bench.py --gpus 1 --steps ${STEPS:-20} --warmup 5 > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log > $O/r06_bench.json
python3 -c "
import json; d=json.load(open('$O/r06_bench.json')); print(d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline'].get('traffic'), d['roofline_filter']['frac'], d['config'].get('checks')); print(json.dumps(d.get('end_to_end'))[:700])"
tail -2 $O/bench.err
fi
if [ "${EXTRA:-1}" = 1 ]; then
timeout 300 python bench.py --workload s1 --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end > $O/s1.log 2>/dev/null; tail -1 $O/s1.log > $O/r06_bench_s1.json; cut -c1-220 $O/r06_bench_s1.json
timeout 1200 python bench.py --workload tn --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end > $O/tn.log 2>$O/tn.err; tail -1 $O/tn.log > $O/r06_bench_tn_full.json; cut -c1-220 $O/r06_bench_tn_full.json; tail -2 $O/tn.err
fi
