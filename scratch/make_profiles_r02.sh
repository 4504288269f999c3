#!/bin/bash
# Regenerates profiles/r02_* on a GPU box: bench line (full default workload, with the CPU baseline), rocprofv3
# kernel stats and PMC traffic of the same command.  Run from the repo root (gpurun: bash scratch/make_profiles_r02.sh).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$R"
O=gpurun_out/profiles_r02; mkdir -p $O
# 1. the default bench (driver's flags would be --steps 20 --warmup 5; 3 + 1 here keeps the box time down)
timeout 1500 python bench.py --steps 3 --warmup 1 > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log > $O/r02_bench.json
# 2. kernel stats of the same command (1 step)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python bench.py --inner --steps 1 --warmup 1 --no-cpu-baseline > $O/stats.log 2>&1
cp "$(find $O/stats -name 's_kernel_stats.csv' | head -1)" $O/r02_kernel_stats.csv
# 3. PMC passes (counters in their own runs, kernel-trace only)
timeout 1200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python bench.py --inner --steps 1 --warmup 0 --no-cpu-baseline > $O/f.log 2>&1
timeout 1200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python bench.py --inner --steps 1 --warmup 0 --no-cpu-baseline > $O/w.log 2>&1
F=$(find $O/pmc_fetch -name "f_counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "w_counter_collection.csv" | head -1)
python profiles/summarize_pmc.py "$F" "$W" $O/r02_pmc_wgs.json 3 3100000000 > $O/r02_pmc_summary.txt
# 4. S1 (configs[1]) for comparison with round 1
timeout 300 python bench.py --workload s1 --steps 20 --warmup 5 --no-cpu-baseline > $O/s1.log 2>/dev/null; tail -1 $O/s1.log > $O/r02_bench_s1.json
rm -rf $O/stats $O/pmc_fetch $O/pmc_write
head -30 $O/r02_pmc_summary.txt; head -16 $O/r02_kernel_stats.csv | cut -c1-160; cut -c1-600 $O/r02_bench.json; tail -3 $O/bench.err
