#!/bin/bash
# profiles/r02_* except the bench line itself (scratch/r2_run19.sh made that): kernel stats + PMC traffic of the
# default workload, S1 line.  Run from the repo root on the GPU box.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$R"
O=gpurun_out/profiles_r02; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python bench.py --inner --steps 1 --warmup 1 > $O/stats.log 2>&1
cp "$(find $O/stats -name 's_kernel_stats.csv' | head -1)" $O/r02_kernel_stats.csv
timeout 1200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python bench.py --inner --steps 1 --warmup 0 > $O/f.log 2>&1
timeout 1200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python bench.py --inner --steps 1 --warmup 0 > $O/w.log 2>&1
F=$(find $O/pmc_fetch -name "f_counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "w_counter_collection.csv" | head -1)
python profiles/summarize_pmc.py "$F" "$W" $O/r02_pmc_wgs.json 3 3100000000 > $O/r02_pmc_summary.txt
timeout 300 python bench.py --workload s1 --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end > $O/s1.log 2>/dev/null; tail -1 $O/s1.log > $O/r02_bench_s1.json
rm -rf $O/stats $O/pmc_fetch $O/pmc_write
head -30 $O/r02_pmc_summary.txt; head -16 $O/r02_kernel_stats.csv | cut -c1-160; cut -c1-300 $O/r02_bench_s1.json
