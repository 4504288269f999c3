#!/bin/bash
# Regenerates profiles/ on a GPU box: bench line, rocprofv3 kernel stats, PMC traffic.  Run from the repo root.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$R"
O=gpurun_out/profiles; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/w.log 2>&1
F=$(find $O/pmc_fetch -name "f_counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "w_counter_collection.csv" | head -1)
python profiles/summarize_pmc.py $F $W $O/r01_pmc.json > $O/r01_pmc_summary.txt
cp $O/r01_pmc.json profiles/r01_pmc.json   # so that the bench line below carries the traffic of THIS build
cp $(find $O/stats -name "s_kernel_stats.csv" | head -1) $O/r01_kernel_stats.csv
timeout 900 python bench.py > $O/bench.log 2>$O/bench.err
tail -1 $O/bench.log > $O/r01_bench.json
cat $O/r01_pmc_summary.txt | head -25; head -12 $O/r01_kernel_stats.csv | cut -c1-150; cat $O/r01_bench.json
