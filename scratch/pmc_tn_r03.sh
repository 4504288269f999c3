#!/bin/bash
# PMC traffic of the full-size tumor/normal workload (configs[4]) -> profiles/r03_pmc_tn.json (bench.py --workload tn reads it)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$R"
O=gpurun_out/profiles_r03; mkdir -p $O
timeout 1500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_tn -o f -- python bench.py --inner --workload tn --steps 1 --warmup 0 --no-check > $O/f_tn.log 2>&1
timeout 1500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_tn -o w -- python bench.py --inner --workload tn --steps 1 --warmup 0 --no-check > $O/w_tn.log 2>&1
F=$(find $O/pmc_fetch_tn -name "f_counter_collection.csv" | head -1); W=$(find $O/pmc_write_tn -name "w_counter_collection.csv" | head -1)
python profiles/summarize_pmc.py "$F" "$W" $O/r03_pmc_tn.json 2 3100000000 > $O/r03_pmc_summary_tn.txt
rm -rf $O/pmc_fetch_tn $O/pmc_write_tn
head -12 $O/r03_pmc_summary_tn.txt
