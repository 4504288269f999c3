#!/bin/bash
# round 3, batch 21: ModelDist (row N4) -- parity tests, wall time of the executable, kernel stats
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/exp21; mkdir -p $O
timeout 900 python -m pytest tests/test_modeldist.py -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log; mkdir -p gpurun_out; cat gpurun_out/modeldist_wall.txt
mkdir -p /tmp/md && cp tests/golden/modeldist/child.histo /tmp/md/ && cd /tmp/md
for i in 1 2 3; do s=$(date +%s%N); RFX_CLI_TRACE=1 $R/rufus_amd/bin/ModelDist child.histo 25 150 8 2>$R/$O/trace_$i.txt >/dev/null; e=$(date +%s%N); echo "ModelDist wall $(( (e - s) / 1000000 )) ms"; cat $R/$O/trace_$i.txt; done
head -4 child.histo.7.7.model
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o md -- rufus_amd/bin/ModelDist /tmp/md/child.histo 25 150 8 > /dev/null 2>$O/prof.err
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_modeldist.csv && head -8 $f
