#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
grep -H . /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag /sys/kernel/mm/transparent_hugepage/shmem_enabled 2>/dev/null
free -g | head -3
RFX_TRACE_LOAD=1 RFX_CLI_TRACE=1 timeout 900 python bench.py --end-to-end-only > gpurun_out/e2e_trace.txt 2>&1
free -g | head -3
tail -c 700 gpurun_out/e2e_trace.txt
