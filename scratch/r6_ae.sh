#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python bench.py --end-to-end-only > gpurun_out/e2e_trace.txt 2>&1
tail -c 1500 gpurun_out/e2e_trace.txt
