#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_cli_gpu.py -x -q -m gpu 2>&1 | tail -n 1
RFX_TRACE_LOAD=1 RFX_CLI_TRACE=1 timeout 900 python bench.py --end-to-end-only > gpurun_out/e2e_trace.txt 2>&1
tail -n 1 gpurun_out/e2e_trace.txt | cut -c1-400
