#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_cli_gpu.py -x -q -m gpu 2>&1 | tail -n 2
for i in 1 2; do timeout 900 python bench.py --end-to-end-only 2>/dev/null | tail -n 1 | cut -c1-640; done
