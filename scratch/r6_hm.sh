#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1200 python -m pytest tests/test_cli_gpu.py -x -q -m gpu -k "100k" 2>&1 | tail -n 5 | cut -c1-300
