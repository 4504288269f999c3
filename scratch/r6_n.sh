#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_cli_gpu.py tests/test_text_gpu.py -x -q -m gpu -k "text" 2>&1 | tail -5
