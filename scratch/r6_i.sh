#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_text_gpu.py -x -q -m gpu 2>&1 | tail -15
