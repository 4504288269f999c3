#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_scale_gpu.py -x -q -m gpu -k "made_ahead or gives_its_own" 2>&1 | tail -12
