#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
RFX_FUZZ_SEEDS=70000-72000 timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "filter_matches_oracle or merge_hashlist" 2>&1 | tail -n 2
