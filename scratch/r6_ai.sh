#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
echo "=== fresh box"; scratch/ubench/vmm_cost; scratch/ubench/vmm_cost | grep "total\|Reserve\|chunk 0" -A0
timeout 1500 python -m pytest tests/test_cli_gpu.py -x -q -m gpu 2>&1 | tail -n 1
echo "=== after the CLI tests"; scratch/ubench/vmm_cost; scratch/ubench/vmm_cost | grep "total"
