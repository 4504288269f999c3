#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_scale_gpu.py::test_wgs_slice_properties > gpurun_out/t_all.log 2>&1; echo "rc=$?" >> gpurun_out/t_all.log
tail -n 15 gpurun_out/t_all.log
