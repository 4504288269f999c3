#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_scale_gpu.py -x -q -m gpu -k "two_ranks or blocks_and_passes" > gpurun_out/t_scale.log 2>&1; echo "rc=$?" >> gpurun_out/t_scale.log
tail -n 30 gpurun_out/t_scale.log
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --genome 300000000 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/b_tr1.json 2> gpurun_out/b_tr1.err; echo "rc=$?" >> gpurun_out/b_tr1.err
tail -c 400 gpurun_out/b_tr1.err; cut -c1-400 gpurun_out/b_tr1.json
