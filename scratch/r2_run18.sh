#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r18; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msp or leaf or count" > $O/t.log 2>&1; tail -4 $O/t.log
timeout 600 python bench.py --genome 1000000000 --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/b1g.log 2> $O/b1g.err
tail -1 $O/b1g.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline']['avg_launch_ms_by_kernel'])"
tail -2 $O/b1g.err
