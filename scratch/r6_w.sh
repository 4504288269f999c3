#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_cli_gpu.py -x -q -m gpu 2>&1 | tail -3
PAIRS=32000000; G=320000000
D=/dev/shm/rfx_cli_scale; mkdir -p $D; O=gpurun_out/cli_trace6; mkdir -p $O
BIN=rufus_amd/bin
$BIN/rfx_synth_fastq $G 0 100 12345 0 $PAIRS $D/reads.fq || exit 1
for i in 1 2 3; do
  s=$(date +%s.%N)
  RFX_CLI_TRACE=1 $BIN/jellyfish count --disk -m 25 -L 2 -s 8G -t 64 -o $D/out.Jhash -C $D/reads.fq 2> $O/trace.w$i
  e=$(date +%s.%N)
  python3 -c "print('count wall=%.2fs' % ($e-$s))"
  grep -E "device open|staging|input parsed|finished on|output closed" $O/trace.w$i | cut -c1-80
done
rm -rf $D
timeout 900 python bench.py --end-to-end-only 2>gpurun_out/r6w_e2e.err | tail -1 > gpurun_out/r6w_e2e.json
python - <<PY
import json
d=json.load(open("gpurun_out/r6w_e2e.json"))
print(d["stages_s"], "value %.2f M" % (d["value"]/1e6), "pj %.2f M" % (d["parallel_jelly"]["value"]/1e6), d["parallel_jelly"]["jellyfish count x 3_s"])
PY
