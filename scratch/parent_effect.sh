#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
D=/dev/shm/rfx_pe; mkdir -p $D; BIN=rufus_amd/bin
$BIN/rfx_synth_fastq 320000000 0 100 12345 0 32000000 $D/reads.fq || exit 1
run() { s=$(date +%s.%N); RFX_CLI_TRACE=1 $BIN/jellyfish count --disk -m 25 -L 2 -s 8G -t 64 -o $D/out.Jhash -C $D/reads.fq 2> $D/trace; e=$(date +%s.%N); python3 -c "print('$1: count wall %.2f s' % ($e-$s))"; grep -E "device open|parsed|payload out|output closed" $D/trace | tr '\n' ' '; echo; }
run "alone"
python3 - <<'PY' &
import time, torch
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
time.sleep(25)
PY
sleep 8
run "torch parent idle (cuda context open)"
wait
python3 - <<'PY' &
import time
from rufus_amd import capi
c = capi.Context(0); time.sleep(14)
PY
sleep 5
run "rufus ctx open in another process"
wait
rm -rf $D
