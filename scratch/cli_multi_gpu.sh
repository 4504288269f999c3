#!/bin/bash
# E-cli at scale on the one-GPU box: a 30x sample of 64 M reads counted by one context and by RUFUS_GPUS=0,0 / 0,0,0,0 (n
# contexts on the device: the multi-device code path); payloads must be identical.  usage: cli_multi_gpu.sh [pairs=32000000]
cd "$GRAFT_REPO_ROOT" || exit 1
PAIRS=${1:-32000000}; G=$((PAIRS*10))
D=/dev/shm/rfx_mg; rm -rf $D; mkdir -p $D; BIN=$PWD/rufus_amd/bin
$BIN/rfx_synth_fastq $G 0 100 12345 0 $PAIRS $D/s.fq || exit 1
for g in "" "0,0" "0,0,0,0"; do
  s=$(date +%s.%N)
  RUFUS_GPUS=$g RFX_CLI_TRACE=1 $BIN/jellyfish count --disk -m 25 -L 2 -s 8G -t 64 -o $D/o_${g//,/}.Jhash -C $D/s.fq 2> $D/trace_${g//,/}.txt; rc=$?
  e=$(date +%s.%N)
  python3 -c "print('RUFUS_GPUS=\"$g\" rc=$rc: %.2f s = %.1f M reads/s' % ($e-$s, 2*$PAIRS/($e-$s)/1e6))"
  grep "finished on the device\|parsed and queued\|output closed" $D/trace_${g//,/}.txt | tr '\n' ';'; echo
done
python3 - <<PY
import hashlib
def payload(p):
    f = open(p, "rb"); n = int(f.read(9)); f.seek(9 + n); h = hashlib.sha256()
    while True:
        b = f.read(1 << 26)
        if not b: break
        h.update(b)
    return h.hexdigest()[:16]
a, b, c = payload("$D/o_.Jhash"), payload("$D/o_00.Jhash"), payload("$D/o_0000.Jhash")
print("payloads:", a, b, c, "IDENTICAL" if a == b == c else "DIFFERENT")
PY
rm -rf $D
