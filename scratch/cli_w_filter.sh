#!/bin/bash
# Config W through the drop-in RUFUS.Filter: the subject's 3.1e8 read pairs (2 x 98 GB of FASTQ in tmpfs) against the
# k-mers of the 1000 planted SNVs.  usage: cli_w_filter.sh [pairs=310000000] [genome=3100000000]
cd "$GRAFT_REPO_ROOT" || exit 1
PAIRS=${1:-310000000}; G=${2:-3100000000}
D=/dev/shm/rfx_wf; mkdir -p $D; O=gpurun_out/cli_w; mkdir -p $O; BIN=rufus_amd/bin
s=$(date +%s.%N); $BIN/rfx_synth_fastq $G 0 1000 12345 0 $PAIRS $D/m1.fq $D/m2.fq || exit 1; e=$(date +%s.%N)
python3 -c "print('generate: %.1f s' % ($e-$s))"; ls -la $D/m1.fq $D/m2.fq
python3 - <<PY
import sys
sys.path.insert(0, ".")
from rufus_amd import capi
sy = capi.Synth.sample($G, 0, n_snv=1000, seed=12345)
comp = bytes.maketrans(b"ACGT", b"TGCA")
with open("$D/hl", "w") as f:
    for p, ref, alt in sy.snvs():
        c = bytearray(sy.genome(p - 24, 49)); c[24:25] = alt
        for i in range(25):
            km = bytes(c[i:i + 25]); km = min(km, km[::-1].translate(comp))
            f.write(km.decode() + " 12\n")
PY
wc -l $D/hl
s=$(date +%s.%N)
RFX_CLI_TRACE=1 timeout 900 $BIN/RUFUS.Filter $D/hl $D/m1.fq $D/m2.fq $D/out 25 15 1 64 > $D/log.txt 2> $O/filter.trace; rc=$?
e=$(date +%s.%N)
python3 -c "print('RUFUS.Filter rc=$rc: wall %.1f s = %.1f M reads/s' % ($e-$s, 2*$PAIRS/($e-$s)/1e6))"
cat $O/filter.trace; tail -c 120 $D/log.txt | tr '\r' '\n' | tail -2
python3 -c "n=sum(1 for _ in open('$D/out.Mutations.Mate1.fastq'))//4; print('pulled pairs:', n, '(library path, hash list of 24567 k-mers found by the trio: 11560)')"
rm -rf $D
