#!/bin/bash
# End-to-end rate of the drop-in RUFUS.Filter on big mate files in tmpfs.  usage: cli_filter_scale.sh [pairs] [genome] [threads]
cd "$GRAFT_REPO_ROOT" || cd "$(dirname "$0")/.." || exit 1
PAIRS=${1:-16000000}; G=${2:-160000000}; T=${3:-64}
D=/dev/shm/rfx_filter_scale; mkdir -p $D
BIN=rufus_amd/bin
$BIN/rfx_synth_fastq $G 0 50 12345 0 $PAIRS $D/m1.fq $D/m2.fq || exit 1
# a hash list: the child's k-mers at the planted SNVs are what RUFUS would find; any k-mer list does for timing
python3 - <<PY
import sys
sys.path.insert(0, ".")
from rufus_amd import capi
sy = capi.Synth.sample($G, 0, n_snv=50, seed=12345)
comp = bytes.maketrans(b"ACGT", b"TGCA")
with open("$D/hl", "w") as f:
    for p, ref, alt in sy.snvs():
        c = bytearray(sy.genome(p - 24, 49)); c[24:25] = alt
        for i in range(25):
            km = bytes(c[i:i + 25]); km = min(km, km[::-1].translate(comp))
            f.write(km.decode() + " 12\n")
PY
s=$(date +%s.%N)
RFX_CLI_TRACE=1 $BIN/RUFUS.Filter $D/hl $D/m1.fq $D/m2.fq $D/out 25 15 1 $T > $D/log.txt || exit 1
e=$(date +%s.%N)
python3 -c "print('cli_filter threads=$T reads=%d wall=%.2fs rate=%.1f M reads/s' % (2*$PAIRS, $e-$s, 2*$PAIRS/($e-$s)/1e6))"
tail -c 200 $D/log.txt | tr '\r' '\n' | tail -2
wc -l $D/out.Mutations.Mate1.fastq $D/out.Mutations.Mate2.fastq
if [ -x oracle/_ref/RUFUS.Filter ] && [ "$PAIRS" -le 2000000 ]; then
  oracle/_ref/RUFUS.Filter $D/hl $D/m1.fq $D/m2.fq $D/ref 25 15 1 1 > /dev/null
  cmp $D/ref.Mutations.Mate1.fastq $D/out.Mutations.Mate1.fastq && cmp $D/ref.Mutations.Mate2.fastq $D/out.Mutations.Mate2.fastq && echo "byte-identical to oracle/_ref/RUFUS.Filter (1 thread)"
fi
rm -rf $D
