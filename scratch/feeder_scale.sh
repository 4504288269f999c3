#!/bin/bash
# The stranded SAM feeder into named pipes (runRufus.sh:964-967): alone (pipes drained by cat) and in front of the
# drop-in RUFUS.Filter.  usage: feeder_scale.sh [pairs=16000000]
cd "$GRAFT_REPO_ROOT" || exit 1
PAIRS=${1:-16000000}; G=$((PAIRS*10))
D=/dev/shm/rfx_feed; mkdir -p $D; BIN=$PWD/rufus_amd/bin; REF=$PWD/oracle/_ref
RFX_SYNTH_SAM=1 $BIN/rfx_synth_fastq $G 0 50 12345 0 $PAIRS $D/in.sam || exit 1
cd $D
for t in 0 4 8; do
  rm -f p.mate1.fastq p.mate2.fastq; mkfifo p.mate1.fastq p.mate2.fastq
  (cat p.mate1.fastq > /dev/null &); (cat p.mate2.fastq > /dev/null &)
  s=$(date +%s.%N); RFX_PTS_THREADS=$t $BIN/PassThroughSamCheck.stranded x.chr p < in.sam; e=$(date +%s.%N)
  python3 -c "print('feeder, $t helpers -> drained pipes: %.2f s = %.1f M reads/s' % ($e-$s, 2*$PAIRS/($e-$s)/1e6))"; sleep 0.2
done
python3 - <<PY
import sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from rufus_amd import capi
sy = capi.Synth.sample($G, 0, n_snv=50, seed=12345)
comp = bytes.maketrans(b"ACGT", b"TGCA")
with open("hl", "w") as f:
    for p, ref, alt in sy.snvs():
        c = bytearray(sy.genome(p - 24, 49)); c[24:25] = alt
        for i in range(25):
            km = bytes(c[i:i + 25]); f.write(min(km, km[::-1].translate(comp)).decode() + " 12\n")
PY
rm -f p.mate1.fastq p.mate2.fastq; mkfifo p.mate1.fastq p.mate2.fastq
s=$(date +%s.%N); ($BIN/PassThroughSamCheck.stranded y.chr p < in.sam &); $BIN/RUFUS.Filter hl p.mate1.fastq p.mate2.fastq out 25 15 1 64 > log.txt; e=$(date +%s.%N)
python3 -c "print('feeder | RUFUS.Filter (both drop-in, named pipes): %.2f s = %.1f M reads/s' % ($e-$s, 2*$PAIRS/($e-$s)/1e6))"; wc -l out.Mutations.Mate1.fastq
cp out.Mutations.Mate1.fastq two.m1; cp out.Mutations.Mate2.fastq two.m2
for src in file pipe; do
  s=$(date +%s.%N)
  if [ $src = file ]; then RFX_CLI_TRACE=1 $BIN/RUFUS.Filter --sam s.chr hl in.sam sam 25 15 1 64 > log2.txt 2> trace.txt
  else cat in.sam | $BIN/RUFUS.Filter --sam s.chr hl stdin sam 25 15 1 64 > log2.txt; fi
  e=$(date +%s.%N)
  python3 -c "print('RUFUS.Filter --sam ($src): %.2f s = %.1f M reads/s' % ($e-$s, 2*$PAIRS/($e-$s)/1e6))"
  cmp sam.Mutations.Mate1.fastq two.m1 && cmp sam.Mutations.Mate2.fastq two.m2 && cmp s.chr y.chr && echo "   same Mutations.Mate1/2.fastq and chr log as the two-process route"
done
grep "filter" trace.txt | tail -3
if [ -x $REF/PassThroughSamCheck.stranded ]; then
  head -n 2000000 in.sam > small.sam
  rm -f p.mate1.fastq p.mate2.fastq; mkfifo p.mate1.fastq p.mate2.fastq
  (cat p.mate1.fastq > /dev/null &); (cat p.mate2.fastq > /dev/null &)
  s=$(date +%s.%N); $REF/PassThroughSamCheck.stranded z.chr p < small.sam; e=$(date +%s.%N)
  python3 -c "print('reference feeder -> drained pipes (2 M reads): %.2f s = %.2f M reads/s' % ($e-$s, 2e6/($e-$s)/1e6))"
fi
cd /; rm -rf $D
