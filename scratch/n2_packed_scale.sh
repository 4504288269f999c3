#!/bin/bash
# SURVEY row N2 at scale: the subject's SAM stream (64 M reads, 22 GB) parsed ONCE.  `cat in.sam | jellyfish count --sam --spool
# --keep-packed` (count + spool + packed-read cache), then RUFUS.Filter on the spool as text (--sam: round 3) and from the
# cache (--packed: round 4); same Mutations.Mate1/2.fastq, same chromosome log.  usage: n2_packed_scale.sh [pairs=32000000]
cd "$GRAFT_REPO_ROOT" || exit 1
PAIRS=${1:-32000000}; G=$((PAIRS*10))
D=/dev/shm/rfx_n2; mkdir -p $D; BIN=$PWD/rufus_amd/bin
RFX_SYNTH_SAM=1 $BIN/rfx_synth_fastq $G 0 50 12345 0 $PAIRS $D/in.sam || exit 1
cd $D
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
t() { python3 -c "print('$1: %.2f s = %.1f M reads/s' % ($3-$2, 2*$PAIRS/($3-$2)/1e6))"; }
s=$(date +%s.%N); cat in.sam | $BIN/jellyfish count --disk -m 25 -L 2 -s 8G -t 64 -C --sam a.chr -o a.Jhash /dev/stdin; e=$(date +%s.%N)
t "jellyfish count --sam (pipe), no spool, no cache" $s $e
s=$(date +%s.%N); cat in.sam | $BIN/jellyfish count --disk -m 25 -L 2 -s 8G -t 64 -C --sam b.chr --spool spool.sam -o b.Jhash /dev/stdin; e=$(date +%s.%N)
t "jellyfish count --sam --spool" $s $e
s=$(date +%s.%N); cat in.sam | $BIN/jellyfish count --disk -m 25 -L 2 -s 8G -t 64 -C --sam c.chr --spool spool.sam --keep-packed cache.bin -o c.Jhash /dev/stdin; e=$(date +%s.%N)
t "jellyfish count --sam --spool --keep-packed" $s $e
ls -l cache.bin spool.sam | awk '{print "   ", $5, $9}'
cmp <(tail -c +2000 a.Jhash | md5sum) <(tail -c +2000 c.Jhash | md5sum) > /dev/null && echo "   (the payload tails of the three counts agree)"
for rep in 1 2; do
s=$(date +%s.%N); RFX_CLI_TRACE=1 $BIN/RUFUS.Filter --sam s.chr hl spool.sam text 25 15 1 64 > log_text.txt 2> trace_text.txt; e=$(date +%s.%N)
t "RUFUS.Filter --sam spool.sam (text route, round 3)" $s $e
s=$(date +%s.%N); RFX_CLI_TRACE=1 $BIN/RUFUS.Filter --packed cache.bin p.chr hl spool.sam packed 25 15 1 64 > log_packed.txt 2> trace_packed.txt; e=$(date +%s.%N)
t "RUFUS.Filter --packed cache.bin (round 4)" $s $e
done
grep "packed cache" log_packed.txt; grep "filter" trace_packed.txt | tail -4
cmp text.Mutations.Mate1.fastq packed.Mutations.Mate1.fastq && cmp text.Mutations.Mate2.fastq packed.Mutations.Mate2.fastq && cmp s.chr p.chr && echo "   same Mutations.Mate1/2.fastq and chr log"; wc -l packed.Mutations.Mate1.fastq
cd /; rm -rf $D
