#!/bin/bash
# SURVEY 8 row N1 at scale: SAM text -> `jellyfish count --sam` against `PassThroughSamCheck | jellyfish count`
# (drop-in and, when built, the reference's PassThroughSamCheck as the feeder).  usage: cli_sam_scale.sh [pairs=32000000]
cd "$GRAFT_REPO_ROOT" || exit 1
PAIRS=${1:-32000000}; G=$((PAIRS*10))
D=/dev/shm/rfx_sam; mkdir -p $D; BIN=rufus_amd/bin
RFX_SYNTH_SAM=1 $BIN/rfx_synth_fastq $G 0 100 12345 0 $PAIRS $D/in.sam || exit 1
ls -la $D/in.sam
s=$(date +%s.%N); RFX_CLI_TRACE=1 $BIN/jellyfish count --sam $D/a.chr --disk -m 25 -L 2 -s 8G -t 64 -o $D/a.Jhash -C $D/in.sam 2>&1 | grep "parsed\|closed"; e=$(date +%s.%N)
python3 -c "print('count --sam (file): %.2f s = %.1f M reads/s' % ($e-$s, 2*$PAIRS/($e-$s)/1e6))"
s=$(date +%s.%N); cat $D/in.sam | $BIN/jellyfish count --sam $D/b.chr --disk -m 25 -L 2 -s 8G -t 64 -o $D/b.Jhash -C /dev/stdin; e=$(date +%s.%N)
python3 -c "print('cat | count --sam (pipe): %.2f s = %.1f M reads/s' % ($e-$s, 2*$PAIRS/($e-$s)/1e6))"
s=$(date +%s.%N); $BIN/PassThroughSamCheck $D/c.chr < $D/in.sam | $BIN/jellyfish count --disk -m 25 -L 2 -s 8G -t 64 -o $D/c.Jhash -C /dev/stdin; e=$(date +%s.%N)
python3 -c "print('PassThroughSamCheck (drop-in) | count: %.2f s = %.1f M reads/s' % ($e-$s, 2*$PAIRS/($e-$s)/1e6))"
if [ -x oracle/_ref/PassThroughSamCheck ]; then
  s=$(date +%s.%N); oracle/_ref/PassThroughSamCheck $D/d.chr < $D/in.sam | $BIN/jellyfish count --disk -m 25 -L 2 -s 8G -t 64 -o $D/d.Jhash -C /dev/stdin; e=$(date +%s.%N)
  python3 -c "print('PassThroughSamCheck (reference binary) | count: %.2f s = %.1f M reads/s' % ($e-$s, 2*$PAIRS/($e-$s)/1e6))"
  cmp $D/a.chr $D/d.chr && echo "chr log identical to the reference binary's"
fi
python3 - <<PY
def payload(p):
    import hashlib
    f = open(p, "rb"); n = int(f.read(9)); f.seek(9 + n); h = hashlib.sha256()
    while True:
        b = f.read(1 << 26)
        if not b: break
        h.update(b)
    return h.hexdigest()[:16]
print("payloads:", payload("$D/a.Jhash"), payload("$D/b.Jhash"), payload("$D/c.Jhash"))
PY
cmp $D/a.chr $D/b.chr && cmp $D/a.chr $D/c.chr && echo "chr logs identical"; cat $D/a.chr | tr '\n' ' '; echo
rm -rf $D
