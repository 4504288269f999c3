#!/bin/bash
# jellyfish count reading a pipe (what RunJellyForRUFUS.sh does): cat | count, FASTQ and SAM
cd "$GRAFT_REPO_ROOT" || exit 1
PAIRS=${1:-32000000}; G=$((PAIRS*10))
D=/dev/shm/rfx_pipe; mkdir -p $D; BIN=rufus_amd/bin
$BIN/rfx_synth_fastq $G 0 100 12345 0 $PAIRS $D/in.fq || exit 1
s=$(date +%s.%N); cat $D/in.fq > /dev/null; e=$(date +%s.%N); python3 -c "print('cat > /dev/null: %.2f s' % ($e-$s))"
s=$(date +%s.%N); cat $D/in.fq | cat > /dev/null; e=$(date +%s.%N); python3 -c "print('cat | cat > /dev/null: %.2f s (what a pipe carries)' % ($e-$s))"
s=$(date +%s.%N); cat $D/in.fq | RFX_CLI_TRACE=1 $BIN/jellyfish count --disk -m 25 -L 2 -s 8G -t 64 -o $D/a.Jhash -C /dev/stdin 2>&1 | grep "parsed\|finished\|closed"; e=$(date +%s.%N)
python3 -c "print('cat | count (FASTQ pipe): %.2f s = %.1f M reads/s' % ($e-$s, 2*$PAIRS/($e-$s)/1e6))"
rm -rf $D
