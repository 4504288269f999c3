#!/bin/bash
# The modified `jellyfish merge` and `query` on three 30x samples of a 1 Gb genome (3 x 11.5 GB .Jhash): position-range
# slices (default plan) against one range.  usage: cli_merge_scale.sh [pairs=100000000] [genome=1000000000]
cd "$GRAFT_REPO_ROOT" || exit 1
PAIRS=${1:-100000000}; G=${2:-1000000000}
D=/dev/shm/rfx_ms; mkdir -p $D; BIN=rufus_amd/bin
for w in 0 1 2; do
  $BIN/rfx_synth_fastq $G $w 1000 12345 0 $PAIRS $D/s.fq || exit 1
  s=$(date +%s.%N); $BIN/jellyfish count --disk -m 25 -L 2 -s 8G -t 64 -o $D/s$w.Jhash -C $D/s.fq || exit 1; e=$(date +%s.%N)
  python3 -c "print('count sample $w: %.1f s' % ($e-$s))"; rm $D/s.fq
done
ls -la $D
cd $D
s=$(date +%s.%N); RFX_CLI_TRACE=1 $OLDPWD/$BIN/jellyfish merge s0.Jhash s1.Jhash s2.Jhash > merge.sliced 2> trace.sliced || exit 1; e=$(date +%s.%N)
python3 -c "print('merge, default plan: %.1f s' % ($e-$s))"
s=$(date +%s.%N); RFX_MERGE_SLICES=1 $OLDPWD/$BIN/jellyfish merge s0.Jhash s1.Jhash s2.Jhash > merge.one || exit 1; e=$(date +%s.%N)
python3 -c "print('merge, one range: %.1f s' % ($e-$s))"
cmp merge.sliced merge.one && echo "identical output"; wc -l merge.sliced
awk '{print ">" $1 "\n" $1}' merge.sliced > q.fa
s=$(date +%s.%N); $OLDPWD/$BIN/jellyfish query -s q.fa s0.Jhash > query.txt || exit 1; e=$(date +%s.%N)
python3 -c "print('query of %d k-mers: %.1f s' % (sum(1 for _ in open('query.txt')), $e-$s))"
awk '$2 >= 5 && $2 <= 1200' query.txt | wc -l
cd /; rm -rf $D
