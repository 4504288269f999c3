#!/bin/bash
# the WHOLE trio of config W through the drop-in executables, FIFO-fed as scripts/RunJellyForRUFUS.sh:23-31
# does it (no text at rest): generator -> named pipe -> jellyfish count x 3 -> modified merge -> query + [MinCov, MaxDepth]
# -> RUFUS.Filter on the subject's two mate pipes; then the N3 lookup of the hash list in all three databases at once.
# usage: cli_w_trio.sh [pairs=310000000] [genome=3100000000]     -> gpurun_out/cli_w_trio/r03_cli_w_trio.txt
cd "$GRAFT_REPO_ROOT" || exit 1
PAIRS=${1:-310000000}; G=${2:-3100000000}; NSNV=${NSNV:-1000}
D=/dev/shm/rfx_trio; rm -rf $D; mkdir -p $D; O=$PWD/gpurun_out/cli_w_trio; mkdir -p $O; BIN=$PWD/rufus_amd/bin
R=$O/r03_cli_w_trio.txt; : > $R
say() { echo "$@" | tee -a $R; }
say "config W through the executables, FIFO-fed: genome $G, $PAIRS pairs per sample (x 150 bp x 2), $NSNV SNVs; $(nproc) hardware threads, cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
T0=$(date +%s.%N)
for w in 0 1 2; do
  mkfifo $D/s$w.fq
  ($BIN/rfx_synth_fastq $G $w $NSNV 12345 0 $PAIRS $D/s$w.fq &)
  s=$(date +%s.%N)
  RFX_COUNT_HISTO=1 RFX_CLI_TRACE=1 timeout 1200 $BIN/jellyfish count --disk -m 25 -L 2 -s 8G -t 64 -o $D/s$w.Jhash -C $D/s$w.fq 2> $O/count$w.trace; rc=$?
  e=$(date +%s.%N)
  n=$(python3 -c "import os; p='$D/s$w.Jhash'; sz=os.path.getsize(p); hl=int(open(p,'rb').read(9)); print((sz-9-hl)//11)")
  say "$(python3 -c "print('sample $w: generator | jellyfish count rc=$rc: %.1f s = %.1f M reads/s, %s records' % ($e-$s, 2*$PAIRS/($e-$s)/1e6, '$n'))")"
  grep "count:\|write:" $O/count$w.trace | tr '\n' ';' | cut -c1-600 >> $R; echo >> $R
  rm $D/s$w.fq
done
s=$(date +%s.%N); timeout 900 $BIN/jellyfish merge $D/s0.Jhash $D/s1.Jhash $D/s2.Jhash > $D/merge.txt; e=$(date +%s.%N)
say "$(python3 -c "print('jellyfish merge (modified): %.1f s' % ($e-$s))"), $(wc -l < $D/merge.txt) lines"
awk '{print ">"$1"\n"$1}' $D/merge.txt > $D/q.fa
s=$(date +%s.%N); $BIN/jellyfish query -s $D/q.fa $D/s0.Jhash | awk '$2 >= 5 && $2 <= 1200' > $D/child.HashList; e=$(date +%s.%N)
say "$(python3 -c "print('jellyfish query + [5,1200]: %.1f s' % ($e-$s))"), $(wc -l < $D/child.HashList) mutant k-mers (library path: 24567)"
mkfifo $D/m1.fq $D/m2.fq
($BIN/rfx_synth_fastq $G 0 $NSNV 12345 0 $PAIRS $D/m1.fq $D/m2.fq &)
s=$(date +%s.%N); (cd $D && RFX_CLI_TRACE=1 timeout 1200 $BIN/RUFUS.Filter child.HashList m1.fq m2.fq child 25 15 1 64 > filter.log 2> $O/filter.trace); e=$(date +%s.%N)
say "$(python3 -c "print('generator | RUFUS.Filter (two mate pipes): %.1f s = %.1f M reads/s' % ($e-$s, 2*$PAIRS/($e-$s)/1e6))"), $(( $(wc -l < $D/child.Mutations.Mate1.fastq) / 4 )) pairs pulled (library path: 11560)"
s=$(date +%s.%N); $BIN/jellyfish query -s $D/q.fa -o $D/l0 -o $D/l1 -o $D/l2 $D/s0.Jhash $D/s1.Jhash $D/s2.Jhash; e=$(date +%s.%N)
say "$(python3 -c "print('N3: one jellyfish query over the three 36 GB databases: %.1f s' % ($e-$s))"); present in child/mother/father: $(awk '$2>0' $D/l0 | wc -l) / $(awk '$2>0' $D/l1 | wc -l) / $(awk '$2>0' $D/l2 | wc -l)"
s=$(date +%s.%N); for w in 0 1 2; do $BIN/jellyfish query -s $D/q.fa $D/s$w.Jhash > $D/x$w; done; e=$(date +%s.%N)
say "$(python3 -c "print('   the same as three calls: %.1f s' % ($e-$s))"); identical: $(cmp -s $D/l0 $D/x0 && cmp -s $D/l1 $D/x1 && cmp -s $D/l2 $D/x2 && echo yes || echo NO)"
say "$(python3 -c "print('whole chain: %.1f s wall' % ($(date +%s.%N)-$T0))")"
head -3 $D/s0.Jhash.histo | tr '\n' ' ' >> $R; echo >> $R
rm -rf $D
