#!/bin/bash
# The trio of config W through the drop-in executables, FILE-fed (the text generator writes one sample's FASTQ into tmpfs, the
# count reads it, the text is removed): what the tools take without the generator's 16-CPU quota in the way.
# count x 3 -> modified merge -> query + [MinCov, MaxDepth].   usage: cli_w_trio_files.sh [pairs=310000000] [genome=3100000000]
cd "$GRAFT_REPO_ROOT" || exit 1
PAIRS=${1:-310000000}; G=${2:-3100000000}; NSNV=${NSNV:-1000}
D=/dev/shm/rfx_trio; rm -rf $D; mkdir -p $D; O=$PWD/gpurun_out/cli_w_trio_files; mkdir -p $O; BIN=$PWD/rufus_amd/bin
echo "config W through the executables, file-fed: genome $G, $PAIRS pairs per sample (x 150 bp x 2), $NSNV SNVs; $(nproc) hardware threads, cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
for w in 0 1 2; do
  $BIN/rfx_synth_fastq $G $w $NSNV 12345 0 $PAIRS $D/s$w.fq || exit 1
  s=$(date +%s.%N)
  RFX_COUNT_HISTO=1 RFX_CLI_TRACE=1 timeout 1200 $BIN/jellyfish count --disk -m 25 -L 2 -s 8G -t 64 -o $D/s$w.Jhash -C $D/s$w.fq 2> $O/count$w.trace; rc=$?
  e=$(date +%s.%N)
  n=$(python3 -c "import os; p='$D/s$w.Jhash'; sz=os.path.getsize(p); hl=int(open(p,'rb').read(9)); print((sz-9-hl)//11)")
  python3 -c "print('sample $w: jellyfish count rc=$rc: %.1f s = %.1f M reads/s, %s records' % ($e-$s, 2*$PAIRS/($e-$s)/1e6, '$n'))"
  grep "count:\|write:" $O/count$w.trace | tr '\n' ';' | cut -c1-900; echo
  rm $D/s$w.fq
done
for rep in 1 2; do
  s=$(date +%s.%N); RFX_CLI_TRACE=1 ${MERGE_ENV:-} timeout 900 $BIN/jellyfish merge $D/s0.Jhash $D/s1.Jhash $D/s2.Jhash > $D/merge.txt 2> $O/merge$rep.trace; e=$(date +%s.%N)
  python3 -c "print('jellyfish merge (modified), run $rep: %.1f s' % ($e-$s))"; wc -l < $D/merge.txt
  grep "merge:\|main" $O/merge$rep.trace | tr '\n' ';'; echo
done
md5sum $D/merge.txt
awk '{print ">"$1"\n"$1}' $D/merge.txt > $D/q.fa
s=$(date +%s.%N); $BIN/jellyfish query -s $D/q.fa $D/s0.Jhash | awk '$2 >= 5 && $2 <= 1200' > $D/child.HashList; e=$(date +%s.%N)
python3 -c "print('jellyfish query + [5,1200]: %.1f s' % ($e-$s))"; echo "$(wc -l < $D/child.HashList) mutant k-mers (library path: 24567)"
rm -rf $D
