#!/bin/bash
# e2e stage probes: merge / query / filter walls with load traces
cd "$GRAFT_REPO_ROOT" || exit 1
B=$PWD/rufus_amd/bin
D=/dev/shm/rfx_af; rm -rf $D; mkdir -p $D; cd $D
NP=32000000; G=$((NP*10)); NS=$((G/1000000))
$B/rfx_synth_fastq $G 0 $NS 12345 0 $NP c.m1.fq c.m2.fq
$B/rfx_synth_fastq $G 1 $NS 12345 0 $NP mother.fq
$B/rfx_synth_fastq $G 2 $NS 12345 0 $NP father.fq
for n in child mother father; do
  f="$n.fq"; [ $n = child ] && f="c.m1.fq c.m2.fq"
  RFX_COUNT_HISTO=1 $B/jellyfish count --disk -m 25 -L 2 -s 8G -t 14 -o $n.Jhash -C $f
done
t() { local s=$(date +%s.%N); "$@"; local e=$(date +%s.%N); echo "   wall $(echo "$e - $s" | bc) s: $1 $2" >&2; }
for i in 1 2; do
echo "--- merge"
RFX_TRACE_LOAD=1 RFX_CLI_TRACE=1 t $B/jellyfish merge child.Jhash mother.Jhash father.Jhash > merge.txt
done
awk '{print ">"$1"\n"$1}' merge.txt > q.fa
for i in 1 2; do
echo "--- query"
RFX_TRACE_LOAD=1 RFX_CLI_TRACE=1 t $B/jellyfish query -s q.fa child.Jhash > query.txt
done
awk '$2>=5 && $2<=1200' query.txt > child.HashList
wc -l child.HashList
for i in 1 2; do
echo "--- filter"
RFX_CLI_TRACE=1 t $B/RUFUS.Filter child.HashList c.m1.fq c.m2.fq child 25 15 1 14 > /dev/null
echo "--- filter, clean exit"
RFX_CLEAN_EXIT=1 RFX_CLI_TRACE=1 t $B/RUFUS.Filter child.HashList c.m1.fq c.m2.fq childb 25 15 1 14 > /dev/null
done
cmp child.Mutations.Mate1.fastq childb.Mutations.Mate1.fastq && cmp child.Mutations.Mate2.fastq childb.Mutations.Mate2.fastq && echo same outputs
rm -rf $D
