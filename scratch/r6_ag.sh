#!/bin/bash
# in-box A/B: (a) filter device open async vs RFX_SYNC_OPEN=1; (b) count with/without dropped page-table entries
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_cli_gpu.py tests/test_jellyfish_gpu.py -x -q -m gpu 2>&1 | tail -n 3
B=$PWD/rufus_amd/bin
D=/dev/shm/rfx_ag; rm -rf $D; mkdir -p $D; cd $D
NP=32000000; G=$((NP*10)); NS=$((G/1000000))
$B/rfx_synth_fastq $G 0 $NS 12345 0 $NP c.m1.fq c.m2.fq
t() { local s=$EPOCHREALTIME; "$@" > /dev/null 2>&1; local e=$EPOCHREALTIME; python3 -c "print('   wall %.3f s' % ($e - $s))"; }
RFX_COUNT_HISTO=1 $B/jellyfish count --disk -m 25 -L 2 -s 8G -t 14 -o child.Jhash -C c.m1.fq c.m2.fq
$B/jellyfish dump -c -L 40 child.Jhash 2>/dev/null | head -n 8000 > child.HashList
for i in 1 2 3; do
  echo "count (all drops)"; t $B/jellyfish count --disk -m 25 -L 2 -s 8G -t 14 -o x.Jhash -C c.m1.fq c.m2.fq
  echo "count RFX_KEEP_PTES=1"; RFX_KEEP_PTES=1 t $B/jellyfish count --disk -m 25 -L 2 -s 8G -t 14 -o y.Jhash -C c.m1.fq c.m2.fq
  echo "filter"; t $B/RUFUS.Filter child.HashList c.m1.fq c.m2.fq x 25 15 1 14
  echo "filter RFX_SYNC_OPEN=1"; RFX_SYNC_OPEN=1 t $B/RUFUS.Filter child.HashList c.m1.fq c.m2.fq y 25 15 1 14
done
RFX_CLI_TRACE=1 $B/jellyfish count --disk -m 25 -L 2 -s 8G -t 14 -o x.Jhash -C c.m1.fq c.m2.fq 2>&1 | grep rfx
RFX_CLI_TRACE=1 $B/RUFUS.Filter child.HashList c.m1.fq c.m2.fq x 25 15 1 14 2>&1 | grep "rfx "
cmp x.Mutations.Mate1.fastq y.Mutations.Mate1.fastq && cmp x.Mutations.Mate2.fastq y.Mutations.Mate2.fastq && echo "filter outputs equal"
python3 - <<'P'
import sys
def payload(p):
    f=open(p,'rb'); n=int(f.read(9)); f.seek(9+n); return f.read()
print("count payloads equal:", payload('x.Jhash')==payload('y.Jhash')==payload('child.Jhash'))
P
rm -rf $D
