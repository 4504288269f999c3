#!/bin/bash
# Config W through the drop-in executable: one 30x sample of the 3.1 Gb synthetic genome (6.2e8 reads, 200 GB of FASTQ
# in tmpfs) -> jellyfish count (--disk) -> .Jhash -> histo / query.  usage: cli_w_sample.sh [pairs=310000000] [genome=3100000000]
cd "$GRAFT_REPO_ROOT" || exit 1
PAIRS=${1:-310000000}; G=${2:-3100000000}; KK=${K:-25}; SAMPLE=${SAMPLE:-0}
D=/dev/shm/rfx_w; mkdir -p $D; O=gpurun_out/cli_w; mkdir -p $O; BIN=rufus_amd/bin
df -h /dev/shm | tail -1
s=$(date +%s.%N); $BIN/rfx_synth_fastq $G $SAMPLE 1000 12345 0 $PAIRS $D/child.fq || exit 1; e=$(date +%s.%N)
python3 -c "print('generate: %.1f s' % ($e-$s))"; ls -la $D/child.fq
s=$(date +%s.%N)
RFX_CLI_TRACE=1 timeout 900 $BIN/jellyfish count --disk -m $KK -L 2 -s 8G -t 64 -o $D/child.Jhash -C $D/child.fq 2> $O/count.trace; rc=$?
e=$(date +%s.%N)
python3 -c "print('jellyfish count rc=$rc: wall %.1f s = %.1f M reads/s' % ($e-$s, 2*$PAIRS/($e-$s)/1e6))"
cat $O/count.trace
ls -la $D/child.Jhash
if [ -n "$AB" ]; then  # the same count without the pre-populated output mapping (round 3's writer)
  s=$(date +%s.%N); RFX_NO_PREMAP=1 RFX_CLI_TRACE=1 timeout 900 $BIN/jellyfish count --disk -m $KK -L 2 -s 8G -t 64 -o $D/child2.Jhash -C $D/child.fq 2> $O/count_nopremap.trace; rc=$?; e=$(date +%s.%N)
  python3 -c "print('RFX_NO_PREMAP=1 jellyfish count rc=$rc: wall %.1f s' % ($e-$s))"; cat $O/count_nopremap.trace
  python3 -c "
import sys
a, b = open('$D/child.Jhash', 'rb'), open('$D/child2.Jhash', 'rb')
for f in (a, b): f.seek(9 + int(f.read(9)))      # (the headers hold the command lines: the output names differ)
same = True
while same:
    x, y = a.read(1 << 26), b.read(1 << 26)
    same = x == y
    if not x: break
print('payloads identical' if same else 'PAYLOADS DIFFER')"; rm -f $D/child2.Jhash
fi
python3 - <<PY
import os
sz = os.path.getsize("$D/child.Jhash"); blob = open("$D/child.Jhash","rb").read(9); hl = int(blob)
n = (sz - 9 - hl) / ((2 * $KK + 7) // 8 + 4)
print("records in the file: %.0f  (bench.py's library path counts 3244291368 for this sample at the full size)" % n)
PY
s=$(date +%s.%N); RFX_TRACE_LOAD=1 RFX_CLI_TRACE=1 timeout 600 $BIN/jellyfish histo -f -o $D/child.histo $D/child.Jhash 2> $O/histo.trace; e=$(date +%s.%N)
grep "load_fd" $O/histo.trace | head -12; python3 -c "print('jellyfish histo: %.1f s' % ($e-$s))"; head -8 $D/child.histo | tr '\n' ' '; echo
python3 - <<PY
tot = 0; distinct = 0
for ln in open("$D/child.histo"):
    c, n = ln.split(); tot += int(c) * int(n); distinct += int(n)
print("histo: distinct(>=2) %d, instances in them %d" % (distinct, tot))
PY
# a handful of lookups: read 1's first k-mers (present, count ~30) and a poly-A (absent or rare)
sed -n 2p $D/child.fq | cut -c1-$KK > $D/q.txt; sed -n 2p $D/child.fq | cut -c40-$((39+KK)) >> $D/q.txt
s=$(date +%s.%N); timeout 600 $BIN/jellyfish query $D/child.Jhash $(cat $D/q.txt) $(printf "A%.0s" $(seq 1 $KK)); e=$(date +%s.%N)
python3 -c "print('jellyfish query (3 k-mers): %.2f s' % ($e-$s))"
# the hash-list lookup at this size (runRufus.sh:925-926): 8000 k-mers (the first k-mers of reads spread over
# the file) against the 36 GB database -- the records at their positions only, and, for comparison, the walk over the whole file
awk 'NR % 4 == 2' $D/child.fq | head -n 400000 | awk -v k=$KK 'NR % 50 == 0 { print ">" NR "\n" substr($0, 30, k) }' | grep -v N > $D/q8000.fa
grep -c ">" $D/q8000.fa
s=$(date +%s.%N); timeout 600 $BIN/jellyfish query -s $D/q8000.fa $D/child.Jhash > $D/qa.txt; e=$(date +%s.%N)
python3 -c "print('jellyfish query -s (8000 k-mers, records at the queried positions): %.2f s' % ($e-$s))"
s=$(date +%s.%N); RFX_QUERY_NO_SPARSE=1 timeout 900 $BIN/jellyfish query -s $D/q8000.fa $D/child.Jhash > $D/qb.txt; e=$(date +%s.%N)
python3 -c "print('the same with RFX_QUERY_NO_SPARSE=1 (every position range of the file loaded): %.2f s' % ($e-$s))"
cmp $D/qa.txt $D/qb.txt && echo "same lines"; awk '{ s += ($2 > 0) } END { print s " of " NR " present" }' $D/qa.txt
nvidia-smi >/dev/null 2>&1; rocm-smi --showmemuse 2>/dev/null | grep -i "vram" | head -2
rm -rf $D
