#!/bin/bash
# One sample of config W FIFO-fed into jellyfish count, as scripts/RunJellyForRUFUS.sh:28 feeds it (generator -> named pipe):
# how long from "finished on the device" to "output closed" when the input has no size to guess the output's from.
# usage: cli_w_fifo_sample.sh [pairs=310000000] [genome=3100000000]
cd "$GRAFT_REPO_ROOT" || exit 1
PAIRS=${1:-310000000}; G=${2:-3100000000}
D=/dev/shm/rfx_fifo; rm -rf $D; mkdir -p $D; O=$PWD/gpurun_out/cli_w_fifo; mkdir -p $O; BIN=$PWD/rufus_amd/bin
echo "config W, one sample, FIFO-fed: genome $G, $PAIRS pairs; $(nproc) hardware threads, cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
for mode in ${MODES:-premap nopremap}; do
  mkfifo $D/s.fq
  ($BIN/rfx_synth_fastq $G 0 1000 12345 0 $PAIRS $D/s.fq &)
  s=$(date +%s.%N)
  unset RFX_NO_PREALLOC RFX_NO_PREMAP
  if [ $mode = nopremap ]; then export RFX_NO_PREALLOC=1; fi      # round 3's writer
  if [ $mode = fallocate ]; then export RFX_NO_PREMAP=1; fi       # pages allocated ahead, not mapped ahead
  RFX_CLI_TRACE=1 timeout 1200 $BIN/jellyfish count --disk -m 25 -L 2 -s 8G -t 64 -o $D/s_$mode.Jhash -C $D/s.fq 2> $O/count_$mode.trace; rc=$?
  e=$(date +%s.%N)
  python3 -c "print('$mode: generator | jellyfish count rc=$rc: %.1f s' % ($e-$s))"
  grep "count:\|write:" $O/count_$mode.trace
  ls -la $D/s_$mode.Jhash
  rm $D/s.fq
done
if [ -f $D/s_premap.Jhash -a -f $D/s_nopremap.Jhash ]; then python3 -c "
a, b = open('$D/s_premap.Jhash', 'rb'), open('$D/s_nopremap.Jhash', 'rb')
for f in (a, b): f.seek(9 + int(f.read(9)))
same = True
while same:
    x, y = a.read(1 << 26), b.read(1 << 26)
    same = x == y
    if not x: break
print('payloads identical' if same else 'PAYLOADS DIFFER')"; fi
rm -rf $D
