#!/bin/bash
# W-sample count A/B in one box: device closed before the process leaves (default) or left to the kernel (RFX_LEAVE_NO_CLOSE=1);
# what the NEXT process (histo) waits for at its first device allocation
cd "$GRAFT_REPO_ROOT" || exit 1
PAIRS=310000000; G=3100000000
D=/dev/shm/rfx_w; mkdir -p $D; BIN=$PWD/rufus_amd/bin
$BIN/rfx_synth_fastq $G 0 1000 12345 0 $PAIRS $D/child.fq || exit 1
t() { local s=$EPOCHREALTIME; "$@"; local e=$EPOCHREALTIME; python3 -c "print('   wall %.2f s' % ($e - $s))"; }
run() {  # $1 = label, rest = env assignments
  echo "=== count: $1"; shift
  t env "$@" RFX_CLI_TRACE=1 $BIN/jellyfish count --disk -m 25 -L 2 -s 8G -t 64 -o $D/child.Jhash -C $D/child.fq 2>&1 | grep "parsed\|finished\|payload out\|closed\|wall"
  echo "--- histo right after it"
  t env RFX_TRACE_LOAD=1 $BIN/jellyfish histo -f -o $D/child.histo $D/child.Jhash 2>&1 | grep "records allocated\|wall" | sed -n '1p;$p'
}
for i in 1 2 3; do
run "default (device closed, then _exit)" X=1
run "RFX_LEAVE_NO_CLOSE=1" RFX_LEAVE_NO_CLOSE=1
done
rm -rf $D
