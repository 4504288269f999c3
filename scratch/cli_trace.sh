#!/bin/bash
# phase marks of the drop-in `jellyfish count` + host-only ingest rates on the GPU box
cd "$GRAFT_REPO_ROOT" || exit 1
PAIRS=${1:-32000000}; G=${2:-320000000}
D=/dev/shm/rfx_cli_scale; mkdir -p $D; O=gpurun_out/cli_trace; mkdir -p $O
BIN=rufus_amd/bin
$BIN/rfx_synth_fastq $G 0 100 12345 0 $PAIRS $D/reads.fq || exit 1
g++ -O2 -std=c++17 -pthread -o /tmp/ingest_harness tests/host/ingest_harness.cpp -Lrufus_amd -lrufus_hip -Wl,-rpath,$PWD/rufus_amd
for T in 16 32 64 128; do
  s=$(date +%s.%N); INGEST_MMAP=1 INGEST_NOSUM=1 /tmp/ingest_harness $T 4194304 25165824 4194304 $D/reads.fq > /dev/null; e=$(date +%s.%N)
  python3 -c "print('host-only ingest (mmap) T=$T: %.1f M reads/s' % (2*$PAIRS/($e-$s)/1e6))"
done
for T in 32 64 128; do
  s=$(date +%s.%N)
  RFX_CLI_TRACE=1 RFX_COUNT_DEFER=0 $BIN/jellyfish count --disk -m 25 -L 2 -s 8G -t $T -o $D/out.Jhash -C $D/reads.fq 2> $O/trace.$T
  e=$(date +%s.%N)
  python3 -c "print('cli_count T=$T wall=%.2fs rate=%.1f M reads/s' % ($e-$s, 2*$PAIRS/($e-$s)/1e6))"
  cat $O/trace.$T
done
ls -la $D/out.Jhash
rm -rf $D
