#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
PAIRS=32000000; G=1000000000
D=/dev/shm/rfx_cli_scale; mkdir -p $D; O=gpurun_out/cli_trace6; mkdir -p $O
BIN=rufus_amd/bin
$BIN/rfx_synth_fastq $G 0 100 12345 0 $PAIRS $D/reads.fq || exit 1
ls -la $D/reads.fq
for H in 0 0; do
  if [ $H = 1 ]; then export RFX_HOST_PARSE=1; else unset RFX_HOST_PARSE; fi
  s=$(date +%s.%N)
  RFX_CLI_TRACE=1 $BIN/jellyfish count --disk -m 25 -L 2 -s 8G -t 64 -o $D/out.Jhash -C $D/reads.fq 2> $O/trace.$H
  e=$(date +%s.%N)
  python3 -c "print('cli_count host_parse=$H wall=%.2fs rate=%.1f M reads/s' % ($e-$s, 2*$PAIRS/($e-$s)/1e6))"
  cat $O/trace.$H | cut -c1-150
done
rm -rf $D
