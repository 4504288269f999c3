#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
D=/dev/shm/rfx_h; mkdir -p $D; BIN=rufus_amd/bin
$BIN/rfx_synth_fastq 320000000 0 100 12345 0 32000000 $D/reads.fq || exit 1
RFX_CLI_TRACE=1 $BIN/jellyfish count --disk -m 25 -L 2 -s 8G -t 16 -o $D/out.Jhash -C $D/reads.fq 2>&1 | grep "finished\|parsed"
ls -la $D/out.Jhash
for i in 1 2 3; do s=$(date +%s.%N); RFX_TRACE_LOAD=1 RFX_CLI_TRACE=1 $BIN/jellyfish histo -f -o $D/h.txt $D/out.Jhash; e=$(date +%s.%N); python3 -c "print('histo wall %.2f' % ($e-$s))"; done
rm -rf $D
