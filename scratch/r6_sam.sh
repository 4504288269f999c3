#!/bin/bash
# the SAM-pipe routes (scripts/RunJellyForRUFUS.sh:28, runRufus.sh:964-967) on the round-6 build, with the
# `cat | cat` ceiling of a pipe on this box beside them.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_sam_pipe.txt
{
echo "# SAM through a pipe on the round-6 build (scratch/r6_sam.sh): jellyfish count --sam and RUFUS.Filter --sam, files and pipes"
bash scratch/cli_sam_scale.sh 32000000 2>&1 | grep -v "^\[rfx"
D=/dev/shm/rfx_sam2; mkdir -p $D
RFX_SYNTH_SAM=1 rufus_amd/bin/rfx_synth_fastq 160000000 0 50 12345 0 16000000 $D/in.sam
s=$(date +%s.%N); cat $D/in.sam | cat > /dev/null; e=$(date +%s.%N)
python3 -c "import os; n=os.path.getsize('$D/in.sam'); print('cat | cat > /dev/null (the pipe itself): %.2f s = %.1f GB/s = %.1f M reads/s of this SAM text' % ($e-$s, n/($e-$s)/1e9, 32e6/($e-$s)/1e6))"
rm -rf $D
bash scratch/feeder_scale.sh 16000000 2>&1 | grep -v "^\[rfx" | head -30
} | tee $O
