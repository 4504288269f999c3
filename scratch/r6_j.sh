#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
D=/dev/shm/rfx_tp; mkdir -p $D
rufus_amd/bin/rfx_synth_fastq 1000000000 0 300 12345 0 16000000 $D/a.fq > /dev/null 2>&1
ls -la $D
cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc
scratch/ubench/text_path $D/a.fq 16
scratch/ubench/text_path $D/a.fq 8 | grep -E "memcpy|pread"
rm -rf $D
