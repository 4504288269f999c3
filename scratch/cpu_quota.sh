#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max)"
D=/dev/shm/rfx_q; mkdir -p $D
rufus_amd/bin/rfx_synth_fastq 320000000 0 100 12345 0 16000000 $D/reads.fq || exit 1
g++ -O2 -std=c++17 -pthread -o /tmp/ingest_harness tests/host/ingest_harness.cpp -Lrufus_amd -lrufus_hip -Wl,-rpath,$PWD/rufus_amd
for T in 8 16 32 64 128; do
  a=$(grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' ')
  s=$(date +%s.%N); INGEST_MMAP=1 INGEST_NOSUM=1 /tmp/ingest_harness $T 4194304 25165824 4194304 $D/reads.fq > /dev/null; e=$(date +%s.%N)
  b=$(grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' ')
  python3 -c "print('mmap  T=$T: %.1f M reads/s' % (32e6/($e-$s)/1e6), '| before:', '$a', '| after:', '$b')"
  s=$(date +%s.%N); INGEST_NOSUM=1 /tmp/ingest_harness $T 4194304 25165824 4194304 $D/reads.fq > /dev/null; e=$(date +%s.%N)
  python3 -c "print('pread T=$T: %.1f M reads/s' % (32e6/($e-$s)/1e6))"
  s=$(date +%s.%N); INGEST_MMAP=1 INGEST_NOSUM=1 numactl --interleave=all /tmp/ingest_harness $T 4194304 25165824 4194304 $D/reads.fq > /dev/null 2>&1; e=$(date +%s.%N)
  python3 -c "print('mmap+numactl T=$T: %.1f M reads/s' % (32e6/($e-$s)/1e6))"
done
which numactl taskset
for T in 32 64; do
  s=$(date +%s.%N); INGEST_MMAP=1 INGEST_NOSUM=1 taskset -c 0-63 /tmp/ingest_harness $T 4194304 25165824 4194304 $D/reads.fq > /dev/null; e=$(date +%s.%N)
  python3 -c "print('mmap taskset node0 cores T=$T: %.1f M reads/s' % (32e6/($e-$s)/1e6))"
done
numastat -m 2>/dev/null | grep -i "shmem\|MemUsed" | head
rm -rf $D
