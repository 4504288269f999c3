#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python - <<'PY'
import sys
sys.path.insert(0, ".")
from tests.synth import make_trio, fastq_bytes
t = make_trio(genome_len=5_000_000, n_pairs=1_000_000, n_snv=20, seed=1)
open("/tmp/c.fq", "wb").write(fastq_bytes(t["child"], 1) + fastq_bytes(t["child"], 2))
PY
for i in 1 2; do S=$(date +%s.%N); rufus_amd/bin/jellyfish count --timing /tmp/t.txt --disk -m 25 -L 2 -s 8G -t 8 -o /tmp/c.Jhash -C /tmp/c.fq; E=$(date +%s.%N); python3 -c "print('count wall', $E-$S)"; cat /tmp/t.txt; done
S=$(date +%s.%N); cat /tmp/c.fq > /dev/null; E=$(date +%s.%N); python3 -c "print('cat wall', $E-$S)"
S=$(date +%s.%N); wc -l /tmp/c.fq; E=$(date +%s.%N); python3 -c "print('wc wall', $E-$S)"
nproc
