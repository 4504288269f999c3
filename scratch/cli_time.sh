#!/bin/bash
# wall time of the drop-in `jellyfish count` on a synthetic 2 M-read FASTQ (host parse + pack + upload + GPU)
cd ${GRAFT_REPO_ROOT:-.}
python - <<'PY'
import numpy as np, sys
sys.path.insert(0, ".")
from tests.synth import make_trio, fastq_bytes
t = make_trio(genome_len=5_000_000, n_pairs=1_000_000, n_snv=20, seed=1)
open("/tmp/c1.fq", "wb").write(fastq_bytes(t["child"], 1))
open("/tmp/c2.fq", "wb").write(fastq_bytes(t["child"], 2))
PY
ls -la /tmp/c1.fq /tmp/c2.fq
cat /tmp/c1.fq /tmp/c2.fq > /tmp/c.fq
for i in 1 2; do S=$(date +%s.%N); rufus_amd/bin/jellyfish count --disk -m 25 -L 2 -s 8G -t 8 -o /tmp/c.Jhash -C /tmp/c.fq; E=$(date +%s.%N); echo "count wall $(echo "$E - $S" | bc -l 2>/dev/null || python3 -c "print($E-$S)") s"; done
ls -la /tmp/c.Jhash
S=$(date +%s.%N); rufus_amd/bin/jellyfish histo -f -o /tmp/c.histo /tmp/c.Jhash; E=$(date +%s.%N); python3 -c "print('histo wall', $E-$S)"; head -3 /tmp/c.histo
