#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python - <<'PY'
import sys
sys.path.insert(0, ".")
from tests.synth import make_trio, fastq_bytes
import oracle
t = make_trio(genome_len=5_000_000, n_pairs=1_000_000, n_snv=20, seed=1)
open("/tmp/m1.fq", "wb").write(fastq_bytes(t["child"], 1)); open("/tmp/m2.fq", "wb").write(fastq_bytes(t["child"], 2))
import numpy as np
g = t["child"].s[0]
with open("/tmp/hl.txt", "w") as f:
    for i in range(0, 500):
        s = g[i * 7].tobytes().decode()[10:35]
        if "N" not in s: f.write(s + " 9\n")
PY
for i in 1 2; do S=$(date +%s.%N); rufus_amd/bin/RUFUS.Filter /tmp/hl.txt /tmp/m1.fq /tmp/m2.fq /tmp/out 25 15 1 8 > /dev/null; E=$(date +%s.%N); python3 -c "print('filter wall', $E-$S)"; done
wc -l /tmp/out.Mutations.Mate1.fastq
