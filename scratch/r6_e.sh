#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
B="python bench.py --inner --no-cpu-baseline --no-end-to-end --no-check"
for v in "" fpnopush; do
  L=""; [ -n "$v" ] && L="$GRAFT_REPO_ROOT/scratch/variants/librufus_$v.so"
  RFX_LIB=$L timeout 600 $B --genome 1000000000 --passes 2 --steps 2 --warmup 1 2>gpurun_out/r6e_$v.err | tail -1 | python scratch/r5_summ.py "1g $v" | head -1
done
bash scratch/r6_sq.sh 2>&1 | grep -A1 -E "^k_filter"
