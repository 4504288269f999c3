#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
B="python bench.py --inner --no-cpu-baseline --no-end-to-end --no-check"
for e in "" 1 "" 1; do
  unset RFX_REPLAY_OLD; [ -n "$e" ] && export RFX_REPLAY_OLD=1
  timeout 600 $B --genome 1000000000 --passes 2 --steps 3 --warmup 2 2>/dev/null | tail -1 | python scratch/r5_summ.py "1g old=$e" | head -2 | cut -c1-160
done
for e in "" 1; do
  unset RFX_REPLAY_OLD; [ -n "$e" ] && export RFX_REPLAY_OLD=1
  timeout 900 $B --steps 3 --warmup 3 2>/dev/null | tail -1 | python scratch/r5_summ.py "W old=$e" | head -2 | cut -c1-160
done
