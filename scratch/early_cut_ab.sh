#!/bin/bash
# A/B of the blocks cut ahead (rfx_count_set_early) on config W, same box: bench.py with and without RFX_BENCH_NO_EARLY=1.
cd "$GRAFT_REPO_ROOT" || exit 1
for e in 1 "" 1 ""; do
  RFX_BENCH_NO_EARLY=$e python bench.py --inner --steps 6 --warmup 3 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; c=d['config']
print('RFX_BENCH_NO_EARLY=%-1s  %.1f M reads/s  %.1f ms per trio  chain %.1f ms (frac %.4f)  k_msp_part1 %.1f ms in %.1f launches per sample  budget %s  peak %.1f GB  mapped %.1f GB  checksums %s' % ('$e', d['value']/1e6, d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['avg_launch_ms_by_kernel']['k_msp_part1'], r['launches_by_kernel_per_chain']['k_msp_part1'], c.get('early_cut_budget_bytes'), c['hbm_peak_bytes']/1e9, c['hbm_mapped_bytes']/1e9, c['checks']['multiset_checksums']))"
done
