import sys, os
sys.path.insert(0, ".")
import numpy as np
from rufus_amd import capi, wgs
G, cov = 100_000_000, 30
ctx = capi.Context(0)
pairs = G * cov // 300
n_snv = 33
for k, size, lower, min_cov, max_cov, passes in ((25, 1 << 27, 2, 5, 1200, 2), (25, 1 << 36, 2, 5, 1200, 1), (25, 8 << 30, 1, 5, 1200, 3), (25, 8 << 30, 3, 5, 1200, 2),
                                                  (31, 1 << 30, 2, 5, 1200, 2), (25, 8 << 30, 2, 2, 40, 2), (27, 1 << 40, 2, 5, 1200, 2)):
    sys_ = [capi.Synth.sample(G, w, n_snv=n_snv, seed=12345) for w in range(3)]
    samples = [wgs.make_sample(ctx, sy, pairs, 1 << 24, 15, want_good=(i == 0), compact=True) for i, sy in enumerate(sys_)]
    try:
        trio = wgs.WgsTrio(ctx, k, size, lower, min_cov, max_cov, 1, passes=passes)
        res = trio.run(samples)
        chk = wgs.self_check(ctx, trio, samples, sys_, res, pairs, 15)
        print("k", k, "size 2^%d" % (size.bit_length() - 1), "lower", lower, "MinCov", min_cov, "MaxDepth", max_cov, "passes", passes, "->", "records", res["n_records"], "mutant", res["n_mutant"], "pulled", res["n_pulled"],
              "| self-check passed:", {kk: chk[kk] for kk in ("records_verified", "order_pos_count_violations", "mutant_in_controls", "snv_kmers_found", "snv_kmers_expected", "passes_compared")}, flush=True)
    except AssertionError as e:
        print("k", k, "size", size, "lower", lower, "MinCov", min_cov, "MaxDepth", max_cov, "passes", passes, "-> self-check FAILED:", repr(e)[:300], flush=True)
    except Exception as e:
        print("k", k, "size", size, "lower", lower, "-> ERROR:", repr(e)[:300], flush=True)
    for s_ in samples:
        for b in s_:
            b.free()
