import sys, os, time
sys.path.insert(0, ".")
import numpy as np
from rufus_amd import capi, wgs
from rufus_amd.dist import revcomp_keys
G, cov, k = 50_000_000, 600, 25
ctx = capi.Context(0)
pairs = G * cov // 300
sys_ = [capi.Synth.sample(G, w, n_snv=20, seed=12345) for w in range(3)]
samples = [wgs.make_sample(ctx, sy, pairs, 1 << 24, 15, want_good=(i == 0), compact=True) for i, sy in enumerate(sys_)]
trio = wgs.WgsTrio(ctx, k, 8 << 30, 2, 5, 100000, 1, passes=2)
res = trio.run(samples)
keys = np.asarray(res["mutant_keys"], np.uint64)
rng = np.random.default_rng(1)
keys = keys[rng.permutation(len(keys))]
blocks = samples[0][:4]
nreads = sum(b.n for b in blocks)
print("mutant keys available", len(keys), "reads filtered per call", nreads, flush=True)
for n in (2000, 24000, 60000, 122000, 131000, 262000, 524000, 1000000, 2262221):
    sub = keys[:n]
    both = np.concatenate([sub, revcomp_keys(sub, k)])
    mset = capi.MutantSet(ctx, both, k)
    mset.filter_many(blocks, 1, last_base_skipped=True)
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(3):
        r = mset.filter_many(blocks, 1, last_base_skipped=True)
    ctx.sync(); dt = (time.perf_counter() - t0) / 3
    hit = sum(int(nh) for _, nh in r)
    print("hash list of %8d k-mers (set %8d): %7.2f ms per %d reads = %6.0f M reads/s = %.3f of the 61 B/read roofline; hit reads %d" % (n, len(both), dt * 1e3, nreads, nreads / dt / 1e6, 61 * nreads / dt / 8e12, hit), flush=True)
    mset.free()
