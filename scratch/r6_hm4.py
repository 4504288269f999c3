import sys, os
sys.path.insert(0, ".")
import numpy as np
from rufus_amd import capi, wgs
from rufus_amd.dist import revcomp_keys
G, cov, k = 50_000_000, 300, 25
ctx = capi.Context(0)
pairs = G * cov // 300
sys_ = [capi.Synth.sample(G, w, n_snv=20, seed=12345) for w in range(3)]
samples = [wgs.make_sample(ctx, sy, pairs, 1 << 24, 15, want_good=(i == 0), compact=True) for i, sy in enumerate(sys_)]
trio = wgs.WgsTrio(ctx, k, 8 << 30, 2, 5, 1200, 1, passes=2)
res = trio.run(samples)
keys = np.asarray(res["mutant_keys"], np.uint64)
both = np.concatenate([keys, revcomp_keys(keys, k)])
blocks = samples[0][:1]
pc = lambda m: int(np.unpackbits(m.view(np.uint8)).sum())
os.environ["RFX_FILTER_NO_PAIR"] = "1"
mset = capi.MutantSet(ctx, both, k); ref = mset.filter_many(blocks, 1, last_base_skipped=True)[0][0].copy(); mset.free()
os.environ.pop("RFX_FILTER_NO_PAIR")
mset = capi.MutantSet(ctx, both, k)
ex, mi = [], []
for rep in range(6):
    m = mset.filter_many(blocks, 1, last_base_skipped=True)[0][0].copy()
    ex.append(pc(m & ~ref)); mi.append(pc(ref & ~m))
h = []
for rep in range(3):
    hits, m, n = mset.filter(blocks[0], 1, True, want_hits=True, want_mask=True)
    h.append((pc(m & ~ref), pc(ref & ~m)))
print(os.environ.get("RFX_LIB", "default"), "mask-only runs: extra", ex, "missing", mi, "| with counts: (extra, missing)", h, flush=True)
mset.free()
