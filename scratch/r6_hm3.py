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
blocks = samples[0][:1]
def masks(both, env, reps=3, counts=False):
    for kk in ("RFX_FILTER_NO_PAIR", "RFX_FILTER_GENERIC", "RFX_FILTER_OLD", "RFX_FILTER_PAIR_BITS"):
        os.environ.pop(kk, None)
    os.environ.update(env)
    mset = capi.MutantSet(ctx, both, k)
    out = []
    for rep in range(reps):
        if counts:
            r = mset.filter(blocks[0], 1, last_base_skipped=True)
            out.append(np.asarray(r[0] if isinstance(r, tuple) else r).copy())
        else:
            out.append(mset.filter_many(blocks, 1, last_base_skipped=True)[0][0].copy())
    mset.free()
    return out
pc = lambda m: int(np.unpackbits(m.view(np.uint8)).sum())
for n in (4000, 15000, 30000, 60000, 122281):
    sub = keys[:n]
    both = np.concatenate([sub, revcomp_keys(sub, k)])
    ref = masks(both, {"RFX_FILTER_NO_PAIR": "1"}, 1)[0]
    for name, env in (("pair3", {}), ("pair2", {"RFX_FILTER_PAIR_BITS": "2"})):
        ms = masks(both, env)
        extra = [pc(m & ~ref) for m in ms]; missing = [pc(ref & ~m) for m in ms]
        print(n, name, "ref", pc(ref), "runs", [pc(m) for m in ms], "extra", extra, "missing", missing, flush=True)
