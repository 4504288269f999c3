import sys, os
sys.path.insert(0, ".")
import numpy as np
from rufus_amd import capi, wgs
from rufus_amd.dist import revcomp_keys
G, cov, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ctx = capi.Context(0)
pairs = G * cov // 300
sys_ = [capi.Synth.sample(G, w, n_snv=20, seed=12345) for w in range(3)]
samples = [wgs.make_sample(ctx, sy, pairs, 1 << 24, 15, want_good=(i == 0), compact=True) for i, sy in enumerate(sys_)]
trio = wgs.WgsTrio(ctx, k, 8 << 30, 2, 5, 100000, 1, passes=2)     # (MaxHashDepth out of the way: more keys)
res = trio.run(samples)
keys = np.asarray(res["mutant_keys"], np.uint64)
both = np.concatenate([keys, revcomp_keys(keys, k)])
blocks = samples[0][:2]
pc = lambda m: int(np.unpackbits(m.view(np.uint8)).sum())
print("G", G, "cov", cov, "k", k, "mutant keys", len(keys), "set", len(both), "n_pulled", res["n_pulled"], flush=True)
def variant(env, skipped):
    for kk in ("RFX_FILTER_NO_PAIR", "RFX_FILTER_GENERIC", "RFX_FILTER_OLD"):
        os.environ.pop(kk, None)
    os.environ.update(env)
    mset = capi.MutantSet(ctx, both, k)
    outs = []
    for rep in range(3):
        outs.append([m.copy() for m, _ in mset.filter_many(blocks, 1, last_base_skipped=skipped)])
    h, m, n = mset.filter(blocks[0], 2, skipped, want_hits=True, want_mask=True)     # thresh 2, counts
    mset.free()
    return outs, (h.copy(), m.copy())
for skipped in (True, False):
    ref = None
    for name, env in (("k_filter_q/fast/big", {"RFX_FILTER_NO_PAIR": "1"}), ("default", {}), ("generic", {"RFX_FILTER_GENERIC": "1"}), ("old", {"RFX_FILTER_OLD": "1"})):
        outs, (h, m2) = variant(env, skipped)
        if ref is None:
            ref, refh = outs[0], h
        same = all(all(np.array_equal(a, b) for a, b in zip(ref, o)) for o in outs)
        print("  last_base_skipped", skipped, name, "hit reads", [sum(pc(m) for m in o) for o in outs], "all runs == reference:", same,
              "| counts equal:", bool(np.array_equal(h, refh)), "thresh-2 mask == (counts >= 2):", bool(np.array_equal(np.unpackbits(m2.view(np.uint8), bitorder="little")[:len(h)].astype(bool), h >= 2)), flush=True)
