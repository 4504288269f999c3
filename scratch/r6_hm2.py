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
print("keys", len(keys), "both", len(both), "n_pulled", res["n_pulled"], flush=True)
blocks = samples[0]
def run(env):
    for kk in ("RFX_FILTER_NO_PAIR", "RFX_FILTER_GENERIC", "RFX_FILTER_OLD"):
        os.environ.pop(kk, None)
    os.environ.update(env)
    mset = capi.MutantSet(ctx, both, k)
    out = []
    for rep in range(3):
        ms = [m.copy() for m, _ in mset.filter_many(blocks, 1, last_base_skipped=True)]
        nh = [int(n) for _, n in mset.filter_many(blocks, 1, last_base_skipped=True)]
        out.append((ms, nh))
    mset.free()
    return out
ref = None
for name, env in (("pair (default)", {}), ("k_filter_q", {"RFX_FILTER_NO_PAIR": "1"}), ("generic", {"RFX_FILTER_GENERIC": "1"}), ("old", {"RFX_FILTER_OLD": "1"})):
    outs = run(env)
    bits = [sum(int(np.unpackbits(m.view(np.uint8)).sum()) for m in ms) for ms, _ in outs]
    same = all(all(np.array_equal(a, b) for a, b in zip(outs[0][0], o[0])) for o in outs[1:])
    if ref is None and name != "pair (default)":
        ref = outs[0][0]
    vs = None if ref is None else all(np.array_equal(a, b) for a, b in zip(ref, outs[0][0]))
    print(name, "hit reads per run", bits, "n_hit_reads", [sum(nh) for _, nh in outs], "runs identical:", same, "== k_filter_q:", vs, flush=True)
    if name == "pair (default)":
        first = outs
# where do pair-filter runs differ from the reference
diff = [int((np.unpackbits((a ^ b).view(np.uint8))).sum()) for a, b in zip(ref, first[0][0])]
print("bits differing pair vs k_filter_q per block:", diff)
