import sys, os, json
sys.path.insert(0, ".")
import numpy as np
from rufus_amd import capi, wgs
G = int(sys.argv[1]); cov = int(sys.argv[2]); k = int(sys.argv[3]) if len(sys.argv) > 3 else 25
ctx = capi.Context(0)
pairs = G * cov // 300
n_snv = max(20, min(1000, G // 3_000_000))
sys_ = [capi.Synth.sample(G, w, n_snv=n_snv, seed=12345) for w in range(3)]
samples = [wgs.make_sample(ctx, sy, pairs, 1 << 24, 15, want_good=(i == 0), compact=True) for i, sy in enumerate(sys_)]
out = {}
for passes in (1, 2, 3, 4, 5):
    trio = wgs.WgsTrio(ctx, k, 8 << 30, 2, 5, 1200, 1, passes=passes)
    res = trio.run(samples)
    cs = []
    for attempt in range(1):
        pass
    out[passes] = (res["n_records"], res["n_pulled"], res["n_mutant"], [h.hex() if hasattr(h, "hex") else h for h in res.get("checksums", [])])
    print(passes, out[passes], flush=True)
