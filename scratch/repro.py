import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rufus_amd import capi, wgs
ctx = capi.Context(0)
compact = bool(int(sys.argv[1])) if len(sys.argv) > 1 else True
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n_pairs, G = 25_000, 250_000
sys_ = [capi.Synth.sample(G, w, n_snv=12, seed=777) for w in range(3)]
samples = [wgs.make_sample(ctx, sy, n_pairs, 7001, 15, want_good=(i == 0), compact=compact) for i, sy in enumerate(sys_)]
ctx.sync(); print('generated', [b.device_bytes for b in samples[0]], flush=True)
t = capi.CountTable(ctx, 25, 8 << 30, True, mode=capi.COUNT_MSP)
t.add(samples[1][0]); ctx.sync(); print('added', flush=True)
rec = t.finish(2); print('finished', len(rec), flush=True); rec.free(); t.free()
trio = wgs.WgsTrio(ctx, 25, 8 << 30, 2, 5, 1200, 1, passes=passes)
for it in range(3):
    inc = trio.run(samples)
    print("run", it, inc["n_records"], inc["n_mutant"], inc["n_pulled"], flush=True)
    res = trio.run(samples, keep_shard_records=True)
    print("keep", it, res["n_records"], res["n_mutant"], res["n_pulled"], flush=True)
    for shard in res["shard_records"]:
        for r in shard:
            r.free()
