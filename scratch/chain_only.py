"""Kernel times of ONE sample's count chain (experiment builds whose results may be void).
usage: RFX_LIB=... python scratch/chain_only.py [genome] [passes]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rufus_amd import capi, wgs
G = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = capi.Context(0)
sy = capi.Synth.sample(G, 0, n_snv=100, seed=12345)
blocks = wgs.make_sample(ctx, sy, G * 30 // 300, 1 << 24, 15, want_good=False)
ctx.sync()
for rep in range(2):
    ctx.prof(True); ctx.prof_reset()
    for sh in range(S):
        t = capi.CountTable(ctx, 25, 8 << 30, True, mode=capi.COUNT_MSP)
        if S > 1: t.set_shard(sh, S)
        try:
            for b in blocks: t.add(b)
            rec = t.finish(2, want_histo=True)
            n = len(rec[0]); rec[0].free()
        except Exception as e:
            print("failed:", e); n = -1
        finally:
            t.free()
    d = ctx.prof_dict()
    print(os.environ.get("RFX_LIB", "main").split("librufus_")[-1], "rep", rep, "records", n, "chain %.1f" % sum(v[0] for v in d.values()), {k: round(v[0], 1) for k, v in d.items() if v[0] > 0.7})
