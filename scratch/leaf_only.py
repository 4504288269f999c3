"""k_msp_leaf time of one 1 Gb sample (2 shard passes) for an experiment build whose results may be void: errors are caught,
the HIP-event brackets of the launches that ran are printed.  usage: RFX_LIB=... python scratch/leaf_only.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rufus_amd import capi, wgs
G = 1_000_000_000
ctx = capi.Context(0)
sy = capi.Synth.sample(G, 0, n_snv=100, seed=12345)
blocks = wgs.make_sample(ctx, sy, G * 30 // 300, 1 << 24, 15, want_good=False, compact=True)
for rep in range(2):
    ctx.prof(True); ctx.prof_reset()
    for sh in range(2):
        t = capi.CountTable(ctx, 25, 8 << 30, True, mode=capi.COUNT_MSP)
        t.set_shard(sh, 2)
        try:
            for b in blocks: t.add(b)
            rec, h = t.finish(2, want_histo=True); rec.free()
        except Exception as e:
            print("error (expected for void builds):", str(e)[:100])
        t.free()
    p = ctx.prof_dict(); ctx.prof(False)
    print("rep", rep, {k: (round(v[0], 1), v[1]) for k, v in p.items() if k in ("k_msp_leaf", "k_part3", "k_msp_part1")})
