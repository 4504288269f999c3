"""K5 alone at the size of config W: the subject's 6.2e8 reads resident, a set of N random keys (+ reverse complements),
HIP-event time of k_filter per pass over all blocks.  usage: filter_bench.py [genome] [n_keys] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rufus_amd import capi, wgs
from rufus_amd.dist import revcomp_keys
G = int(sys.argv[1]) if len(sys.argv) > 1 else 3_100_000_000
NK = int(sys.argv[2]) if len(sys.argv) > 2 else 24567
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ctx = capi.Context(0)
sy = capi.Synth.sample(G, 0, n_snv=1000, seed=12345)
blocks = wgs.make_sample(ctx, sy, G // 10, 1 << 24, 15, want_good=True, compact=True)
ctx.sync()
rng = np.random.default_rng(1)
keys = rng.integers(0, 1 << 50, NK, dtype=np.uint64)
mset = capi.MutantSet(ctx, np.concatenate([keys, revcomp_keys(keys, 25)]), 25)
n_reads = sum(b.n for b in blocks)
for tag in range(REPS):
    ctx.prof_filter(("k_filter",)); ctx.prof(True); ctx.prof_reset()
    t0 = time.perf_counter()
    hits = 0
    for b in blocks:
        _, mask, nh = mset.filter(b, 1, True, want_hits=False)
        hits += nh
    ctx.sync(); dt = time.perf_counter() - t0
    ms = ctx.prof_dict()["k_filter"][0]
    if os.environ.get("FQ_TIMING"):
        import ctypes as C
        buf = (C.c_ulonglong * 8)()
        capi.lib().rfx_debug_fq.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
        capi.lib().rfx_debug_fq(buf, 1)
        v = list(buf)
        print(f"   drain: {v[1]} calls, {v[2]} entries, {v[0] / max(v[1], 1):.0f} memtime ticks per call; kernel {v[3] / max(v[4], 1):.0f} ticks per wave, {v[4]} waves; drain share {v[0] / max(v[3], 1):.3f}")
    print(f"{os.environ.get('TAG','')} reads {n_reads} keys {2*NK}: k_filter {ms:.2f} ms ({61*n_reads/ms/1e9*1e3/8000:.3f} of 8 TB/s), wall {dt*1e3:.0f} ms, hit reads {hits}", flush=True)
