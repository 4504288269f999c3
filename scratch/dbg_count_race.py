"""Is the count of one block reproducible within a process whose memory pool has been used before?  (Round 4: it was not --
the leaf's staging chunk, see k_msp_leaf.)   usage: python scratch/dbg_count_race.py [reps=6] [dirty=1] [lower=1]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rufus_amd import capi
from tests.test_scale_gpu import _valid_windows

K, SIZE, MIN_Q = int(os.environ.get("DBG_K", 25)), 8 << 30, 15
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dirty = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lower = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ctx = capi.Context(0)
if dirty:      # leave other records, other survivors behind in the pool
    sy = capi.Synth.sample(300_000, 0, n_snv=50, seed=99)
    for i in range(3):
        b = ctx.synth_reads(sy, i * 7777, 200_000, MIN_Q, True, True)
        t = capi.CountTable(ctx, K, SIZE)
        t.add(b)
        r = t.finish(2)
        r.free(); t.free(); b.free()
s0 = capi.Synth.sample(1 << 30, 0, n_snv=200, seed=12345)
blk = ctx.synth_reads(s0, 0, 1 << 20, MIN_Q, True)
want = _valid_windows(blk.get()["acgt"], blk.n, 150, K)
first = None
for rep in range(reps):
    t = capi.CountTable(ctx, K, SIZE)
    t.add(blk)
    rec, h = t.finish(lower, want_histo=True)
    tot = int(sum(int(x) * i for i, x in enumerate(h)))
    keys, counts, pos = rec.get()
    same = first is None or (np.array_equal(first[0], keys) and np.array_equal(first[1], counts))
    print(f"rep {rep}: records {len(keys)}, sum(histo) - want = {tot - want}, sum(counts) - want = {int(counts.sum(dtype=np.uint64)) - want}, "
          f"same records as rep 0: {same}", flush=True)
    if first is None:
        first = (keys, counts)
    rec.free(); t.free()
