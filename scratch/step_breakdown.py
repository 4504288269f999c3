"""Host-side breakdown of one bench step: wall time of every stage of TrioShard.run, each ending
synchronised, to see where time goes that is not kernel time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from rufus_amd import capi
from rufus_amd.dist import TrioShard
from tests.synth import make_trio, flat_reads

trio = make_trio(genome_len=5_000_000, n_pairs=500_000, n_snv=20, seed=12345, read_seed=1000)
ctx = capi.Context(0)
blocks = {}
for name in ("child", "mother", "father"):
    seq, qual, off = flat_reads(trio[name])
    blocks[name] = ctx.upload(capi.PackedReads(seq, off, qual, bench.MIN_Q, capi.PACK_COUNT | capi.PACK_FILTER))
shard = TrioShard(ctx, bench.K, bench.JF_SIZE, bench.LOWER, bench.MIN_COV, bench.MAX_DEPTH, bench.THRESH, capacity=1 << 26)
be = shard.be
for it in range(4):
    t = [time.perf_counter()]
    pend = [be.count_begin(blocks[n], bench.LOWER) for n in ("child", "mother", "father")]; t.append(time.perf_counter())
    recs = []
    for p in pend:
        rec, h = be.count_end(p); recs.append(rec); t.append(time.perf_counter())
    keys, counts = be.unique(recs[0], recs[1:], bench.MIN_COV, bench.MAX_DEPTH); t.append(time.perf_counter())
    pulled = be.filter_pairs(keys, blocks["child"], bench.THRESH); t.append(time.perf_counter())
    for r in recs: r.free()
    t.append(time.perf_counter())
    d = np.diff(np.array(t)) * 1e3
    print("begin x3 %.3f | end %.3f %.3f %.3f | unique %.3f filter %.3f free %.3f | total %.3f" % (*d[:7], (t[-1] - t[0]) * 1e3))
