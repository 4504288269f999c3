"""Every way the library can cut one count must leave the SAME record multiset (rfx_records_checksum + record count):
MSP one pass (x3), the other leaf geometry, refined bins, shard passes 2 / 3 / 5 (set_shard: the shards' sums add up),
deferred passes inside the table (set_passes), n tables of a device group (sharded by read block; here n contexts on one
GPU), P2L -- on data far beyond the oracle: a sparse sample (2 M reads of a 1 Gb genome: nearly every k-mer a singleton) and
a dense one (30x of a 30 Mb genome, 60x of a 10 Mb one), k = 25 / 31, -L 1 / 2.
usage: python tests/soak_determinism.py [quick]      (tests/test_scale_gpu.py runs run())"""
import ctypes as C
import os
import sys
import threading
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rufus_amd import capi, wgs

SIZE, MIN_Q = 8 << 30, 15


def add2(a, b):
    return [(x + y) % (1 << 64) for x, y in zip(a, b)]


def one(c, blocks, k, lower, mode=capi.COUNT_MSP, shard=None, passes=None):
    t = capi.CountTable(c, k, SIZE, mode=mode)
    if passes is not None:
        t.set_passes(passes)
    if shard is not None:
        t.set_shard(*shard)
    for b in blocks:
        t.add(b)
    rec = t.finish(lower)
    out = (list(rec.checksum()), len(rec))
    rec.free()
    t.free()
    return out


def group(sy, n_pairs, per_block, k, lower, n_dev, passes):
    ctxs = [capi.Context(0) for _ in range(n_dev)]
    devs = (C.c_int * n_dev)(*([0] * n_dev))
    for c in ctxs:
        assert capi.lib().rfx_ctx_allow_peers(c._h, devs, n_dev) == 0
    peers = capi.lib().rfx_peers_create(n_dev)
    tables, blocks = [], []
    for i, c in enumerate(ctxs):
        bl = wgs.make_sample(c, sy, n_pairs, per_block, MIN_Q, want_good=False, compact=True)
        for j, b in enumerate(bl):
            if j % n_dev != i:
                b.free()
        bl = [b for j, b in enumerate(bl) if j % n_dev == i]
        t = capi.CountTable(c, k, SIZE)
        t.set_passes(passes)
        t.set_peers(peers, i)
        for b in bl:
            t.add(b)
        tables.append(t)
        blocks.append(bl)
    out = [None] * n_dev

    def fin(i):
        out[i] = tables[i].finish(lower)
    th = [threading.Thread(target=fin, args=(i,)) for i in range(n_dev)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join(600)
    assert all(o is not None for o in out)
    cs, n = [0, 0], 0
    for rec, t, bl, c in zip(out, tables, blocks, ctxs):
        cs, n = add2(cs, rec.checksum()), n + len(rec)
        rec.free()
        t.free()
        for b in bl:
            b.free()
        c.close()
    capi.lib().rfx_peers_free(peers)
    return cs, n


def run(ctx, quick=False, say=lambda s: print(s, flush=True)) -> int:
    """-> number of cuts that disagree with the one-pass MSP count (0 = clean)."""
    bad = 0
    datasets = [("sparse: 1 Gb genome, 2 M reads", 1 << 30, 1 << 20, 1 << 18), ("dense: 30 Mb genome at 30x", 30_000_000, 3_000_000, 1 << 19),
                ("deep: 10 Mb genome at 60x", 10_000_000, 2_000_000, 1 << 19)]
    if quick:
        datasets = datasets[1:2]
    for name, G, n_pairs, per_block in datasets:
        sy = capi.Synth.sample(G, 0, n_snv=100, seed=4711)
        blocks = wgs.make_sample(ctx, sy, n_pairs, per_block, MIN_Q, want_good=False, compact=True)
        for k in (25, 31):
            for lower in (1, 2):
                t0 = time.time()
                ref = one(ctx, blocks, k, lower)
                res = {"msp again": one(ctx, blocks, k, lower), "msp once more": one(ctx, blocks, k, lower)}
                for geo in ("0", "1"):
                    os.environ["RFX_MSP_GEO"] = geo
                    res["geo " + geo] = one(ctx, blocks, k, lower)
                del os.environ["RFX_MSP_GEO"]
                os.environ["RFX_LEAF_FORCE_MIXED"] = "1"
                res["recount route"] = one(ctx, blocks, k, lower)
                del os.environ["RFX_LEAF_FORCE_MIXED"]
                for bits in ("17", "21"):
                    os.environ["RFX_MSP_REFINE_BITS"] = bits
                    res["refine " + bits] = one(ctx, blocks, k, lower)
                del os.environ["RFX_MSP_REFINE_BITS"]
                for S in (2, 3, 5):
                    cs, n = [0, 0], 0
                    for sh in range(S):
                        c1, n1 = one(ctx, blocks, k, lower, shard=(sh, S))
                        cs, n = add2(cs, c1), n + n1
                    res[f"{S} shard passes"] = (cs, n)
                res["deferred, 3 passes"] = one(ctx, blocks, k, lower, passes=3)
                for n_dev, passes in ((2, 1), (3, 2)):
                    res[f"group of {n_dev}, {passes} pass(es)"] = group(sy, n_pairs, per_block, k, lower, n_dev, passes)
                if not quick:
                    res["p2l"] = one(ctx, blocks, k, lower, mode=capi.COUNT_P2L)
                wrong = {w: v for w, v in res.items() if v != ref}
                bad += len(wrong)
                say(f"{name}, k={k}, -L {lower}: {ref[1]} records, checksum {ref[0][0]:016x} / {ref[0][1]:016x}; {len(res)} other cuts "
                      f"{'all agree' if not wrong else 'DISAGREE: ' + str(wrong)}  ({time.time() - t0:.1f} s)")
        for b in blocks:
            b.free()
    return bad


if __name__ == "__main__":
    n_bad = run(capi.Context(0), quick=len(sys.argv) > 1)
    print("soak:", "clean" if not n_bad else f"{n_bad} DISAGREEMENTS")
    sys.exit(1 if n_bad else 0)
