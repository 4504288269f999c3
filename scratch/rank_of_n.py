"""What ONE rank of N does per step of the 30x trio (BASELINE configs[3]), measured on one GPU at full size.

Rank g of N holds blocks g, g + N, .. of every sample.  Per sample and step it
  (A) partitions ITS blocks in one pass (k_msp_part1 + k_part2: every bin, no shard pass -- rfx_count_add of a table
      without a shard), then sends the record runs of the other ranks' bins: (N - 1) / N of its records leave, as many arrive;
  (B) counts the bins it owns -- 1 / N of the sample's records, from every rank -- : refinement, leaf, survivor sort
      (rfx_count_finish of a table of shard g of N; here its records are cut from ALL blocks by the run-map route, which
      is not timed: on N devices they arrive over xGMI).
Set difference and filter follow on 1 / N of the data.  Prints a table: ms per sample for A and B, GB sent per rank and
sample, the exchange at 0.3 TB/s per rank (7 links x ~50 GB/s each way, RCCL all-to-all: an assumption, not a measurement),
the predicted step with the exchange hidden behind A (RFX_WGS_OVERLAP) and not.

usage: python scratch/rank_of_n.py [genome] [N ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rufus_amd import capi, wgs

G = int(sys.argv[1]) if len(sys.argv) > 1 else 3_100_000_000
NS = [int(x) for x in sys.argv[2:]] or [2, 4, 8]
K, SIZE, LOWER = 25, 8 << 30, 2
ctx = capi.Context(0)
pairs = G * 30 // 300
sys_ = [capi.Synth.sample(G, w, n_snv=max(20, min(1000, G // 3_000_000)), seed=12345) for w in range(3)]
samples = [wgs.make_sample(ctx, sy, pairs, 1 << 24, 15, want_good=(i == 0), compact=True) for i, sy in enumerate(sys_)]
ctx.sync()
n_blocks = len(samples[0])
print(f"genome {G}: {2 * pairs} reads per sample in {n_blocks} blocks; one GPU emulates rank 0 of N", flush=True)


def timed(fn):
    ctx.sync()
    t0 = time.perf_counter()
    out = fn()
    ctx.sync()
    return out, (time.perf_counter() - t0) * 1e3


rows = []
for N in [NS[0]] + NS:      # (the first round maps the arena: measured again)
    a_ms, b_ms, sent, n_rec_own = [], [], [], []
    for si, blocks in enumerate(samples):
        mine = blocks[0::N]
        # (A) one-pass partition of the rank's own blocks
        t = capi.CountTable(ctx, K, SIZE, True, mode=capi.COUNT_MSP)

        def part():
            for b in mine:
                t.add(b)
        _, ms = timed(part)
        n_local = sum(s[3] for s in t.segments())
        t.free()
        a_ms.append(ms)
        sent.append(n_local * 12 * (N - 1) / N)
        # (B) the owner's share: shard 0 of N over all blocks (records by the run-map route), finish timed alone
        store = capi.RunMaps(ctx, 0)
        t = capi.CountTable(ctx, K, SIZE, True, mode=capi.COUNT_MSP)
        t.set_shard(0, N)
        t.set_runmaps(store)
        for b in blocks:
            t.add(b)
            store.drop(b)
        (rec, h), ms = timed(lambda: t.finish(LOWER, want_histo=True))
        n_rec_own.append(len(rec))
        rec.free()
        t.free()
        store.free()
        b_ms.append(ms)
    ex_ms = [s / 0.3e12 * 1e3 for s in sent]
    step_hidden = sum(max(a, e) + b for a, e, b in zip(a_ms, ex_ms, b_ms))
    step_plain = sum(a + e + b for a, e, b in zip(a_ms, ex_ms, b_ms))
    rows.append((N, a_ms, b_ms, sent, ex_ms, step_hidden, step_plain))
    print(f"N={N}: A (partition of {len(samples[0][0::N])} blocks) {np.mean(a_ms):.0f} ms/sample, B (owner's finish) {np.mean(b_ms):.0f} ms/sample, "
          f"sent {np.mean(sent) / 1e9:.1f} GB/sample/rank = {np.mean(ex_ms):.0f} ms at 0.3 TB/s; "
          f"count part of a step: {step_hidden:.0f} ms (exchange behind the partition) / {step_plain:.0f} ms (not) "
          f"-> {3 * 2 * pairs / step_hidden / 1e3:.0f} / {3 * 2 * pairs / step_plain / 1e3:.0f} M reads/s (+ set difference, filter)",
          flush=True)
