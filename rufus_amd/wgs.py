"""WGS-scale driver of the trio hot path (BASELINE.json configs[2]-[3]: synthetic 30x WGS trio, k = 25).

A sample is a LIST of packed read blocks resident in HBM (a block holds < 2^32 k-mer windows, i.e. about
32 M reads of 150 bp).  A 30x human-size sample has 7.8e10 k-mer instances = 187 GB of super-k-mer
records -- more than fits next to the reads -- so the trio runs in S minimizer-shard passes
(``rfx_count_set_shard``): pass s counts shard s of every sample (all blocks into one table; the partition
is refined chunk by chunk at finish), takes the set difference on the shard (the shard of a k-mer is the
same in every sample), adds up the histograms and frees the shard's records.  The mutant k-mers of all
shards, put back into (pos,key) order, are the hash list; the subject's blocks are then filtered.

Everything that computes is a C-ABI call into the HIP library; this module only sequences them (the same
sequence the drop-in executables run per sample).
"""
from __future__ import annotations

import os
import time

import numpy as np

from . import capi
from .dist import revcomp_keys

_EVEN = np.uint64(0x5555555555555555)


def pulled_pairs(mask: np.ndarray, n_reads: int) -> int:
    """Number of pairs (reads 2p, 2p+1) with a hit, from the packed per-read hit mask of a block."""
    m = mask[:(n_reads + 63) // 64]
    return int(np.bitwise_count((m | (m >> np.uint64(1))) & _EVEN).sum())


def plan_passes(n_reads_total: int, read_len: int, k: int, resident_bytes: int, hbm_bytes: int, n_samples: int = 3,
                coverage_hint: float = 30.0) -> int:
    """Smallest number of shard passes whose transients fit beside the resident reads.

    Per pass and sample: super-k-mer records (2.4 B per k-mer instance / S), the refinement scratch
    (1/8 of that), the survivor arrays (44 B per surviving k-mer: two partition levels + the records), and
    the records of the samples already counted in this pass (20 B each)."""
    windows = n_reads_total * max(read_len - k + 1, 0)           # per sample
    distinct = windows / max(coverage_hint * (read_len - k + 1) / read_len, 1.0)
    for s in range(1, 257):
        records = 2.5 * windows / s
        transient = records * 1.125 + 44.0 * distinct / s * 1.3 + (n_samples - 1) * 20.0 * distinct / s
        if resident_bytes + transient < 0.92 * hbm_bytes:
            return s
    return 256


class WgsTrio:
    """count x (subject + controls) -> histograms -> hash list -> filter, in minimizer-shard passes."""

    def __init__(self, ctx: capi.Context, k: int, size: int, lower: int, min_cov: int, max_cov: int, thresh: int,
                 passes: int = 1):
        self.ctx, self.k, self.size = ctx, k, size
        self.lsize = capi.ceil_log2(size)
        self.cols = capi.jf_matrix(self.lsize, k)
        self.lower, self.min_cov, self.max_cov, self.thresh = lower, min_cov, max_cov, thresh
        self.passes = passes

    def count_shard(self, blocks, shard: int):
        t = capi.CountTable(self.ctx, self.k, self.size, True, mode=capi.COUNT_MSP)
        try:
            if self.passes > 1:
                t.set_shard(shard, self.passes)
            for b in blocks:
                t.add(b)
            return t.finish(self.lower, want_histo=True)
        finally:
            t.free()

    def pos_of(self, keys: np.ndarray) -> np.ndarray:
        return np.array([capi.jf_pos(self.cols, self.k, self.lsize, int(x)) for x in keys], dtype=np.uint64)

    def run(self, samples, keep_shard_records: bool = False):
        """samples: [subject blocks, control blocks, ...] (lists of capi.ReadBlock)."""
        histos = [np.zeros(capi.HISTO_BINS, dtype=np.uint64) for _ in samples]
        n_rec = [0] * len(samples)
        keys, kept = [], []
        trace = os.environ.get("RFX_WGS_TRACE")
        t_last = time.perf_counter()

        def lap(what):
            nonlocal t_last
            if trace:
                self.ctx.sync()
                now = time.perf_counter()
                print(f"[wgs] {what}: {(now - t_last) * 1e3:.1f} ms", flush=True)
                t_last = now

        for sh in range(self.passes):
            recs = []
            for si, blocks in enumerate(samples):
                rec, h = self.count_shard(blocks, sh)
                recs.append(rec)
                histos[si] += h
                n_rec[si] += len(rec)
                lap(f"pass {sh} sample {si} count ({len(rec)} records)")
            k_, _ = capi.unique_to_subject(self.ctx, recs[0], recs[1:], self.min_cov, self.max_cov)
            lap(f"pass {sh} set difference ({len(k_)} k-mers)")
            keys.append(k_)
            if keep_shard_records:
                kept.append(recs)
            else:
                for r in recs:
                    r.free()
        keys = np.concatenate(keys) if keys else np.zeros(0, np.uint64)
        if len(keys):
            keys = keys[np.lexsort((keys, self.pos_of(keys)))]
        lap("hash list order")
        n_pulled = 0
        masks = []
        if len(keys):
            mset = capi.MutantSet(self.ctx, np.concatenate([keys, revcomp_keys(keys, self.k)]), self.k)
            try:
                for b in samples[0]:
                    _, mask, _ = mset.filter(b, self.thresh, last_base_skipped=True, want_hits=False)
                    n_pulled += pulled_pairs(mask, b.n)
                    masks.append(mask)
            finally:
                mset.free()
        lap("filter")
        out = {"n_mutant": len(keys), "mutant_keys": keys, "n_pulled": n_pulled, "n_records": n_rec, "histos": histos,
               "hit_masks": masks}
        if keep_shard_records:
            out["shard_records"] = kept
        return out


def make_sample(ctx: capi.Context, sy: capi.Synth, n_pairs: int, block_pairs: int = 1 << 24, min_q: int = 15,
                want_good: bool = True, first_pair: int = 0):
    """Blocks of a synthetic sample, generated on the device (pairs first_pair .. first_pair + n_pairs)."""
    blocks, p = [], first_pair
    while p < first_pair + n_pairs:
        n = min(block_pairs, first_pair + n_pairs - p)
        blocks.append(ctx.synth_reads(sy, p, n, min_q, want_good))
        p += n
    return blocks
