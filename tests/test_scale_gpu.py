"""Scale path on the GPU: the device generator against its host twin, the multi-block / shard-pass trio
driver (rufus_amd/wgs.py) against the oracle at sizes the oracle finishes in seconds, and a >= 1 Gb-genome
trio checked through size-independent properties (BASELINE.json configs[2])."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle
from rufus_amd import capi, tools, wgs
from tests.synth import synth_fastq, synth_flat

pytestmark = pytest.mark.gpu

K, SIZE, LOWER, MIN_COV, MAX_DEPTH, MIN_Q, THRESH = 25, 8 << 30, 2, 5, 1200, 15, 1


@pytest.mark.parametrize("compact", [False, True])
@pytest.mark.parametrize("which,first,n_pairs,want_good", [(0, 0, 3001, True), (1, 12345, 2048, False),
                                                           (0, 7_000_000_000, 513, True)])
def test_device_generator_matches_host_twin(ctx, which, first, n_pairs, want_good, compact):
    """rfx_synth_reads == rfx_pack_reads(rfx_synth_text): codes, both masks, offsets, lengths -- also when the block
    is kept in the compact form (no offsets / lengths, masks only of the reads with an N)."""
    sy = capi.Synth.sample(300_000, which, n_snv=50, seed=4242)
    blk = ctx.synth_reads(sy, first, n_pairs, MIN_Q, want_good, compact)
    dense_bytes = 2 * n_pairs * (40 + 20 + 8 + (20 if want_good else 0)) + 4
    assert (blk.device_bytes < dense_bytes - 2 * n_pairs * 20) if compact else (blk.device_bytes == dense_bytes)
    got = blk.get(want_good)
    seq, qual = sy.text(first, n_pairs)
    s, q, off = synth_flat(seq, qual)
    ref = capi.PackedReads(s, off, q, MIN_Q, capi.PACK_COUNT | capi.PACK_FILTER)
    assert blk.n == 2 * n_pairs and blk.bases == 2 * n_pairs * 150
    assert np.array_equal(got["codes"], ref.codes[:len(got["codes"])])
    assert np.array_equal(got["acgt"], ref.acgt[:len(got["acgt"])])
    if want_good:
        assert np.array_equal(got["good"], ref.good[:len(got["good"])])
    assert np.array_equal(got["word_off"], ref.word_off) and np.array_equal(got["len"], ref.len[:blk.n])
    blk.free()


@pytest.mark.parametrize("n_random,k", [(100, 25), (9000, 25), (40000, 25), (100000, 31), (3000, 12), (5000, 16), (30000, 17),
                                        (20000, 32)])
def test_queue_filter_on_compact_blocks(ctx, monkeypatch, n_random, k):
    """K5 on compact (uniform 150 bp) blocks, every bitmap size / workgroup shape of k_filter_q and -- sets of more than
    4096 keys, k >= 16 -- the pair filter k_filter_p with two and three bits per entry: per-read hit counts
    against the oracle's scan (src/RUFUS.Filter.cpp:196-277), and the round-2 kernels and the generic kernel give the
    same counts.  The set holds EVERY k-mer of some reads, so whole reads are candidates and the per-wave queue
    overflows and drains in the middle of a read."""
    sy = capi.Synth.sample(300_000, 0, n_snv=50, seed=99)
    n_pairs = 5000
    blk = ctx.synth_reads(sy, 0, n_pairs, MIN_Q, True, True)
    seq, qual = sy.text(0, n_pairs)
    reads = [r.tobytes() for r in seq]
    quals = [q.tobytes() for q in qual]
    rng = np.random.default_rng(n_random + k)
    kmers = []
    for i in rng.integers(0, len(reads), 40):
        kmers += [reads[i][j:j + k].decode() for j in range(150 - k + 1)]
    for i in rng.integers(0, len(reads), 400):
        j = int(rng.integers(0, 150 - k))
        kmers.append(reads[i][j:j + k].decode())
    kmers = [x for x in kmers if "N" not in x]
    kmers += ["".join(rng.choice(list("ACGT"), k)) for _ in range(n_random)]
    text = ("\n".join(f"{x} 7" for x in kmers) + "\n").encode()
    fs = oracle.FilterSet(text)
    mset = capi.MutantSet(ctx, capi.hashlist_keys(text, k), k)
    for skipped, single in ((True, False), (False, True)):
        want = np.array([fs.scan(a, b, k, MIN_Q, single_end=single) for a, b in zip(reads, quals)], dtype=np.uint32)
        hits, mask, nh = mset.filter(blk, 2, skipped)
        assert np.array_equal(hits, want), np.flatnonzero(hits != want)[:5]
        bits = tools._mask_bits(mask, len(reads))
        assert np.array_equal(bits, want >= 2) and nh == int(bits.sum()) and nh > 0
        for env in ("RFX_FILTER_OLD", "RFX_FILTER_GENERIC", "RFX_FILTER_NO_PAIR"):
            monkeypatch.setenv(env, "1")
            hits2, _, _ = mset.filter(blk, 2, skipped)
            monkeypatch.delenv(env)
            assert np.array_equal(hits2, want)
        # thresh = 1 and no counts asked for: the pair filter's hits set the mask's bits themselves (no count array)
        _, mask1, nh1 = mset.filter(blk, 1, skipped, want_hits=False)
        bits1 = tools._mask_bits(mask1, len(reads))
        assert np.array_equal(bits1, want >= 1) and nh1 == int(bits1.sum())
    mset.free()
    blk.free()


def _oracle_trio(sys_, n_pairs, k=K):
    """Oracle records / hash list / pulled pairs of a synthetic trio regenerated as text on the host."""
    fq, recs = [], []
    for sy in sys_:
        seq, qual = sy.text(0, n_pairs)
        fq.append((seq, qual))
        recs.append(oracle.count(None, k, SIZE, lower=LOWER, reads=[r.tobytes() for r in seq]))
    hl = oracle.hash_list(recs[0], recs[1:], MIN_COV, MAX_DEPTH)
    seq, qual = fq[0]
    m1 = synth_fastq(seq[0::2], qual[0::2])
    m2 = synth_fastq(seq[1::2], qual[1::2])
    pulled = oracle.FilterSet(hl.encode()).pairs(m1, m2, k, MIN_Q, THRESH) if hl else np.zeros(0, np.int64)
    return recs, hl, pulled


@pytest.mark.parametrize("passes,block_pairs,refine,bins,compact,per_seg",
                         [(1, 1 << 20, None, None, False, False), (3, 7001, None, None, True, False),
                          (2, 9000, "16", None, False, False), (5, 25000, "21", None, False, False),
                          (3, 11000, "19", "32768", True, False), (1, 25000, None, "16384", True, False),
                          (2, 9000, "16", None, False, True), (3, 11000, "19", "32768", True, True)])
def test_trio_in_blocks_and_passes_matches_oracle(ctx, monkeypatch, passes, block_pairs, refine, bins, compact, per_seg):
    """Multi-block samples, shard passes and chunked refinement of the partition give the oracle's records
    (shards interleaved), histogram, hash list and pulled pairs.  bins > 8192 takes the big-block path
    (scatter, separate histogram pass, exact-size segment) that WGS-size blocks use.  per_seg: the refinement
    with one launch per read block (what samples of more than 64 blocks fall back to) instead of one per chunk
    over the slices of all blocks."""
    if per_seg:
        monkeypatch.setenv("RFX_PART3_PER_SEG", "1")
    if refine:
        monkeypatch.setenv("RFX_MSP_REFINE_BITS", refine)
    if bins:
        monkeypatch.setenv("RFX_P2L_BINS", bins)
    n_pairs, G = 25_000, 250_000
    sys_ = [capi.Synth.sample(G, w, n_snv=12, seed=777) for w in range(3)]
    recs_o, hl_o, pulled_o = _oracle_trio(sys_, n_pairs)
    samples = [wgs.make_sample(ctx, sy, n_pairs, block_pairs, MIN_Q, want_good=(i == 0), compact=compact)
               for i, sy in enumerate(sys_)]
    trio = wgs.WgsTrio(ctx, K, SIZE, LOWER, MIN_COV, MAX_DEPTH, THRESH, passes=passes)
    # the bench's route: subject first, the controls struck off its candidates one at a time, nothing kept
    inc = trio.run(samples)
    assert inc["n_records"] == [len(r.keys) for r in recs_o]
    assert tools.keys_to_text(inc["mutant_keys"], K) == [ln.split()[0] for ln in hl_o.splitlines()]
    assert inc["n_pulled"] == len(pulled_o)
    assert all(np.array_equal(inc["histos"][si], oracle.histo(recs_o[si].counts, full=True)[0]) for si in range(3))
    res = trio.run(samples, keep_shard_records=True)
    for si in range(3):
        parts = [r[si].get() for r in res["shard_records"]]
        keys = np.concatenate([p[0] for p in parts])
        counts = np.concatenate([p[1] for p in parts])
        pos = np.concatenate([p[2] for p in parts])
        o = np.lexsort((keys, pos))
        assert np.array_equal(keys[o], recs_o[si].keys) and np.array_equal(counts[o], recs_o[si].counts)
        assert np.array_equal(res["histos"][si], oracle.histo(recs_o[si].counts, full=True)[0])
    for shard in res["shard_records"]:
        for r in shard:
            r.free()
    lines = hl_o.splitlines()
    assert res["n_mutant"] == len(lines) > 0
    assert tools.keys_to_text(res["mutant_keys"], K) == [ln.split()[0] for ln in lines]
    got = np.concatenate([np.flatnonzero(_pairs_of(m, b.n)) + off for m, b, off in
                          zip(res["hit_masks"], samples[0], np.cumsum([0] + [b.n // 2 for b in samples[0]][:-1]))])
    assert np.array_equal(got, pulled_o.astype(np.int64)) and res["n_pulled"] == len(pulled_o)
    for s in samples:
        for b in s:
            b.free()


_COMP = bytes.maketrans(b"ACGT", b"TGCA")


def _pairs_of(mask, n_reads):
    bits = np.unpackbits(mask.view(np.uint8), bitorder="little")[:n_reads].astype(bool)
    return bits[0::2] | bits[1::2]


def _valid_windows(acgt: np.ndarray, n_reads: int, L: int, k: int) -> int:
    """Number of length-k windows made of ACGT only, from the packed masks of fixed-length reads."""
    wpr = (L + 31) // 32
    m = acgt.reshape(n_reads, wpr)
    bits = np.unpackbits(m.view(np.uint8), axis=1, bitorder="little")[:, :L]
    run = np.zeros(n_reads, dtype=np.int32)
    tot = 0
    for j in range(L):
        run = np.where(bits[:, j] == 1, run + 1, 0)
        tot += int((run >= k).sum())
    return tot


def test_wgs_slice_properties(ctx):
    """30x of a 2^30-base genome (2.1e8 reads per sample, 6.4e8 in the trio; RFX_SCALE_GENOME overrides):
    far beyond what the oracle can count, so check what must hold at any size --
      * one block counted alone with lower = 1: sum(count) == number of ACGT-only windows (host-computed);
      * records of every sample strictly (pos,key)-sorted with count >= 2 (checked on the device side by
        rfx_records_load of the drained payload, and on the host for one shard);
      * two shard passes give the same record counts, histograms and hash list as one pass;
      * hash list: >= 97 % of the 25 alt-allele k-mers of every planted SNV, next to nothing else (the odd
        site where five reads share an error), none of it present in either parent;
      * a 40 k-read slice of the same sample matches the oracle bit for bit."""
    G = int(os.environ.get("RFX_SCALE_GENOME", 1 << 30))
    n_pairs = G // 10            # 2 x 150 bp per pair: 30x
    n_snv = 200
    sys_ = [capi.Synth.sample(G, w, n_snv=n_snv, seed=12345) for w in range(3)]
    # -- one block, lower = 1
    blk = ctx.synth_reads(sys_[0], 0, 1 << 20, MIN_Q, True)
    want = _valid_windows(blk.get()["acgt"], blk.n, 150, K)
    t = capi.CountTable(ctx, K, SIZE)
    t.add(blk)
    rec, h = t.finish(1, want_histo=True)
    assert int(sum(int(x) * i for i, x in enumerate(h))) == want
    # .. and the same records whenever it is counted (with lower = 1 every distinct k-mer is a survivor: 2.3e8 of them
    # through the leaf's staging chunks -- round 4: a workgroup could write past its chunk, a handful of garbage keys per run)
    keys, counts, _ = rec.get()
    assert int(counts.sum(dtype=np.uint64)) == want
    rec.free()
    t.free()
    for _ in range(2):
        t = capi.CountTable(ctx, K, SIZE)
        t.add(blk)
        rec = t.finish(1)
        k2, c2, _ = rec.get()
        assert np.array_equal(k2, keys) and np.array_equal(c2, counts)
        rec.free()
        t.free()
        del k2, c2
    del keys, counts
    blk.free()
    # -- the trio, one pass and two passes
    samples = [wgs.make_sample(ctx, sy, n_pairs, 1 << 24, MIN_Q, want_good=(i == 0)) for i, sy in enumerate(sys_)]
    res1 = wgs.WgsTrio(ctx, K, SIZE, LOWER, MIN_COV, MAX_DEPTH, THRESH, passes=1).run(samples, keep_shard_records=True)
    recs = res1["shard_records"][0]
    for r in recs:
        keys, counts, pos = r.get()
        assert np.all((pos[1:] > pos[:-1]) | ((pos[1:] == pos[:-1]) & (keys[1:] > keys[:-1])))
        assert int(counts.min()) >= 2
        del keys, counts, pos
    # mutant k-mers: exactly the alt-allele k-mers of the planted SNVs (minus the few that sampling loses)
    expect = set()
    for p, ref, alt in sys_[0].snvs():
        ctxt = bytearray(sys_[0].genome(p - K + 1, 2 * K - 1))
        assert ctxt[K - 1:K] == ref
        ctxt[K - 1:K] = alt
        for i in range(K):
            km = bytes(ctxt[i:i + K])
            expect.add(min(km, km[::-1].translate(_COMP)))
    got = [x.encode() for x in tools.keys_to_text(res1["mutant_keys"], K)]
    # (at 1e9 sites x 30 reads x 0.16 % per wrong base, a handful of sites see the SAME error five times:
    # real child-only k-mers, ~2e-9 per site and base -- allow for 30 such sites)
    extra = set(got) - expect
    assert len(extra) <= 25 * 30, f"{len(extra)} mutant k-mers are not SNV k-mers"
    assert len(got) == len(set(got)) and len(set(got) & expect) >= 0.97 * 25 * n_snv
    for parent in recs[1:]:
        assert not parent.query(res1["mutant_keys"]).any()
    assert res1["n_pulled"] > 0
    n_rec1, h1, keys1, pulled1 = res1["n_records"], res1["histos"], res1["mutant_keys"], res1["n_pulled"]
    for r in recs:
        r.free()
    del res1
    res2 = wgs.WgsTrio(ctx, K, SIZE, LOWER, MIN_COV, MAX_DEPTH, THRESH, passes=2).run(samples, verify=True)
    assert res2["n_records"] == n_rec1 and res2["n_pulled"] == pulled1
    assert np.array_equal(res2["mutant_keys"], keys1)
    assert all(np.array_equal(a, b) for a, b in zip(res2["histos"], h1))
    # .. and two passes with blocks hashed ONCE for both (WgsTrio.early_budget: room for the second pass's records of
    # about two and a half samples' blocks here): fewer k_msp_part1 launches, the same record multisets
    trio = wgs.WgsTrio(ctx, K, SIZE, LOWER, MIN_COV, MAX_DEPTH, THRESH, passes=2)
    trio.early_budget = 14 << 30
    ctx.prof(True)
    ctx.prof_reset()
    res3 = trio.run(samples, verify=True)
    n_launch = ctx.prof_dict()["k_msp_part1"][1]
    ctx.prof(False)
    n_blocks = sum(len(s) for s in samples)
    assert n_blocks < n_launch < 2 * n_blocks and not trio._early
    assert res3["n_records"] == n_rec1 and res3["n_pulled"] == pulled1 and np.array_equal(res3["mutant_keys"], keys1)
    assert all(np.array_equal(a, b) for a, b in zip(res3["histos"], h1))
    assert res3["verify"]["checksum"] == res2["verify"]["checksum"] and res3["verify"]["bad_order"] == 0
    print(f"k_msp_part1 launches: {n_launch} with blocks cut ahead, {2 * n_blocks} without")
    # .. and when the headroom turns out not to be there (an out-of-memory error with blocks cut ahead): the same two
    # passes once more without them, not a third pass
    os.environ["RFX_WGS_INJECT_OOM"] = "0:early:0"
    try:
        trio = wgs.WgsTrio(ctx, K, SIZE, LOWER, MIN_COV, MAX_DEPTH, THRESH, passes=2)
        trio.early_budget = 14 << 30
        res4 = trio.run(samples)
    finally:
        del os.environ["RFX_WGS_INJECT_OOM"]
    assert trio.passes == 2 and trio.early_budget == 0 and not trio._early
    assert res4["n_records"] == n_rec1 and np.array_equal(res4["mutant_keys"], keys1)
    # .. and two passes with RUN MAPS (WgsTrio.map_budget: a pool for two samples' maps): every block is hashed once, into
    # its map, and both passes cut their records from reads + map (plus a small launch per block and pass over the reads
    # without a map) -- the same record multisets; the third sample's maps take the room of the first's
    trio = wgs.WgsTrio(ctx, K, SIZE, LOWER, MIN_COV, MAX_DEPTH, THRESH, passes=2)
    trio.map_budget = 2 * sum(b.n * 33 + (1 << 20) for b in samples[0])     # (32 B per read + the list of reads without a map)
    ctx.prof(True)
    ctx.prof_reset()
    res5 = trio.run(samples, verify=True)
    prof = ctx.prof_dict()
    ctx.prof(False)
    assert trio.replayed_blocks == 2 * n_blocks and prof["k_msp_replay"][1] == 2 * n_blocks and prof["k_msp_map"][1] == n_blocks
    assert prof.get("k_msp_part1", (0, 0))[1] <= 2 * n_blocks           # (only the small launches over the reads without a map)
    assert res5["n_records"] == n_rec1 and res5["n_pulled"] == pulled1 and np.array_equal(res5["mutant_keys"], keys1)
    assert all(np.array_equal(a, b) for a, b in zip(res5["histos"], h1))
    assert res5["verify"]["checksum"] == res2["verify"]["checksum"] and res5["verify"]["bad_order"] == 0
    res5 = trio.run(samples)                                            # (the pool is kept from run to run)
    assert trio.replayed_blocks == 2 * n_blocks and np.array_equal(res5["mutant_keys"], keys1)
    # half the room: the maps that fit are replayed, the other blocks are hashed twice
    trio.close()
    trio.map_budget //= 2
    res5 = trio.run(samples, verify=True)
    assert 0 < trio.replayed_blocks < 2 * n_blocks and res5["verify"]["checksum"] == res2["verify"]["checksum"]
    # the headroom turns out not to be there: the same two passes without maps
    trio.close()
    os.environ["RFX_WGS_INJECT_OOM"] = "0:maps:0"
    try:
        trio._injected = False
        trio.map_budget = 1 << 30
        res6 = trio.run(samples)
    finally:
        del os.environ["RFX_WGS_INJECT_OOM"]
    assert trio.passes == 2 and trio.map_budget == 0 and trio._store is None and trio.replayed_blocks == 0
    assert res6["n_records"] == n_rec1 and np.array_equal(res6["mutant_keys"], keys1)
    used_before = ctx.mem_stats()["used"]
    trio.close()
    for s in samples:
        for b in s:
            b.free()
    # -- slice parity against the oracle
    seq, _ = sys_[0].text(5_000_000, 20_000)
    sl = [r.tobytes() for r in seq]
    jf = tools.jellyfish_count(ctx, [b"".join(b">r\n" + r + b"\n" for r in sl)], K, SIZE, lower=2)
    assert jf.records.payload() == oracle.count(None, K, SIZE, lower=2, reads=sl).payload()
    jf.records.free()


@pytest.mark.skipif(os.environ.get("RFX_SKIP_FULL") == "1", reason="RFX_SKIP_FULL=1")
@pytest.mark.parametrize("workload", ["wgs", "tn"])
def test_full_size_runs_check_themselves(ctx, workload):
    """BASELINE.json configs[2] (30x trio, 3.1 Gb, 6.2e8 reads per sample, k = 25) and configs[4] (tumor 60x / normal
    30x, k = 31) at FULL size on the one GPU, planned passes, through wgs.self_check -- the same routine bench.py runs
    after its timed region: records of every shard verified on the device (strict (pos,key) order, pos = M * key,
    count >= 2, sum(count) = sum(i * histo[i])), mutant k-mers held by the subject within [MinCov, MaxDepth] and by no
    control and equal to the planted SNVs' k-mers, S + 1 passes == S passes (record counts, histograms, hash list,
    pulled pairs), a sampled block's k-mer instances == the ACGT-only windows of the generator's host text; and a
    40 k-read slice of the subject against the oracle, payload byte for byte."""
    import torch  # (only for the HBM size)
    tn = workload == "tn"
    k = 31 if tn else K
    G = int(os.environ.get("RFX_FULL_GENOME", 3_100_000_000))
    covs = [60, 30] if tn else [30, 30, 30]
    pairs = [G * c // 300 for c in covs]
    n_snv = max(20, min(1000, G // 3_000_000))
    sys_ = [capi.Synth.sample(G, w, n_snv=n_snv, seed=12345) for w in range(len(covs))]
    free0, total = torch.cuda.mem_get_info()
    used_elsewhere = ctx.mem_stats()["mapped"]          # the arena of this session's ctx is reused, not extra
    resident = int(sum(pairs) * 2 * 43.2) + pairs[0] * 2 * 20
    passes = wgs.plan_passes(2 * pairs[0], 150, k, resident + max(0, total - free0 - used_elsewhere), total,
                             n_samples=len(covs), coverage_hint=covs[0], wide=k > 30)
    samples = [wgs.make_sample(ctx, sy, n, 1 << 24, MIN_Q, want_good=(i == 0), compact=True)
               for i, (sy, n) in enumerate(zip(sys_, pairs))]
    try:
        trio = wgs.WgsTrio(ctx, k, SIZE, LOWER, MIN_COV, MAX_DEPTH, THRESH, passes=passes)
        res = trio.run(samples)
        chk = wgs.self_check(ctx, trio, samples, sys_, res, pairs[0], MIN_Q)
        assert chk["order_pos_count_violations"] == 0 and chk["mutant_in_controls"] == 0
        assert chk["passes_compared"] == [trio.passes, trio.passes + 1]
        assert res["n_pulled"] > 0 and res["n_mutant"] >= 0.9 * 25 * n_snv
    finally:
        for s_ in samples:
            for b in s_:
                b.free()
    seq, _ = sys_[0].text(pairs[0] // 2, 20_000)
    sl = [r.tobytes() for r in seq]
    jf = tools.jellyfish_count(ctx, [b"".join(b">r\n" + r + b"\n" for r in sl)], k, SIZE, lower=2)
    assert jf.records.payload() == oracle.count(None, k, SIZE, lower=2, reads=sl).payload()
    jf.records.free()


# ------------------------------------------------------------------------------------------------
# strong scaling of one trio over ranks: two ranks share the one GPU, exchange over gloo
# ------------------------------------------------------------------------------------------------
def _wgs_worker(rank, world, port, q, passes, n_pairs, G, block_pairs, k=K, inject=None):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if inject:
        os.environ["RFX_WGS_INJECT_OOM"] = inject
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        c = capi.Context(0)
        sys_ = [capi.Synth.sample(G, w, n_snv=12, seed=777) for w in range(3)]
        p0, p1 = n_pairs * rank // world, n_pairs * (rank + 1) // world
        samples = [wgs.make_sample(c, sy, p1 - p0, block_pairs, MIN_Q, want_good=(i == 0), first_pair=p0)
                   for i, sy in enumerate(sys_)]
        trio = wgs.WgsTrio(c, k, SIZE, LOWER, MIN_COV, MAX_DEPTH, THRESH, passes=passes, group=dist.group.WORLD)
        # the routine bench.py runs after its timed region, here over the two ranks: verified records, probes of the hash
        # list in every shard of every rank (all-reduced), the sampled block
        first = trio.run(samples)
        passes_after_first = trio.passes
        chk = wgs.self_check(c, trio, samples, sys_, first, p1 - p0, MIN_Q, sample_pairs=3000)
        assert chk["order_pos_count_violations"] == 0 and chk["mutant_in_controls"] == 0 and chk["mutant_in_subject"] > 0
        res = trio.run(samples, keep_shard_records=True)
        recs = [[tuple(a.tolist() for a in shard[si].get()) for shard in res["shard_records"]] for si in range(3)]
        pulled = np.concatenate([np.flatnonzero(_pairs_of(m, b.n)) + off for m, b, off in
                                 zip(res["hit_masks"], samples[0],
                                     np.cumsum([0] + [b.n // 2 for b in samples[0]][:-1]))]) + p0
        q.put((rank, recs, [h.tolist() for h in res["histos"]], res["mutant_keys"].tolist(), res["n_pulled"],
               pulled.tolist(), res["n_records"], passes_after_first, first["mutant_keys"].tolist(), first["n_pulled"]))
        c.close()
    finally:
        dist.destroy_process_group()


def _two_ranks(passes, block_pairs, k, n_pairs, G, inject=None):
    import socket
    import torch.multiprocessing as mp
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_wgs_worker, args=(r, world, port, q, passes, n_pairs, G, block_pairs, k, inject))
             for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return got


@pytest.mark.parametrize("inject", ["1:partition:0", "0:receive:0", "1:finish:0", "0:partition:1"])
def test_two_ranks_agree_on_one_more_pass(inject):
    """A pass that does not fit on ONE rank (injected: rank:stage:shard, RFX_WGS_INJECT_OOM) is given up by both at
    the same checkpoint and started over with one more pass on both -- nobody is left inside a collective (round 2
    could only raise on a group); hash list, histograms and pulled pairs are those of the oracle."""
    n_pairs, G = 12_000, 150_000
    passes = 2 if inject.endswith(":1") else 1
    got = _two_ranks(passes, 4000, K, n_pairs, G, inject)
    sys_ = [capi.Synth.sample(G, w, n_snv=12, seed=777) for w in range(3)]
    recs_o, hl_o, pulled_o = _oracle_trio(sys_, n_pairs, K)
    want = [oracle.jf_encode(ln.split()[0]) for ln in hl_o.splitlines()]
    for g in got:
        assert g[7] == passes + 1                                  # both ranks took the extra pass in the first run
        assert g[8] == want and want and g[9] == len(pulled_o)     # ... and that run's results are right
        for si in range(3):
            assert g[2][si] == oracle.histo(recs_o[si].counts, full=True)[0].tolist()
            assert g[6][si] == len(recs_o[si].keys)


@pytest.mark.parametrize("passes,block_pairs,k", [(1, 1 << 20, K), (2, 5000, K), (2, 6000, 31)])
def test_trio_strong_scaled_over_two_ranks(passes, block_pairs, k):
    """WgsTrio(group=...): each rank holds half of every sample's pairs; per pass the ranks exchange their
    super-k-mer records by minimizer-bin owner (flat cut over passes x ranks) and count complete bins.  The
    union of all (pass, rank) shards is the oracle's record list; histograms, hash list and pulled pairs too."""
    world, n_pairs, G = 2, 20_000, 200_000
    got = _two_ranks(passes, block_pairs, k, n_pairs, G)
    sys_ = [capi.Synth.sample(G, w, n_snv=12, seed=777) for w in range(3)]
    recs_o, hl_o, pulled_o = _oracle_trio(sys_, n_pairs, k)
    for si in range(3):
        parts = [sh for g in got for sh in g[1][si]]
        keys = np.concatenate([np.array(p[0], np.uint64) for p in parts])
        counts = np.concatenate([np.array(p[1], np.uint64) for p in parts])
        pos = np.concatenate([np.array(p[2], np.uint64) for p in parts])
        assert all(len(p[0]) for p in parts)
        o = np.lexsort((keys, pos))
        assert np.array_equal(keys[o], recs_o[si].keys) and np.array_equal(counts[o], recs_o[si].counts)
        assert got[0][2][si] == got[1][2][si] == oracle.histo(recs_o[si].counts, full=True)[0].tolist()
        assert got[0][6][si] == got[1][6][si] == len(recs_o[si].keys)
    want = [oracle.jf_encode(ln.split()[0]) for ln in hl_o.splitlines()]
    assert got[0][3] == got[1][3] == want and want
    assert got[0][4] == got[1][4] == len(pulled_o)
    assert sorted(got[0][5] + got[1][5]) == pulled_o.tolist()


@pytest.mark.parametrize("k", [K, 31, 27])
@pytest.mark.parametrize("passes,surv_frac,refine", [(0, None, None), (1, None, None), (3, None, "17"), (4, "0.0005", None),
                                                      (7, "0.02", "20"), (2, "pool", "16"), (1, "pool", None)])
def test_table_counts_a_sample_in_shard_passes(ctx, monkeypatch, passes, surv_frac, refine, k):
    """rfx_count_set_passes: the adds only remember the blocks, finish runs the shard passes inside the table
    and sorts the survivors of all passes once -- the full (pos,key)-ordered record list, as the drop-in
    `jellyfish count` writes it.  A starved survivor store (RFX_MSP_SURV_FRAC) exercises the regrow-and-redo
    of the first and of later passes; "pool": the leaf's staging pool starts without a spare chunk
    (RFX_LEAF_STAGE_TEST), comes short, and the rerun gets the chunks the device asked for."""
    if surv_frac == "pool":
        monkeypatch.setenv("RFX_LEAF_STAGE_TEST", "1")
    elif surv_frac:
        monkeypatch.setenv("RFX_MSP_SURV_FRAC", surv_frac)
    if refine:
        monkeypatch.setenv("RFX_MSP_REFINE_BITS", refine)
    sy = capi.Synth.sample(300_000, 0, n_snv=10, seed=99)
    n_pairs = 30_000
    seq, _ = sy.text(0, n_pairs)
    ref = oracle.count(None, k, SIZE, lower=LOWER, reads=[r.tobytes() for r in seq])
    blocks = wgs.make_sample(ctx, sy, n_pairs, 8000, MIN_Q, want_good=False)
    t = capi.CountTable(ctx, k, SIZE)
    t.set_passes(passes)
    for b in blocks:
        t.add(b)
    rec, h = t.finish(LOWER, want_histo=True)
    assert rec.payload() == ref.payload()
    assert np.array_equal(h, oracle.histo(ref.counts, full=True)[0])
    n = len(rec)
    cuts = [0, 1, n // 3, n // 3, n - 5, n]
    assert b"".join(rec.payload_range(a, b - a) for a, b in zip(cuts, cuts[1:])) == ref.payload()
    rec.free()
    t.free()
    for b in blocks:
        b.free()


@pytest.mark.parametrize("n_dev,passes,k,surv_frac,replicate",
                         [(2, 1, 25, None, False), (3, 2, 25, "0.002", False), (4, 0, 31, None, False), (8, 1, 27, None, False),
                          (5, 1, 25, None, False), (3, 2, 25, "0.002", True), (4, 0, 31, None, True)])
def test_tables_on_several_devices_make_one_sorted_payload(monkeypatch, n_dev, passes, k, surv_frac, replicate):
    """SURVEY 8(e) / row E-cli at the C-ABI: n tables -- here n contexts on the ONE GPU of the box, the code path of n
    devices (own arena, own stream, copies between contexts).  Table i is given read blocks i, i + n, ... ONLY
    (runRufus.sh:776-797, `north_star`: shard by read block): in every shard pass it partitions its blocks, the owners
    of the pass's minimizer-bin ranges pull their record runs from every table, count complete bins, and the
    survivors change hands by output position (rfx_count_set_peers).  The n finishes, run concurrently, return slice i
    of the oracle's (pos,key)-ordered payload: concatenated, THE payload; the histograms add up to the oracle's; a
    table's k_msp_part1 launches are those of ITS blocks (x passes), not of the sample's.  With five tables and three
    blocks two tables hold no read at all and still take part.  replicate: round 3's fallback scheme
    (RFX_PEERS_REPLICATE=1 -- every table is given every block and keeps minimizer shard i)."""
    import threading
    if surv_frac:
        monkeypatch.setenv("RFX_MSP_SURV_FRAC", surv_frac)
    if replicate:
        monkeypatch.setenv("RFX_PEERS_REPLICATE", "1")
    sy = capi.Synth.sample(200_000, 0, n_snv=10, seed=77)
    n_pairs = 20_000
    seq, _ = sy.text(0, n_pairs)
    ref = oracle.count(None, k, SIZE, lower=LOWER, reads=[r.tobytes() for r in seq])
    ctxs = [capi.Context(0) for _ in range(n_dev)]
    devs = (C.c_int * n_dev)(*([0] * n_dev))
    for c in ctxs:
        assert capi.lib().rfx_ctx_allow_peers(c._h, devs, n_dev) == 0
    peers = capi.lib().rfx_peers_create(n_dev)
    assert peers
    tables, blocks = [], []
    per_block = 7000                                    # pairs per block: three blocks
    n_blocks = -(-n_pairs // per_block)
    for i, c in enumerate(ctxs):
        bl = wgs.make_sample(c, sy, n_pairs, per_block, MIN_Q, want_good=False, compact=(i % 2 == 0))
        assert len(bl) == n_blocks
        if not replicate:                               # table i: blocks i, i + n, ...
            for j, b in enumerate(bl):
                if j % n_dev != i:
                    b.free()
            bl = [b for j, b in enumerate(bl) if j % n_dev == i]
        t = capi.CountTable(c, k, SIZE)
        t.set_passes(passes)
        t.set_peers(peers, i)
        c.prof(True)
        for b in bl:
            t.add(b)
        tables.append(t)
        blocks.append(bl)
    out = [None] * n_dev

    def fin(i):
        out[i] = tables[i].finish(LOWER, want_histo=True)

    th = [threading.Thread(target=fin, args=(i,)) for i in range(n_dev)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join(300)
    assert all(o is not None for o in out)
    assert b"".join(rec.payload() for rec, _ in out) == ref.payload()
    assert sum(len(rec) for rec, _ in out) == len(ref.keys) and sum(1 for rec, _ in out if len(rec)) >= min(n_dev, 2)
    assert np.array_equal(sum(h for _, h in out), oracle.histo(ref.counts, full=True)[0])
    if not replicate:   # a device hashes ITS reads only: one k_msp_part1 launch per own block and pass, none without a block
        launches = [c.prof_dict().get("k_msp_part1", (0.0, 0))[1] for c in ctxs]
        for i, (n_l, bl) in enumerate(zip(launches, blocks)):
            assert (n_l == 0) == (len(bl) == 0) and n_l % max(len(bl), 1) == 0, (i, launches)
        assert len(set(n_l // len(bl) for n_l, bl in zip(launches, blocks) if bl)) == 1, launches   # the same number of passes
    for (rec, _), t, bl, c in zip(out, tables, blocks, ctxs):
        rec.free()
        t.free()
        for b in bl:
            b.free()
        c.close()
    capi.lib().rfx_peers_free(peers)


@pytest.mark.parametrize("k,passes,refine,bins", [(31, 1, None, None), (31, 3, "18", None), (27, 2, None, "32768"),
                                                  (31, 2, "21", "16384")])
def test_tumor_normal_k31_in_passes_matches_oracle(ctx, monkeypatch, k, passes, refine, bins):
    """BASELINE.json configs[4] in miniature: tumor at 60x, ONE control at 30x, k = 31 (wide super-k-mer
    records: 64-bit word + 32-bit plane) -- count in blocks and shard passes, set difference against the single
    control, filter: records, histograms, hash list and pulled pairs are the oracle's."""
    if refine:
        monkeypatch.setenv("RFX_MSP_REFINE_BITS", refine)
    if bins:
        monkeypatch.setenv("RFX_P2L_BINS", bins)
    G = 150_000
    sys_ = [capi.Synth.sample(G, 0, n_snv=10, seed=31), capi.Synth.sample(G, 1, n_snv=10, seed=31)]
    n_pairs = [30_000, 15_000]          # 60x and 30x
    recs_o = []
    for sy, n in zip(sys_, n_pairs):
        seq, _ = sy.text(0, n)
        recs_o.append(oracle.count(None, k, SIZE, lower=LOWER, reads=[r.tobytes() for r in seq]))
    hl_o = oracle.hash_list(recs_o[0], recs_o[1:], MIN_COV, MAX_DEPTH)
    seq, qual = sys_[0].text(0, n_pairs[0])
    pulled_o = oracle.FilterSet(hl_o.encode()).pairs(synth_fastq(seq[0::2], qual[0::2]), synth_fastq(seq[1::2], qual[1::2]),
                                                     k, MIN_Q, THRESH)
    samples = [wgs.make_sample(ctx, sy, n, 9000, MIN_Q, want_good=(i == 0)) for i, (sy, n) in enumerate(zip(sys_, n_pairs))]
    res = wgs.WgsTrio(ctx, k, SIZE, LOWER, MIN_COV, MAX_DEPTH, THRESH, passes=passes).run(samples, keep_shard_records=True)
    for si in range(2):
        parts = [r[si].get() for r in res["shard_records"]]
        keys, counts, pos = (np.concatenate([p[j] for p in parts]) for j in range(3))
        o = np.lexsort((keys, pos))
        assert np.array_equal(keys[o], recs_o[si].keys) and np.array_equal(counts[o], recs_o[si].counts)
        assert np.array_equal(res["histos"][si], oracle.histo(recs_o[si].counts, full=True)[0])
    for shard in res["shard_records"]:
        for r in shard:
            r.free()
    assert tools.keys_to_text(res["mutant_keys"], k) == [ln.split()[0] for ln in hl_o.splitlines()] and res["n_mutant"] > 0
    assert res["n_pulled"] == len(pulled_o) > 0
    for s_ in samples:
        for b in s_:
            b.free()


def test_trio_takes_more_passes_when_a_pass_does_not_fit(monkeypatch):
    """The pass plan is an estimate: a context with a tiny HBM budget makes the first plan fail with
    out-of-memory inside a pass; the driver frees what it holds, adds a pass and starts over -- same hash list."""
    n_pairs, G = 25_000, 250_000
    sys_ = [capi.Synth.sample(G, w, n_snv=12, seed=777) for w in range(3)]
    _, hl_o, pulled_o = _oracle_trio(sys_, n_pairs)
    with capi.Context(0, hbm_budget=48 << 20) as small:
        samples = [wgs.make_sample(small, sy, n_pairs, 9000, MIN_Q, want_good=(i == 0)) for i, sy in enumerate(sys_)]
        trio = wgs.WgsTrio(small, K, SIZE, LOWER, MIN_COV, MAX_DEPTH, THRESH, passes=1)
        res = trio.run(samples)
        assert trio.passes > 1, small.mem_stats()
        assert tools.keys_to_text(res["mutant_keys"], K) == [ln.split()[0] for ln in hl_o.splitlines()]
        assert res["n_pulled"] == len(pulled_o)


def test_records_load_fd_streams_a_file_in_chunks(ctx, tmp_path):
    """A .Jhash payload of three 8 M-record chunks: rfx_records_load_fd (pread ring -> parse at an offset) gives the
    same records as rfx_records_load of the whole payload; a truncated file and records out of order are refused."""
    G = 4_000_000
    sy = capi.Synth.sample(G, 0, n_snv=10, seed=4242)
    blocks = wgs.make_sample(ctx, sy, 600_000, want_good=False)
    t = capi.CountTable(ctx, K, SIZE, mode=capi.COUNT_MSP)
    for b in blocks:
        t.add(b)
    rec = t.finish(1)
    n = len(rec)
    assert n > (16 << 20), n
    lsize, cols = capi.ceil_log2(SIZE), capi.jf_matrix(capi.ceil_log2(SIZE), K)
    for clen in (4, 2):
        payload = rec.payload(clen)
        rl = 7 + clen
        path = str(tmp_path / f"p{clen}.bin")
        with open(path, "wb") as f:
            f.write(b"x" * 1234 + payload)
        fd = os.open(path, os.O_RDONLY)
        try:
            got = capi.Records.load_fd(ctx, K, lsize, cols, fd, 1234, n, clen)
            want = capi.Records.load(ctx, K, lsize, cols, payload, clen)
            for a, b in zip(got.get(), want.get()):
                assert np.array_equal(a, b)
            if clen == 4:
                assert np.array_equal(got.get()[0], rec.get()[0]) and np.array_equal(got.get()[1], rec.get()[1])
            got.free()
            want.free()
            with pytest.raises(capi.RufusError, match="short read"):
                capi.Records.load_fd(ctx, K, lsize, cols, fd, 1234, n + 5, clen)
            with pytest.raises(capi.RufusError, match="order"):
                capi.Records.load_fd(ctx, K, lsize, cols, fd, 1234 + rl * 3 + 1, n - 10, clen)
        finally:
            os.close(fd)
    for x in blocks + [t, rec]:
        x.free()


def test_bench_two_ranks_share_the_device(tmp_path):
    """Readiness for a multi-GPU node: `bench.py --gpus 2 --one-device` runs the N-rank path -- rank
    g holds its share of every sample, partition of block i + 1 under the exchange of block i, records to the bin
    owners through dist.exchange_rows (grouped isend / irecv in pieces), histogram all-reduce, hash list all-gather --
    with a REAL second rank, both on the box's one GPU: over RCCL if it accepts two ranks on one device, else over
    gloo.  Same mutant k-mers, pulled pairs and record counts as the one-rank run of the same genome."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    flags = ["--genome", "60000000", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-end-to-end", "--no-check"]
    one = subprocess.run([sys.executable, "bench.py", "--inner"] + flags, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert one.returncode == 0, one.stderr[-2000:]
    ref = json.loads(one.stdout.decode().strip().splitlines()[-1])
    line, used = None, None
    for backend in ("nccl", "gloo"):
        s_ = socket.socket()
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
        s_.close()
        env = dict(os.environ, RFX_BENCH_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
        try:
            p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                                "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2", "--one-device"] + flags,
                               cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        except subprocess.TimeoutExpired:
            continue
        out = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
        if p.returncode == 0 and out:
            line, used = json.loads(out[-1]), backend
            break
    assert line is not None, "neither RCCL nor gloo ran two ranks on the one device"
    print(f"two ranks on one device over {used}: {line['value'] / 1e6:.0f} M reads/s")
    assert line["n_gpus"] == 2 and used in line["config"]["one_device_dry_run"]
    for key in ("mutant_kmers", "pulled_pairs", "records_per_sample"):
        assert line["config"][key] == ref["config"][key], key
    # what the first real multi-GPU run is checked by: every rank's view of the group and the records it exchanged --
    # half of a rank's super-k-mer records leave for the other rank's bins, as many arrive, and what one sends the other gets
    mg = line["config"]["multi_gpu"]
    assert mg["rccl_world_size"] == 2 and [r["rank"] for r in mg["per_rank"]] == [0, 1]
    assert all(r["rccl_world_size"] == 2 and r["device"] == 0 for r in mg["per_rank"])
    sent = [r["record_bytes_sent_per_step"] for r in mg["per_rank"]]
    got = [r["record_bytes_received_per_step"] for r in mg["per_rank"]]
    assert sent[0] == got[1] and sent[1] == got[0] and min(sent) > 0
    n_reads = line["config"]["reads_counted_per_step"]
    assert 0.35 * 264 * n_reads / 2 < sum(sent) / 2 < 0.65 * 264 * n_reads / 2      # ~22 records of 12 B per read, half of them leave


def test_every_cut_of_a_count_leaves_the_same_records(ctx):
    """tests/soak_determinism.py: one-pass MSP (three times), both leaf geometries, the recount route, refined bins, 2 / 3 / 5
    shard passes, passes deferred into the table, device groups of 2 and 3 (sharded by read block) and P2L must leave the
    same record multiset (rfx_records_checksum + record count) -- on a sparse sample (2 M reads of a 1 Gb genome, 2.3e8
    records with -L 1), a 30x and a 60x one, k = 25 / 31, -L 1 / 2: twelve configurations x 15 cuts, far beyond the
    oracle.  (Round 4's staging-chunk overrun made exactly this differ from run to run.)"""
    from tests import soak_determinism
    lines = []
    bad = soak_determinism.run(ctx, say=lines.append)
    assert bad == 0 and len(lines) == 12, "\n".join(lines)


@pytest.mark.parametrize("k", [25, 31])
def test_a_block_is_hashed_once_for_two_shards(ctx, k):
    """rfx_count_set_early / rfx_count_adopt_early: while the table of shard s adds a big block it cuts the runs of shard
    s + 1 in the same k_msp_part1 launch and keeps them as segments the table of shard s + 1 adopts -- that table is then
    not given the block.  Same record multiset (rfx_records_checksum + record count, against the one-pass count) for
    S = 2 and for shards 1 -> 2 of S = 4; k_msp_part1 launches: one per block instead of two; a block too small for
    the big path and a shard boundary inside a coarse bin (shards 1 | 2 of S = 3) are ordinary adds; an un-adopted early segment is freed
    with its table."""
    sy = capi.Synth.sample(30_000_000, 0, n_snv=50, seed=2718)
    big = wgs.make_sample(ctx, sy, 2_300_000, 2_300_000, MIN_Q, want_good=False, compact=True)       # 5.8e8 windows: the big path
    small = wgs.make_sample(ctx, sy, 200_000, 200_000, MIN_Q, want_good=False, compact=True, first_pair=2_300_000)
    assert len(big) == 1 and len(small) == 1

    def finish(t):
        rec = t.finish(LOWER)
        out = (rec.checksum(), len(rec))
        rec.free()
        t.free()
        return out

    def whole():
        t = capi.CountTable(ctx, k, SIZE, mode=capi.COUNT_MSP)
        t.add(big[0])
        t.add(small[0])
        return finish(t)

    (ref_cs, ref_n) = whole()

    def launches():
        return ctx.prof_dict().get("k_msp_part1", (0.0, 0))[1]

    for S, s in ((2, 0), (4, 1), (3, 0), (3, 1)):
        tabs = []
        for sh in range(S):
            t = capi.CountTable(ctx, k, SIZE, mode=capi.COUNT_MSP)
            t.set_shard(sh, S)
            tabs.append(t)
        with pytest.raises(capi.RufusError):
            tabs[S - 1].set_early()                                   # the last shard has no next one
        ctx.prof(True)
        ctx.prof_reset()
        tabs[s].set_early()
        tabs[s].add(big[0])
        tabs[s].add(small[0])
        went_early = tabs[s].early_segments()                         # (the small block follows a big one: same bins, same path)
        assert went_early == (0 if (S, s) == (3, 1) else 2)           # shards 1 | 2 of 3 meet at 171 / 256: inside a coarse bin
        with pytest.raises(capi.RufusError):
            tabs[(s + 2) % S if S > 2 else s].adopt_early(tabs[s])    # not the next shard
        tabs[s + 1].adopt_early(tabs[s])
        assert tabs[s].early_segments() == 0
        if not went_early:
            tabs[s + 1].add(big[0])
            tabs[s + 1].add(small[0])
        for sh in range(S):
            if sh not in (s, s + 1):
                tabs[sh].add(big[0])
                tabs[sh].add(small[0])
        assert launches() == 2 * S - went_early                       # one launch less: the block was hashed once for two shards
        cs, n = [0, 0], 0
        for t in tabs:
            c1, n1 = finish(t)
            cs, n = [(a + b) % (1 << 64) for a, b in zip(cs, c1)], n + n1
        assert (tuple(cs), n) == (ref_cs, ref_n), (S, s)
    ctx.prof(False)
    # a block too small for the big path is an ordinary add
    t = capi.CountTable(ctx, k, SIZE, mode=capi.COUNT_MSP)
    t.set_shard(0, 2)
    t.set_early()
    t.add(small[0])
    assert t.early_segments() == 0
    t.free()
    # early segments nobody adopts go with their table
    used0 = ctx.mem_stats()["used"]
    t = capi.CountTable(ctx, k, SIZE, mode=capi.COUNT_MSP)
    t.set_shard(0, 2)
    t.set_early()
    t.add(big[0])
    assert t.early_segments() == 1
    t.free()
    ctx.sync()
    assert ctx.mem_stats()["used"] == used0
    for b in big + small:
        b.free()


def _shard_tables(ctx, k, S, blocks, store=None):
    """One table per shard of S over `blocks`: [(checksum, n_records, replayed)] per shard."""
    out = []
    for sh in range(S):
        t = capi.CountTable(ctx, k, SIZE, mode=capi.COUNT_MSP)
        t.set_shard(sh, S)
        if store is not None:
            t.set_runmaps(store)
        for b in blocks:
            t.add(b)
        rec = t.finish(LOWER)
        out.append((tuple(rec.checksum()), len(rec), t.replayed()))
        rec.free()
        t.free()
    return out


def test_run_maps_made_together_are_the_maps_made_one_by_one(ctx):
    """rfx_count_prepare_maps: the maps of all the blocks a table is about to add, behind ONE wait for the device (the WGS
    driver calls it before a sample's first pass) -- the same maps as rfx_count_add makes one by one: same records per
    shard, one hashing launch per block, none at the adds; a second call, a table without a store, blocks that have their
    map: nothing happens; a store without room for all makes the maps that fit."""
    k = 25
    sy = capi.Synth.sample(30_000_000, 0, n_snv=50, seed=2718)
    big = wgs.make_sample(ctx, sy, 2_300_000, 2_300_000, MIN_Q, want_good=False, compact=True)
    small = wgs.make_sample(ctx, sy, 200_000, 200_000, MIN_Q, want_good=False, compact=True, first_pair=2_300_000)
    blocks = big + small
    store = capi.RunMaps(ctx)
    ref = _shard_tables(ctx, k, 2, blocks, store)
    ref_bytes = store.bytes()
    store.free()
    store = capi.RunMaps(ctx)
    ctx.prof(True)
    ctx.prof_reset()
    got = []
    for sh in range(2):
        t = capi.CountTable(ctx, k, SIZE, mode=capi.COUNT_MSP)
        t.set_shard(sh, 2)
        t.prepare_maps(blocks)              # (no store yet: nothing)
        assert store.blocks() == (0 if sh == 0 else 2)
        t.set_runmaps(store)
        t.prepare_maps(blocks)
        assert store.blocks() == 2 and store.bytes() == ref_bytes
        n_map = ctx.prof_dict()["k_msp_map"][1]
        t.prepare_maps(blocks)              # (they are there)
        for b in blocks:
            t.add(b)
        assert ctx.prof_dict()["k_msp_map"][1] == n_map == 2
        rec = t.finish(LOWER)
        got.append((tuple(rec.checksum()), len(rec), t.replayed()))
        rec.free()
        t.free()
    ctx.prof(False)
    assert got == ref and [g[2] for g in got] == [2, 2]
    store.free()
    store = capi.RunMaps(ctx, budget_bytes=small[0].n * 32 + (1 << 20), pooled=True)      # room for the small block's only
    t = capi.CountTable(ctx, k, SIZE, mode=capi.COUNT_MSP)
    t.set_shard(0, 2)
    t.set_runmaps(store)
    t.prepare_maps(blocks[::-1])
    assert store.blocks() == 1
    t.free()
    store.free()
    for b in blocks:
        b.free()


@pytest.mark.parametrize("k,compact", [(25, True), (31, True), (27, False)])
def test_later_shard_passes_replay_the_run_map(ctx, k, compact):
    """rfx_runmaps_* / k_msp_replay: with a store of run maps a big block is hashed once (k_msp_map, by the first shard
    pass that adds it) and EVERY pass cuts its records from reads + map -- the SAME records per shard as the passes
    that hash (rfx_records_checksum + record count per shard, S = 2, 3, 4); one hashing launch per block instead of S
    (plus the small k_msp_part1 launches over the ~1 % of reads whose runs do not fit a map), S replays; a store
    without room means hashing as before; a pooled store serves the same; the store's memory goes back with it."""
    sy = capi.Synth.sample(30_000_000, 0, n_snv=50, seed=2718)
    big = wgs.make_sample(ctx, sy, 2_300_000, 2_300_000, MIN_Q, want_good=False, compact=compact)    # 5.8e8 windows: the big path
    small = wgs.make_sample(ctx, sy, 200_000, 200_000, MIN_Q, want_good=False, compact=compact, first_pair=2_300_000)
    blocks = big + small
    used0 = None
    for S in (2, 3, 4):
        ref = _shard_tables(ctx, k, S, blocks)
        assert all(r[2] == 0 for r in ref)
        if used0 is None:
            used0 = ctx.mem_stats()["used"]     # (the ctx keeps the lookup tables of (k, size) from the first table on)
        store = capi.RunMaps(ctx)
        ctx.prof(True)
        ctx.prof_reset()
        got = _shard_tables(ctx, k, S, blocks, store)
        prof = ctx.prof_dict()
        ctx.prof(False)
        assert [g[:2] for g in got] == [r[:2] for r in ref], (S, k)
        # (the small block follows a big one into the same table: same bins, same path -- it gets a map too)
        assert [g[2] for g in got] == [2] * S
        assert prof["k_msp_replay"][1] == 2 * S and prof["k_msp_map"][1] == 2
        # both blocks hashed once, into their maps; beside every replay one small launch over the reads without a map
        assert prof.get("k_msp_part1", (0, 0))[1] <= 2 * S
        n_all = big[0].n + small[0].n
        assert store.blocks() == 2 and 2 * n_all * 32 > store.bytes() >= n_all * 32
        store.free()
        ctx.sync()
        assert ctx.mem_stats()["used"] == used0
    # no room in the store: every pass hashes; a pool that holds the small block's map only
    store = capi.RunMaps(ctx, budget_bytes=1 << 20)
    got = _shard_tables(ctx, k, 2, blocks, store)
    assert store.blocks() == 0 and [g[2] for g in got] == [0, 0]
    store.free()
    store = capi.RunMaps(ctx, budget_bytes=small[0].n * 32 + (1 << 20), pooled=True)
    got = _shard_tables(ctx, k, 2, blocks, store)
    assert store.blocks() == 1 and [g[2] for g in got] == [1, 1] and [g[:2] for g in got] == [r[:2] for r in _shard_tables(ctx, k, 2, blocks)]
    store.clear()
    assert store.blocks() == 0 and store.bytes() == 0
    store.free()
    ctx.sync()
    assert ctx.mem_stats()["used"] == used0
    # a map dropped before the last pass: that pass makes it anew
    store = capi.RunMaps(ctx)
    t0 = capi.CountTable(ctx, k, SIZE, mode=capi.COUNT_MSP)
    t0.set_shard(0, 3)
    t0.set_runmaps(store)
    t0.add(big[0])
    assert store.blocks() == 1 and t0.replayed() == 1
    t0.free()
    store.drop(big[0])
    assert store.blocks() == 0 and store.bytes() == 0
    store.free()
    for b in blocks:
        b.free()


@pytest.mark.parametrize("k", [25, 31, 23])
def test_run_map_on_ragged_reads_matches_oracle(ctx, monkeypatch, k):
    """The run map on what a sequencer does not make: reads of every length from 0 to 160 (shorter than k, one k-mer, a
    whole number of 8-base phases and not), N runs, IUPAC codes, lower case, homopolymers and short tandem repeats (ties between
    equal minimizers; many short runs: reads whose entries do not fit the map's 27 and go the ordinary way), in one table
    that runs its shard passes itself (rfx_count_set_passes, S = 2 .. 5) -- the oracle's payload, with and without maps;
    a block with a read of more than 160 bases gets no map.  RFX_P2L_BINS forces the path of big blocks."""
    monkeypatch.setenv("RFX_P2L_BINS", "32768")
    rng = np.random.default_rng(1000 + k)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    genome = acgt[rng.integers(0, 4, 60_000)]
    reads = []
    for i in range(40_000):
        L = int(rng.integers(0, 161)) if i % 4 else 150
        p = int(rng.integers(0, len(genome) - 160))
        r = genome[p:p + L].copy()
        if L and rng.random() < 0.03:
            r[rng.integers(0, L, int(rng.integers(1, 4)))] = acgt[rng.integers(0, 4)]    # errors
        if L and rng.random() < 0.15:
            q = int(rng.integers(0, L))
            r[q:q + int(rng.integers(1, 40))] = ord("N") if rng.random() < 0.7 else ord("R")
        s = r.tobytes()
        if rng.random() < 0.05:
            s = s.lower()
        reads.append(s)
    for unit in (b"A", b"AC", b"ACG", b"ACGTT", b"AACCGGTTACGTAGC"):      # repeats: every window ties / many short runs
        for L in (150, 160, 97):
            reads += [(unit * 200)[j:j + L] for j in range(len(unit))] * 3
    ref = oracle.count(None, k, SIZE, lower=LOWER, reads=reads)
    blk = ctx.upload(capi.PackedReads.from_reads(reads))
    for passes in (2, 3, 5):
        for no_map in (False, True):
            if no_map:
                monkeypatch.setenv("RFX_NO_RUNMAP", "1")
            else:
                monkeypatch.delenv("RFX_NO_RUNMAP", raising=False)
            ctx.prof(True)
            ctx.prof_reset()
            t = capi.CountTable(ctx, k, SIZE)
            t.set_passes(passes)
            t.add(blk)
            rec, h = t.finish(LOWER, want_histo=True)
            prof = ctx.prof_dict()
            ctx.prof(False)
            assert rec.payload() == ref.payload(), (passes, no_map)
            assert np.array_equal(h, oracle.histo(ref.counts, full=True)[0])
            assert prof.get("k_msp_replay", (0, 0))[1] == (0 if no_map else passes)
            assert prof.get("k_msp_map", (0, 0))[1] == (0 if no_map else 1)
            rec.free()
            t.free()
    monkeypatch.delenv("RFX_NO_RUNMAP", raising=False)
    blk.free()
    # a read of 161 bases: the block gets no map, the passes hash it
    reads2 = reads[:5000] + [genome[100:261].tobytes()]
    ref2 = oracle.count(None, k, SIZE, lower=LOWER, reads=reads2)
    blk = ctx.upload(capi.PackedReads.from_reads(reads2))
    ctx.prof(True)
    ctx.prof_reset()
    t = capi.CountTable(ctx, k, SIZE)
    t.set_passes(2)
    t.add(blk)
    rec = t.finish(LOWER)
    assert "k_msp_replay" not in ctx.prof_dict() and rec.payload() == ref2.payload()
    ctx.prof(False)
    rec.free()
    t.free()
    blk.free()


def test_a_block_whose_reads_do_not_fit_run_maps_is_hashed_by_every_pass(ctx, monkeypatch):
    """Short tandem repeats cut a read into more super-k-mers than a run map's 27 entries hold (equal m-mer hashes: the
    minimizer moves with every base).  A block of such reads puts more of them on the list of reads without a map than
    the list holds -- the map is dropped and every shard pass hashes the block as before; a block with a few of them keeps
    its map and the listed reads go through the ordinary kernel beside every replay.  The oracle's payload either way."""
    monkeypatch.setenv("RFX_P2L_BINS", "32768")
    rng = np.random.default_rng(77)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    units = [bytes(acgt[rng.integers(0, 4, int(rng.integers(1, 7)))]) for _ in range(400)]
    repeats = [(u * 160)[:150] for u in units for _ in range(50)]                 # 20 000 reads, all of them beyond a map
    genome = acgt[rng.integers(0, 4, 50_000)]
    plain = [genome[p:p + 150].tobytes() for p in rng.integers(0, len(genome) - 150, 30_000)]
    for reads, expect_replay in ((repeats, False), (plain + repeats[:600], True)):
        ref = oracle.count(None, K, SIZE, lower=LOWER, reads=reads)
        blk = ctx.upload(capi.PackedReads.from_reads(reads))
        ctx.prof(True)
        ctx.prof_reset()
        t = capi.CountTable(ctx, K, SIZE)
        t.set_passes(2)
        t.add(blk)
        rec = t.finish(LOWER)
        prof = ctx.prof_dict()
        ctx.prof(False)
        assert rec.payload() == ref.payload()
        assert prof["k_msp_map"][1] == 1                                            # hashed once for its map ..
        if expect_replay:                                                           # .. which holds: two replays, the listed reads beside them
            assert prof["k_msp_replay"][1] == 2 and prof["k_msp_part1"][1] == 2
        else:                                                                       # .. which is dropped: two ordinary passes
            assert "k_msp_replay" not in prof and prof["k_msp_part1"][1] >= 2
        rec.free()
        t.free()
        blk.free()


def test_run_maps_made_ahead_on_the_second_stream_are_the_same_maps(ctx):
    """rfx_count_prefetch_maps (round 6; opt-in in the WGS driver, RFX_MAP_AHEAD=1): the maps of ANOTHER sample's blocks
    queued on the ctx's second stream while a table counts -- later shard passes over those blocks find the maps (no
    hashing launch of their own) and cut the same records; a plain (unpooled) store and a pool without room queue
    nothing; dropping a block whose map is on its way, and freeing the store, wait for the launches."""
    k = 25
    sy = capi.Synth.sample(30_000_000, 0, n_snv=50, seed=31)
    a = wgs.make_sample(ctx, sy, 2_300_000, 2_300_000, MIN_Q, want_good=False, compact=True)
    b = wgs.make_sample(ctx, sy, 2_300_000, 1_150_000, MIN_Q, want_good=False, compact=True, first_pair=2_300_000)
    ref = _shard_tables(ctx, k, 2, b)
    need = sum(x.n * 32 + (x.n // 32 + 4097) * 4 + 1024 for x in a + b)
    store = capi.RunMaps(ctx, need + (1 << 20), pooled=True)
    t = capi.CountTable(ctx, k, SIZE, mode=capi.COUNT_MSP)
    t.set_shard(0, 2)
    t.set_runmaps(store)
    t.prepare_maps(a)
    assert store.blocks() == len(a)
    assert t.prefetch_maps(b) == len(b)          # queued behind nothing: the second stream hashes while ...
    assert t.prefetch_maps(b) == 0               # (asked twice: the first batch is collected, nothing left to queue)
    for x in a:
        t.add(x)                                  # ... the first cuts and partitions sample a
    rec = t.finish(LOWER)
    rec.free()
    t.free()
    assert store.blocks() == len(a) + len(b)
    ctx.prof(True)
    ctx.prof_reset()
    got = _shard_tables(ctx, k, 2, b, store)
    prof = ctx.prof_query_all() if hasattr(ctx, "prof_query_all") else None
    ctx.prof(False)
    assert [(c, n) for c, n, _ in got] == [(c, n) for c, n, _ in ref]
    assert all(r[2] == len(b) for r in got)      # every block of every pass replayed from the maps made ahead
    # a block dropped while its map is on its way; the store freed with maps in flight
    store.clear()
    t2 = capi.CountTable(ctx, k, SIZE, mode=capi.COUNT_MSP)
    t2.set_shard(0, 2)
    t2.set_runmaps(store)
    assert t2.prefetch_maps(a) == len(a)
    store.drop(a[0])
    assert store.blocks() == len(a) - 1
    assert t2.prefetch_maps(b) == len(b)
    t2.free()
    store.free()
    # no pool: nothing is queued (the arena's memory is ordered by the first stream)
    plain = capi.RunMaps(ctx)
    t3 = capi.CountTable(ctx, k, SIZE, mode=capi.COUNT_MSP)
    t3.set_shard(0, 2)
    t3.set_runmaps(plain)
    assert t3.prefetch_maps(b) == 0
    t3.free()
    plain.free()
    for x in a + b:
        x.free()


def test_a_table_gives_its_own_run_maps_back_when_memory_runs_out(ctx, monkeypatch):
    """run maps take memory from a finish that was planned to fit without them.  A table that made the store
    itself (rfx_count_set_passes: what `jellyfish count` uses) and then runs out of memory while a block is partitioned
    from its map (here: injected, RFX_TEST_NOMEM_WITH_MAPS) frees the store and partitions the block -- and every later
    block and pass -- by hashing, as if there had never been a store: the oracle's records, no block replayed."""
    k = 25
    sy = capi.Synth.sample(30_000_000, 0, n_snv=50, seed=77)
    blocks = wgs.make_sample(ctx, sy, 2_300_000, 2_300_000, MIN_Q, want_good=False, compact=True)     # one big block
    ref = _shard_tables(ctx, k, 1, blocks)
    plain = capi.CountTable(ctx, k, SIZE)
    plain.set_passes(2)
    for b in blocks:
        plain.add(b)
    rec = plain.finish(LOWER)
    want = (tuple(rec.checksum()), len(rec))
    assert want == ref[0][:2] and plain.replayed() > 0      # (two passes in the table: the maps are used)
    rec.free()
    plain.free()
    monkeypatch.setenv("RFX_TEST_NOMEM_WITH_MAPS", "1")
    t = capi.CountTable(ctx, k, SIZE)
    t.set_passes(2)
    for b in blocks:
        t.add(b)
    rec = t.finish(LOWER)
    assert (tuple(rec.checksum()), len(rec)) == want and t.replayed() == 0
    rec.free()
    t.free()
    for b in blocks:
        b.free()
