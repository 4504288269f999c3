"""GPU parity: the HIP path, called through the C-ABI, against the oracle and the golden fixtures.
Bit-exact everywhere (integer / byte work)."""
import hashlib
import os

import numpy as np
import pytest

import oracle
from rufus_amd import capi, tools
from tests.synth import fastq_bytes, make_trio


def _more_seeds(base):
    """RFX_FUZZ_SEEDS=a-b: the randomised tests also run on seeds a .. b - 1 (campaigns outside the suite: DESIGN.md section 2)."""
    ev = os.environ.get("RFX_FUZZ_SEEDS")
    if not ev:
        return base
    a, b = (int(x) for x in ev.split("-"))
    return base + list(range(a, b))

pytestmark = pytest.mark.gpu


def assert_same_records(jf: tools.JhashFile, orc: oracle.Records):
    keys, counts, pos = jf.records.get()
    assert len(keys) == len(orc.keys)
    assert np.array_equal(keys, orc.keys)
    assert np.array_equal(counts.astype(np.uint64), np.minimum(orc.counts, np.uint64(0xFFFFFFFF)))
    assert np.array_equal(pos, orc.pos)
    assert jf.records.payload() == orc.payload()


# ------------------------------------------------------------------------------------------------
# count + histo + file format
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("label,size", [("s100M", 100_000_000), ("s8G", 8 << 30)])
def test_testrun_count_matches_golden(ctx, testrun, label, size, tmp_path):
    for s in ("Child", "Mother", "Father"):
        e = testrun["expected"]["samples"][s][label]
        out = str(tmp_path / f"{s}.Jhash")
        jf = tools.jellyfish_count(ctx, testrun[s], 25, size, lower=2, out=out, argv=["jellyfish", "count"])
        assert len(jf.records) == e["records"]
        payload = jf.records.payload()
        assert hashlib.sha256(payload).hexdigest() == e["payload_sha256"]
        assert payload[:11].hex() == e["first_record_hex"]
        full = tools.jellyfish_histo(jf, full=True)
        assert full.count("\n") == 10002
        assert hashlib.md5(full.encode()).hexdigest() == e["histo_full_md5"]
        # the file we wrote parses with the oracle's independent reader and reloads on the device
        hdr, pl = oracle.parse_jhash(open(out, "rb").read())
        assert pl == payload and hdr["size"] == 1 << capi.ceil_log2(size) and hdr["canonical"] is True
        back = tools.JhashFile.read(ctx, out)
        assert back.records.payload() == payload
        back.records.free()
        jf.records.free()


@pytest.mark.parametrize("mode", [capi.COUNT_P2L, capi.COUNT_TABLE])
@pytest.mark.parametrize("k,size,canonical,lower", [(25, 1 << 22, True, 0), (31, 8 << 30, True, 2),
                                                     (15, 1 << 16, True, 1), (11, 1 << 10, False, 0),
                                                     (32, 1 << 30, True, 0), (5, 1 << 8, True, 3)])
def test_synthetic_count_matches_oracle(ctx, small_trio, k, size, canonical, lower, mode):
    fq = [fastq_bytes(small_trio["child"], m) for m in (1, 2)]
    jf = tools.jellyfish_count(ctx, fq, k, size, canonical=canonical, lower=lower, mode=mode)
    orc = oracle.count(fq, k, size, lower=lower, canonical=canonical)
    assert_same_records(jf, orc)
    h = jf.records.histo()
    assert np.array_equal(h, oracle.histo(orc.counts, full=True)[0])
    assert tools.jellyfish_histo(jf) == oracle.histo(orc.counts)[1]
    assert tools.jellyfish_dump(jf) == orc.dump_text()
    jf.records.free()


def test_count_edge_cases(ctx):
    k, size = 25, 1 << 20
    reads = [b"", b"ACGT", b"A" * 24, b"A" * 25, b"N" * 40, b"ACGTN" * 30, b"acgtacgtacgtacgtacgtacgtacgtacgt",
             b"TTTTTTTTTTTTTTTTTTTTTTTTTTTTTT", b"ACGTRYACGT" * 20, b"G" * 1000, b"C" * 31 + b"\r" + b"C" * 31,
             (b"ACGGTCAAGTCCATGCAAT" * 40)[:733]]
    fa = b"".join(b">r%d\n%s\n" % (i, r) for i, r in enumerate(reads))
    for mode in (capi.COUNT_P2L, capi.COUNT_TABLE):
        jf = tools.jellyfish_count(ctx, [fa], k, size, mode=mode)
        assert_same_records(jf, oracle.count([fa], k, size))
        jf.records.free()
    # empty input, and an input without a single k-mer
    for data in (b"", b">x\nACGT\n"):
        jf = tools.jellyfish_count(ctx, [data], k, size)
        assert len(jf.records) == 0 and jf.records.payload() == b""
        assert int(jf.records.histo().sum()) == 0
        jf.records.free()
    # upper bound and a counter narrower than the counts (saturating, binary_dumper.hpp:44-48)
    rep = b">x\n" + b"ACGTTGCATGCCGATAGCTAGCTAGGATCCA" * 400 + b"\n"
    jf = tools.jellyfish_count(ctx, [rep], 21, 1 << 16, lower=2, upper=399)
    orc = oracle.count([rep], 21, 1 << 16, lower=2, upper=399)
    assert_same_records(jf, orc)
    jf2 = tools.jellyfish_count(ctx, [rep], 21, 1 << 16)
    orc2 = oracle.count([rep], 21, 1 << 16)
    assert int(orc2.counts.max()) > 255 and jf2.records.payload(1) == orc2.payload(1)
    jf.records.free()
    jf2.records.free()


def test_count_grows_from_a_tiny_table_and_in_blocks(ctx, small_trio):
    """Many blocks into one table that starts far too small: exercises early stop, overflow
    re-insert and rehash; the result must not depend on the capacity or on the block split."""
    child = small_trio["child"]
    k, size = 25, 1 << 27
    seqs = [r.tobytes() for m in (0, 1) for r in child.s[m]]
    ref = oracle.count(None, k, size, lower=2, reads=seqs)
    for cap, nblk in ((1 << 16, 7), (1 << 24, 1)):
        t = capi.CountTable(ctx, k, size, capacity=cap, mode=capi.COUNT_TABLE)
        for part in np.array_split(np.arange(len(seqs)), nblk):
            blk = ctx.upload(capi.PackedReads.from_reads([seqs[i] for i in part]))
            t.add(blk)
            blk.free()
        st = t.stats()
        assert st["distinct"] == len(oracle.count(None, k, size, reads=seqs).keys)
        assert st["distinct"] <= 0.56 * st["capacity"] + 70000
        rec = t.finish(2)
        keys, counts, pos = rec.get()
        assert np.array_equal(keys, ref.keys) and np.array_equal(counts, ref.counts.astype(np.uint32))
        rec.free()
        t.free()


def test_p2l_dense_bins_split_into_rounds_and_many_blocks(ctx):
    """Unrelated random reads: almost every window is a new key, so bins hold more distinct keys than
    the LDS table and k_leaf must split them into ord sub-ranges; several add() calls = several
    instance segments per bin.  Also the mixed case: P2L segments folded into the table path."""
    rng = np.random.default_rng(11)
    seqs = [bytes(r) for r in np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (36000, 150))]]
    seqs += seqs[:3000]                                   # some repeats so that lower=2 keeps something
    k, size = 25, 8 << 30
    ref1 = oracle.count(None, k, size, lower=1, reads=seqs)
    ref2 = oracle.count(None, k, size, lower=2, reads=seqs)
    assert len(ref1.keys) > 4_000_000
    t = capi.CountTable(ctx, k, size, mode=capi.COUNT_P2L)
    for part in np.array_split(np.arange(len(seqs)), 3):
        blk = ctx.upload(capi.PackedReads.from_reads([seqs[i] for i in part]))
        t.add(blk)
        blk.free()
    for lower, ref in ((1, ref1), (2, ref2)):
        rec = t.finish(lower)
        keys, counts, pos = rec.get()
        assert np.array_equal(keys, ref.keys) and np.array_equal(counts, ref.counts.astype(np.uint32))
        assert np.array_equal(pos, ref.pos)
        rec.free()
    # fold into the table path: add the first 1000 reads again as pre-aggregated pairs
    extra = capi.CountTable(ctx, k, size, mode=capi.COUNT_P2L)
    blk = ctx.upload(capi.PackedReads.from_reads(seqs[:1000]))
    extra.add(blk)
    part = extra.finish(1)
    dk, dc, _ = part.dev_ptrs()
    t.add_pairs_dev(dk, dc, len(part))
    ctx.sync()
    rec = t.finish(2)
    ref3 = oracle.count(None, k, size, lower=2, reads=seqs + seqs[:1000])
    assert rec.payload() == ref3.payload()
    for x in (rec, part, blk, extra, t):
        x.free()


@pytest.mark.parametrize("exact", [False, True])
@pytest.mark.parametrize("bins", ["2048", "8192"])
def test_two_level_partition_matches_oracle(ctx, small_trio, bins, exact, monkeypatch):
    """>= 2048 bins switches the P2L path to its two-level partition (coarse bins through LDS-staged
    runs, then fine bins); force it on a small input and compare bit for bit, k=25 and k=31.  Both
    flavours: sizing pass fused into the first partition pass (default) and the exact-size passes."""
    monkeypatch.setenv("RFX_P2L_BINS", bins)
    if exact:
        monkeypatch.setenv("RFX_P2L_EXACT", "1")
    fq = [fastq_bytes(small_trio["father"], m) for m in (1, 2)]
    for k, size, lower in ((25, 8 << 30, 2), (31, 1 << 20, 1)):
        jf = tools.jellyfish_count(ctx, fq, k, size, lower=lower, mode=capi.COUNT_P2L)
        assert_same_records(jf, oracle.count(fq, k, size, lower=lower))
        jf.records.free()
    # ragged reads (lengths 0..400) and a pos-range pass through the same kernels
    rng = np.random.default_rng(5)
    seqs = [bytes(np.frombuffer(b"ACGTN", np.uint8)[rng.choice(5, int(n), p=[.245, .245, .245, .245, .02])])
            for n in rng.integers(0, 400, 3000)]
    t = capi.CountTable(ctx, 25, 1 << 27, mode=capi.COUNT_P2L, pos_lo=1 << 20, pos_hi=100 << 20)
    blk = ctx.upload(capi.PackedReads.from_reads(seqs))
    t.add(blk)
    rec = t.finish(1)
    ref = oracle.count(None, 25, 1 << 27, lower=1, reads=seqs)
    sel = (ref.pos >= (1 << 20)) & (ref.pos < (100 << 20))
    keys, counts, pos = rec.get()
    assert np.array_equal(keys, ref.keys[sel]) and np.array_equal(counts, ref.counts[sel].astype(np.uint32))
    for x in (rec, blk, t):
        x.free()


def test_skewed_input_falls_back_to_the_exact_partition(ctx, monkeypatch):
    """The fused partition pass assumes coarse bins within 25 % of even and < 65536 instances of a
    fine bin per workgroup.  Homopolymer reads break both: the device raises its flag and the block
    is redone on the exact path (visible as a k_bin_count launch); the result is still bit-exact."""
    monkeypatch.setenv("RFX_P2L_BINS", "2048")
    rng = np.random.default_rng(3)
    seqs = [b"A" * 150] * 2500 + [b"AC" * 75] * 700
    seqs += [bytes(r) for r in np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (500, 150))]]
    k, size = 25, 1 << 24
    ref = oracle.count(None, k, size, lower=1, reads=seqs)
    for skew, reads in ((True, seqs), (False, seqs[-500:])):
        ctx.prof(True)
        ctx.prof_reset()
        t = capi.CountTable(ctx, k, size, mode=capi.COUNT_P2L)
        blk = ctx.upload(capi.PackedReads.from_reads(reads))
        t.add(blk)
        rec = t.finish(1)
        launched = ctx.prof_dict()
        ctx.prof(False)
        assert ("k_bin_count" in launched) == skew, launched
        assert "k_part1" in launched and "k_part2" in launched
        if skew:
            keys, counts, pos = rec.get()
            assert np.array_equal(keys, ref.keys) and np.array_equal(counts, ref.counts.astype(np.uint32))
            assert np.array_equal(pos, ref.pos)
            assert int(counts.max()) == 2500 * 126
        for x in (rec, blk, t):
            x.free()


# ------------------------------------------------------------------------------------------------
# MSP path (minimizer super-k-mer partition; default for 23 <= k <= 25)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("k,size,canonical,lower", [(25, 8 << 30, True, 2), (25, 1 << 22, True, 1), (24, 1 << 20, True, 0),
                                                     (23, 1 << 27, False, 1), (25, 1 << 30, False, 3)])
def test_msp_count_matches_oracle(ctx, small_trio, k, size, canonical, lower):
    fq = [fastq_bytes(small_trio["mother"], m) for m in (1, 2)]
    jf = tools.jellyfish_count(ctx, fq, k, size, canonical=canonical, lower=lower, mode=capi.COUNT_MSP)
    orc = oracle.count(fq, k, size, lower=lower, canonical=canonical)
    assert_same_records(jf, orc)
    assert np.array_equal(jf.records.histo(), oracle.histo(orc.counts, full=True)[0])
    jf.records.free()
    # the default mode takes the same path for these k
    ctx.prof(True)
    ctx.prof_reset()
    jf = tools.jellyfish_count(ctx, fq, k, size, canonical=canonical, lower=lower)
    names = ctx.prof_dict()
    ctx.prof(False)
    assert "k_msp_part1" in names and "k_msp_leaf" in names and "k_part1" not in names, names
    assert jf.records.payload() == orc.payload()
    jf.records.free()


def test_auto_mode_falls_back_to_the_table_when_the_budget_is_small(small_trio):
    """rfx_open(device, hbm_budget): the partition buffers of the MSP path do not fit 20 MB, so AUTO counts
    in the open-addressed table (which grows inside the budget) -- same bytes; forcing MSP reports the
    shortage instead of exceeding the budget."""
    small = capi.Context(0, hbm_budget=20 << 20)
    try:
        fq = [fastq_bytes(small_trio["child"], m) for m in (1, 2)]
        reads = [r for f in fq for r in tools.parse_sequences(f)][:4000]
        ref = oracle.count(None, 25, 1 << 24, lower=2, reads=reads)
        blk = small.upload(capi.PackedReads.from_reads(reads))
        small.prof(True)
        small.prof_reset()
        t = capi.CountTable(small, 25, 1 << 24, capacity=1 << 16)
        t.add(blk)
        rec = t.finish(2)
        names = small.prof_dict()
        small.prof(False)
        assert "k_count_reads" in names and "k_msp_leaf" not in names, names
        assert rec.payload() == ref.payload()
        rec.free()
        t.free()
        t = capi.CountTable(small, 25, 1 << 24, mode=capi.COUNT_MSP)
        with pytest.raises(capi.RufusError):
            t.add(blk)
        t.free()
        blk.free()
    finally:
        small.close()


def test_finish_begin_end_pipelines_several_tables(ctx, small_trio):
    """Three tables queued before the first is waited for (MSP: nothing blocks in _begin; the k = 31
    table finishes inside _begin) give the same records and histograms as finish()."""
    fq = {n: [fastq_bytes(small_trio[n], m) for m in (1, 2)] for n in ("child", "mother", "father")}
    cfgs = [("child", 25, 8 << 30, 2), ("mother", 25, 8 << 30, 1), ("father", 31, 1 << 26, 2)]
    tables, blocks, handles = [], [], []
    for n, k, size, lower in cfgs:
        t = capi.CountTable(ctx, k, size)
        blk = ctx.upload(capi.PackedReads.from_reads([r for f in fq[n] for r in tools.parse_sequences(f)]))
        t.add(blk)
        handles.append(t.finish_begin(lower, want_histo=True))
        tables.append(t)
        blocks.append(blk)
    for (n, k, size, lower), t, h in zip(cfgs, tables, handles):
        rec, histo = t.finish_end(h)
        ref = oracle.count(fq[n], k, size, lower=lower)
        assert rec.payload() == ref.payload()
        assert np.array_equal(histo, oracle.histo(ref.counts, full=True)[0])
        rec.free()
    for x in blocks + tables:
        x.free()


def test_msp_rejects_k_outside_its_record_format(ctx):
    t = capi.CountTable(ctx, 22, 1 << 20, mode=capi.COUNT_MSP)
    blk = ctx.upload(capi.PackedReads.from_reads([b"ACGT" * 20]))
    with pytest.raises(capi.RufusError):
        t.add(blk)
    blk.free()
    t.free()


@pytest.mark.parametrize("k,canonical", [(25, True), (28, True), (28, False), (29, False), (31, True)])
def test_msp_low_complexity_records(ctx, k, canonical):
    """Homopolymer stretches after one other base ("G" + "A" * 27 ...) give record words with a single high bit
    set -- the bit patterns an in-band "empty" marker could collide with; random flanks move the minimizer around."""
    rng = np.random.default_rng(k * 2 + canonical)
    reads = []
    for lead in b"CGT":
        for run in range(k - 4, 36):
            for _ in range(12):
                fl = lambda n: bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), n))
                reads.append(fl(int(rng.integers(0, 12))) + bytes([lead]) + b"A" * run + fl(int(rng.integers(0, 12))))
    reads = [r for r in reads if len(r) >= k]
    ref = oracle.count(None, k, 1 << 24, lower=1, reads=reads, canonical=canonical)
    blk = ctx.upload(capi.PackedReads.from_reads(reads))
    t = capi.CountTable(ctx, k, 1 << 24, canonical=canonical, mode=capi.COUNT_MSP)
    t.add(blk)
    rec = t.finish(1)
    assert rec.payload() == ref.payload()
    for x in (rec, t, blk):
        x.free()


@pytest.mark.parametrize("exact", [False, True])
@pytest.mark.parametrize("bins", ["256", "2048", "8192"])
def test_msp_bins_ragged_reads_and_pos_range(ctx, bins, exact, monkeypatch):
    """Every bin count the path supports, fused and exact sizing, reads of length 0..400 with N,
    a pos-range pass, and several add() calls (= several record segments per bin)."""
    monkeypatch.setenv("RFX_P2L_BINS", bins)
    if exact:
        monkeypatch.setenv("RFX_P2L_EXACT", "1")
    rng = np.random.default_rng(int(bins) + exact)
    genome = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 30000)]
    seqs = []
    for n in rng.integers(0, 400, 4000):
        s0 = int(rng.integers(0, len(genome) - 400))
        r = genome[s0:s0 + int(n)].copy()
        r[rng.random(len(r)) < 0.01] = ord("N")
        seqs.append(bytes(r))
    seqs += [b"", b"A" * 24, b"ACGTN" * 30, b"T" * 28, b"T" * 300, b"G" * 25]
    k, size = 25, 1 << 27
    for lo, hi in ((0, 0), (1 << 20, 100 << 20)):
        t = capi.CountTable(ctx, k, size, mode=capi.COUNT_MSP, pos_lo=lo, pos_hi=hi)
        for part in np.array_split(np.arange(len(seqs)), 3):
            blk = ctx.upload(capi.PackedReads.from_reads([seqs[i] for i in part]))
            t.add(blk)
            blk.free()
        for lower in (1, 3):
            rec = t.finish(lower)
            ref = oracle.count(None, k, size, lower=lower, reads=seqs)
            sel = (ref.pos >= lo) & (ref.pos < (hi or 1 << 27))
            keys, counts, pos = rec.get()
            assert np.array_equal(keys, ref.keys[sel]) and np.array_equal(counts, ref.counts[sel].astype(np.uint32))
            assert np.array_equal(pos, ref.pos[sel])
            rec.free()
        t.free()


def test_msp_dense_bins_split_and_survivor_capacity_retry(ctx, monkeypatch):
    """Unrelated random reads: a minimizer bin holds far more distinct k-mers than the LDS table, so
    k_msp_leaf splits it by hash bits; the survivor estimate (60 % of the instances at lower = 1, here
    forced to 0.1 %) is too small, so the emit is rerun with the capacity the device asked for; then
    pre-aggregated pairs are folded in through the table path."""
    monkeypatch.setenv("RFX_P2L_BINS", "256")
    rng = np.random.default_rng(12)
    seqs = [bytes(r) for r in np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (30000, 150))]]
    seqs += seqs[:2500]
    k, size = 25, 8 << 30
    t = capi.CountTable(ctx, k, size, mode=capi.COUNT_MSP)
    blk = ctx.upload(capi.PackedReads.from_reads(seqs))
    t.add(blk)
    for lower, frac in ((1, None), (2, "0.001"), (1, "0.001")):
        if frac:
            monkeypatch.setenv("RFX_MSP_SURV_FRAC", frac)
        ref = oracle.count(None, k, size, lower=lower, reads=seqs)
        rec = t.finish(lower)
        keys, counts, pos = rec.get()
        assert len(keys) == len(ref.keys)
        assert np.array_equal(keys, ref.keys) and np.array_equal(counts, ref.counts.astype(np.uint32))
        assert np.array_equal(pos, ref.pos)
        rec.free()
    monkeypatch.delenv("RFX_MSP_SURV_FRAC")
    extra = capi.CountTable(ctx, k, size, mode=capi.COUNT_MSP)
    blk2 = ctx.upload(capi.PackedReads.from_reads(seqs[:1000]))
    extra.add(blk2)
    part = extra.finish(1)
    dk, dc, _ = part.dev_ptrs()
    t.add_pairs_dev(dk, dc, len(part))
    ctx.sync()
    rec = t.finish(2)
    assert rec.payload() == oracle.count(None, k, size, lower=2, reads=seqs + seqs[:1000]).payload()
    for x in (rec, part, blk, blk2, extra, t):
        x.free()


@pytest.mark.parametrize("force_bits", [None, "15", "17"])
def test_msp_refines_the_partition_when_bins_get_dense(ctx, force_bits, monkeypatch):
    """Four read blocks into one table with 256 bins: > 24 K instances per bin, so finish refines the
    record partition by further minimizer-hash bits (k_bin_hist + a third k_part2 pass over all segments,
    chunk by chunk into a scratch buffer) before the LDS count; forced to 2^17 bins it takes two levels.  A block
    added after a finish joins at the coarse bin count and is refined at the next finish."""
    monkeypatch.setenv("RFX_P2L_BINS", "256")
    if force_bits:
        monkeypatch.setenv("RFX_MSP_REFINE_BITS", force_bits)
    rng = np.random.default_rng(21)
    genome = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 400_000)]
    starts = rng.integers(0, len(genome) - 150, 64_000)
    seqs = [bytes(genome[s0:s0 + 150]) for s0 in starts]
    k, size = 25, 1 << 30
    ctx.prof(True)
    ctx.prof_reset()
    t = capi.CountTable(ctx, k, size, mode=capi.COUNT_MSP)
    parts = np.array_split(np.arange(len(seqs)), 5)
    for part in parts[:4]:
        blk = ctx.upload(capi.PackedReads.from_reads([seqs[i] for i in part]))
        t.add(blk)
        blk.free()
    n4 = sum(len(p) for p in parts[:4])
    rec = t.finish(2)
    names = ctx.prof_dict()
    ctx.prof(False)
    assert "k_bin_hist" in names and "k_part3" in names, names
    assert ("k_part4" in names) == (force_bits == "17"), names
    assert rec.payload() == oracle.count(None, k, size, lower=2, reads=seqs[:n4]).payload()
    rec.free()
    blk = ctx.upload(capi.PackedReads.from_reads([seqs[i] for i in parts[4]]))
    t.add(blk)
    rec = t.finish(1)
    assert rec.payload() == oracle.count(None, k, size, lower=1, reads=seqs).payload()
    for x in (rec, blk, t):
        x.free()


@pytest.mark.parametrize("seed", _more_seeds([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12]))
def test_three_count_paths_agree_on_random_configurations(ctx, seed, monkeypatch):
    """Randomised cross-check: MSP, P2L and the table path must give identical bytes for random k in the
    MSP range, table size, canonical flag, bounds, bin count, block split and read shapes (repeats,
    homopolymers, N, short reads) -- and match the oracle."""
    rng = np.random.default_rng(1000 + seed)
    k = int(rng.choice([23, 24, 25, 25, 26, 27, 28, 29, 30, 31, 31]))
    size = 1 << int(rng.integers(2 * k - 30 if 2 * k > 40 else 10, min(2 * k, 40)))
    canonical = bool(rng.integers(0, 2))
    lower = int(rng.choice([0, 1, 2, 3]))
    upper = [2**64 - 1, 2**64 - 1, 40][int(rng.integers(0, 3))]   # (rng.choice would round 2^64-1 to a float)
    monkeypatch.setenv("RFX_P2L_BINS", str(int(rng.choice([256, 512, 4096, 8192]))))
    if rng.random() < 0.4:     # MSP: force a refinement of the partition (one or two steps)
        monkeypatch.setenv("RFX_MSP_REFINE_BITS", str(int(rng.integers(9, 18))))
    if rng.random() < 0.5:     # MSP: leaf geometry
        monkeypatch.setenv("RFX_MSP_GEO", str(int(rng.integers(0, 2))))
    if rng.random() < 0.2:     # exact two-pass sizing instead of the optimistic one
        monkeypatch.setenv("RFX_P2L_EXACT", "1")
    if seed % 3 == 0:          # MSP leaf: every third bin takes the recount-without-the-record-cache route (what two
        monkeypatch.setenv("RFX_LEAF_FORCE_MIXED", "1")   # different records under one 64-bit cache key would trigger)
    genome = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(rng.integers(2000, 60000)))]
    seqs = []
    for _ in range(int(rng.integers(500, 6000))):
        n = int(rng.choice([0, 10, k - 1, k, k + 1, 60, 150, 151, 300]))
        s0 = int(rng.integers(0, max(1, len(genome) - n)))
        r = genome[s0:s0 + n].copy()
        if rng.random() < 0.1 and len(r):
            r[rng.integers(0, len(r))] = ord("N")
        if rng.random() < 0.03:
            r[:] = ord("ACGT"[int(rng.integers(0, 4))])
        if rng.random() < 0.3:
            r = r[::-1].copy()
        seqs.append(bytes(r))
    splits = sorted(set([0, len(seqs)] + [int(x) for x in rng.integers(0, len(seqs), int(rng.integers(0, 3)))]))
    ref = oracle.count(None, k, size, lower=lower, upper=upper, canonical=canonical, reads=seqs)
    for mode in (capi.COUNT_MSP, capi.COUNT_P2L, capi.COUNT_TABLE, capi.COUNT_AUTO):
        t = capi.CountTable(ctx, k, size, canonical, mode=mode)
        for a, b in zip(splits, splits[1:]):
            blk = ctx.upload(capi.PackedReads.from_reads(seqs[a:b]))
            t.add(blk)
            blk.free()
        rec, h = t.finish(lower, upper, want_histo=True)
        assert rec.payload() == ref.payload(), (mode, k, size, canonical, lower, upper)
        assert np.array_equal(h, oracle.histo(ref.counts, full=True)[0])
        assert rec.checksum() == ref.checksum()
        rec.free()
        t.free()


@pytest.mark.parametrize("n_shards", [2, 3, 7])
def test_msp_shard_passes_partition_the_count(ctx, small_trio, n_shards):
    """rfx_count_set_shard: every pass counts only the k-mers of its minimizer bins; the passes are
    disjoint, interleave to the full count, and cut the key space like the multi-GPU owner ranges."""
    from rufus_amd import dist as rdist
    k, size = 25, 1 << 28
    reads = [r for m in (1, 2) for r in tools.parse_sequences(fastq_bytes(small_trio["child"], m))]
    ref = oracle.count(None, k, size, lower=2, reads=reads)
    blk = ctx.upload(capi.PackedReads.from_reads(reads))
    shards, hsum = [], np.zeros(capi.HISTO_BINS, dtype=np.uint64)
    cs = [0, 0]
    for sh in range(n_shards):
        t = capi.CountTable(ctx, k, size)
        t.set_shard(sh, n_shards)
        t.add(blk)
        (d_rec, d_bs, nb, nrec), = t.segments()
        assert nb >= 256 and nrec > 0
        rec, h = t.finish(2, want_histo=True)
        shards.append(tuple(x.astype(np.uint64) for x in rec.get()))
        cs = [(a + b) % (1 << 64) for a, b in zip(cs, rec.checksum())]
        hsum += h
        rec.free()
        t.free()
    assert all(len(s_[0]) for s_ in shards)
    keys, counts, pos = rdist.merge_shards(shards)
    assert np.array_equal(keys, ref.keys) and np.array_equal(counts, ref.counts) and np.array_equal(pos, ref.pos)
    assert sum(len(s_[0]) for s_ in shards) == len(ref.keys)          # disjoint
    assert tuple(cs) == ref.checksum()      # rfx_records_checksum: the shards' sums add up to the whole multiset's
    assert np.array_equal(hsum, oracle.histo(ref.counts, full=True)[0])
    t = capi.CountTable(ctx, 32, size)
    with pytest.raises(capi.RufusError):
        t.set_shard(0, 2)                                             # no minimizer bins outside the MSP path (k <= 31)
    t.free()
    blk.free()


def test_trio_in_shard_passes_equals_one_pass(ctx, small_trio):
    """The whole hot path (count x3 -> set difference -> filter) in 4 minimizer-shard passes gives the
    mutant k-mers, histograms, record counts and pulled pairs of the single-pass run."""
    from rufus_amd import dist as rdist
    from tests.synth import flat_reads
    blocks = {}
    for n in ("child", "mother", "father"):
        seq, qual, off = flat_reads(small_trio[n])
        blocks[n] = ctx.upload(capi.PackedReads(seq, off, qual, 15, capi.PACK_COUNT | capi.PACK_FILTER))
    out = []
    for passes in (1, 4):
        shard = rdist.TrioShard(ctx, 25, 8 << 30, 2, 5, 1200, 1, passes=passes)
        out.append(shard.run(blocks["child"], [blocks["mother"], blocks["father"]]))
    a, b = out
    assert len(a["mutant_keys"]) > 0 and np.array_equal(a["mutant_keys"], b["mutant_keys"])
    assert a["n_records"] == b["n_records"] and a["n_pulled"] == b["n_pulled"] > 0
    assert all(np.array_equal(x, y) for x, y in zip(a["histos"], b["histos"]))
    assert np.array_equal(a["pulled"], b["pulled"])
    for x in blocks.values():
        x.free()


def test_msp_record_segments_export_and_import(ctx, small_trio, monkeypatch):
    """The multi-GPU building blocks on one GPU: the record segments of two tables (different bin counts)
    are exported, split at an owner boundary, imported into two fresh tables and finished -- each result
    holds exactly the k-mers of its bin range, together they are the direct count."""
    import torch
    from rufus_amd import dist as rdist
    k, size = 25, 1 << 28
    fq = {n: [fastq_bytes(small_trio[n], m) for m in (1, 2)] for n in ("child", "mother")}
    reads = {n: [r for f in fq[n] for r in tools.parse_sequences(f)] for n in fq}
    exported = []
    for n, bins in (("child", "256"), ("mother", "1024")):
        monkeypatch.setenv("RFX_P2L_BINS", bins)
        t = capi.CountTable(ctx, k, size, mode=capi.COUNT_MSP)
        blk = ctx.upload(capi.PackedReads.from_reads(reads[n]))
        t.add(blk)
        (d_rec, d_bs, nb, nrec), = t.segments()
        assert nb == int(bins) and nrec > 0
        rec = torch.empty(nrec, dtype=torch.int64, device="cuda")
        ext = torch.empty(nrec, dtype=torch.int32, device="cuda")      # a record = 64-bit word + 32-bit plane
        bs = torch.empty(nb + 1, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        ctx.memcpy_dev(rec.data_ptr(), d_rec, nrec * 8)
        ctx.memcpy_dev(ext.data_ptr(), t.segment_ext(0), nrec * 4)
        ctx.memcpy_dev(bs.data_ptr(), d_bs, (nb + 1) * 8)
        ctx.sync()
        assert int(bs[-1]) == nrec and bool((bs[1:] >= bs[:-1]).all())
        exported.append((rec, bs.cpu(), ext))
        blk.free()
        t.free()
    monkeypatch.delenv("RFX_P2L_BINS")
    ref = oracle.count(None, k, size, lower=1, reads=reads["child"] + reads["mother"])
    got = []
    for owner in range(2):
        t = capi.CountTable(ctx, k, size, mode=capi.COUNT_MSP)
        for rec, bs, ext in exported:
            nb = len(bs) - 1
            b = rdist.bin_owner_bounds(nb, 2)
            lo, hi = int(bs[b[owner]]), int(bs[b[owner + 1]])
            full = torch.zeros(nb + 1, dtype=torch.int64)
            full[b[owner]:b[owner + 1] + 1] = bs[b[owner]:b[owner + 1] + 1] - lo
            full[b[owner + 1] + 1:] = hi - lo
            run, runx, full = rec[lo:hi].clone(), ext[lo:hi].clone(), full.cuda()
            torch.cuda.synchronize()
            t.add_records_dev(run.data_ptr(), run.numel(), full.data_ptr(), nb, runx.data_ptr())
            ctx.sync()
        out = t.finish(1)
        got.append(out.get())
        out.free()
        t.free()
    assert len(got[0][0]) and len(got[1][0]) and not set(got[0][0].tolist()) & set(got[1][0].tolist())
    keys, counts, pos = rdist.merge_shards([(g[0], g[1].astype(np.uint64), g[2]) for g in got])
    assert np.array_equal(keys, ref.keys) and np.array_equal(counts, ref.counts) and np.array_equal(pos, ref.pos)
    # records come with their planes
    t = capi.CountTable(ctx, k, size, mode=capi.COUNT_MSP)
    with pytest.raises(capi.RufusError):
        t.add_records_dev(exported[0][0].data_ptr(), 1, exported[0][1].cuda().data_ptr(), 256)
    t.free()
    # a table that is not on the MSP path refuses records
    t = capi.CountTable(ctx, 31, size, mode=capi.COUNT_P2L)
    with pytest.raises(capi.RufusError):
        t.add_records_dev(exported[0][0].data_ptr(), 1, exported[0][1].cuda().data_ptr(), 256, exported[0][2].data_ptr())
    t.free()


def test_msp_skewed_input_is_redone_with_exact_sizes(ctx, monkeypatch):
    """Homopolymer / dinucleotide reads put almost every record into one bin: the fixed-capacity
    coarse bins and the 16-bit histogram of the one-pass partition both give up, the block is redone
    with the exact two-pass sizing (visible as a k_msp_count launch); ordinary reads are not."""
    monkeypatch.setenv("RFX_P2L_BINS", "2048")
    rng = np.random.default_rng(3)
    seqs = [b"A" * 150] * 2500 + [b"AC" * 75] * 700
    seqs += [bytes(r) for r in np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (500, 150))]]
    k, size = 25, 1 << 24
    ref = oracle.count(None, k, size, lower=1, reads=seqs)
    for skew, reads in ((True, seqs), (False, seqs[-500:])):
        ctx.prof(True)
        ctx.prof_reset()
        t = capi.CountTable(ctx, k, size, mode=capi.COUNT_MSP)
        blk = ctx.upload(capi.PackedReads.from_reads(reads))
        t.add(blk)
        rec = t.finish(1)
        launched = ctx.prof_dict()
        ctx.prof(False)
        assert ("k_msp_count" in launched) == skew, launched
        if skew:
            keys, counts, pos = rec.get()
            assert np.array_equal(keys, ref.keys) and np.array_equal(counts, ref.counts.astype(np.uint32))
            assert np.array_equal(pos, ref.pos)
            assert int(counts.max()) == 2500 * 126
        for x in (rec, blk, t):
            x.free()


def test_key_range_passes_partition_the_output(ctx, small_trio):
    """pos-range passes (multi-pass / multi-GPU ownership): concatenating the slices in pos order
    reproduces the single-pass payload."""
    fq = [fastq_bytes(small_trio["mother"], m) for m in (1, 2)]
    k, size = 25, 1 << 27
    seqs = tools.parse_sequences(fq[0]) + tools.parse_sequences(fq[1])
    whole = oracle.count(fq, k, size, lower=2).payload()
    blk = ctx.upload(capi.PackedReads.from_reads(seqs))
    cuts = [0, 1 << 25, 3 << 25, 1 << 27]
    parts = []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        t = capi.CountTable(ctx, k, size, pos_lo=lo, pos_hi=hi)
        t.add(blk)
        rec = t.finish(2)
        parts.append(rec.payload())
        rec.free()
        t.free()
    blk.free()
    assert b"".join(parts) == whole


def test_add_pairs_merges_partials(ctx, small_trio):
    """Owner-side reduce of the multi-GPU exchange: two half-counts merged as (key,count) pairs."""
    child = small_trio["child"]
    k, size = 25, 1 << 27
    halves = [[r.tobytes() for r in child.s[m]] for m in (0, 1)]
    ref = oracle.count(None, k, size, lower=2, reads=halves[0] + halves[1])
    owner = capi.CountTable(ctx, k, size)
    for h in halves:
        t = capi.CountTable(ctx, k, size)
        blk = ctx.upload(capi.PackedReads.from_reads(h))
        t.add(blk)
        part = t.finish(1)
        dk, dc, _ = part.dev_ptrs()
        owner.add_pairs_dev(dk, dc, len(part))
        ctx.sync()
        part.free(); blk.free(); t.free()
    rec = owner.finish(2)
    assert rec.payload() == ref.payload()
    rec.free(); owner.free()


# ------------------------------------------------------------------------------------------------
# set difference
# ------------------------------------------------------------------------------------------------
def test_testrun_merge_query_hashlist(ctx, testrun):
    files = {s: tools.jellyfish_count(ctx, testrun[s], 25, 100_000_000, lower=2) for s in ("Child", "Mother", "Father")}
    trio = [files["Child"], files["Mother"], files["Father"]]
    merge = tools.rufus_merge(ctx, trio)
    assert merge == testrun["merge"]
    assert tools.check_jelly_hash_list(files["Child"], merge, 5, 140) == testrun["hashlist"]
    assert tools.hash_list(ctx, files["Child"], trio[1:], 5, 140) == testrun["hashlist"]
    # testRun/runDevTest.sh: an exclude list is one more merge input; -m 8
    assert tools.hash_list(ctx, files["Child"], trio[1:] + [files["Mother"]], 8, 140) == testrun["hashlist_dev"]
    # query: absent k-mers print 0, reverse complements are canonicalised
    q = tools.jellyfish_query(files["Child"], ["A" * 25, "T" * 25, "ACGTACGTACGTACGTACGTACGTC"])
    assert q.splitlines()[0] == "A" * 25 + " 48" and q.splitlines()[1] == "A" * 25 + " 48"
    assert q.splitlines()[2].endswith(" 0")
    # inputs counted with another table size cannot be merged (merge_files.cc:193-203)
    other = tools.jellyfish_count(ctx, testrun["Mother"], 25, 1 << 20, lower=2)
    with pytest.raises(capi.RufusError):
        tools.rufus_merge(ctx, [files["Child"], other])
    for f in trio + [other]:
        f.records.free()


def test_synthetic_merge_and_hashlist(ctx, small_trio):
    k, size = 25, 1 << 27
    fq = {n: [fastq_bytes(small_trio[n], m) for m in (1, 2)] for n in ("child", "mother", "father")}
    files = {n: tools.jellyfish_count(ctx, fq[n], k, size, lower=2) for n in fq}
    orc = {n: oracle.count(fq[n], k, size, lower=2) for n in fq}
    got = tools.rufus_merge(ctx, [files["child"], files["mother"], files["father"]])
    assert got == oracle.merge_unique_text([orc["child"], orc["mother"], orc["father"]])
    for lo, hi in ((5, 1200), (2, 9), (8, 8)):
        assert tools.hash_list(ctx, files["child"], [files["mother"], files["father"]], lo, hi) == \
            oracle.hash_list(orc["child"], [orc["mother"], orc["father"]], lo, hi)
    # single input: everything with count >= 5 is "unique"
    assert tools.rufus_merge(ctx, [files["father"]]) == oracle.merge_unique_text([orc["father"]])
    for f in files.values():
        f.records.free()


@pytest.mark.parametrize("route", ["search", "tiles"])
@pytest.mark.parametrize("seed", _more_seeds([31, 32, 33, 34]))
def test_merge_hashlist_query_on_random_configurations(ctx, seed, route, monkeypatch):
    """Randomised K4 inputs: 2-4 samples of very different sizes drawn from overlapping genomes (so the
    (pos,key) search starts far from or right at its target, hits and misses both), random k, table
    size, -L, coverage window; merge text, hash list and query against the oracle.  "tiles" puts these
    small inputs through the kernels big inputs take (a range of the control per tile of candidates in LDS)."""
    if route == "tiles":
        monkeypatch.setenv("RFX_K4_TILE_MIN", "1")
    rng = np.random.default_rng(seed)
    k = int(rng.choice([15, 21, 25, 31]))
    size = 1 << int(rng.integers(12, min(2 * k, 34)))
    lower = int(rng.choice([1, 2, 3]))
    base = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(rng.integers(3000, 30000)))]
    samples = []
    for _ in range(int(rng.integers(2, 5))):
        g = base.copy()
        mut = rng.integers(0, len(g), int(rng.integers(0, 40)))
        g[mut] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, len(mut))]
        if rng.random() < 0.3:
            g = g[:int(len(g) * rng.uniform(0.05, 0.6))]          # a much smaller sample
        cov = int(rng.integers(2, 12))
        n = max(1, cov * len(g) // 100)
        reads = [bytes(g[s0:s0 + 100]) for s0 in rng.integers(0, max(1, len(g) - 100), n)]
        samples.append(b"".join(b">r\n" + r + b"\n" for r in reads))
    files = [tools.jellyfish_count(ctx, [fa], k, size, lower=lower) for fa in samples]
    orcs = [oracle.count([fa], k, size, lower=lower) for fa in samples]
    for f, o in zip(files, orcs):
        assert f.records.payload() == o.payload()
    assert tools.rufus_merge(ctx, files) == oracle.merge_unique_text(orcs)
    for lo, hi in ((1, 10**6), (int(rng.integers(1, 6)), int(rng.integers(6, 40)))):
        assert tools.hash_list(ctx, files[0], files[1:], lo, hi) == oracle.hash_list(orcs[0], orcs[1:], lo, hi)
        # the same one control at a time, device to device (rfx_records_subtract): candidates, then strike out
        cand = capi.records_subtract(ctx, files[0].records, [], max(5, lo), hi)
        for f in files[1:]:
            nxt = capi.records_subtract(ctx, cand, [f.records])
            cand.free()
            cand = nxt
        ck, cc, cp = cand.get()
        assert "".join(f"{t} {int(c)}\n" for t, c in zip(tools.keys_to_text(ck, k), cc)) == \
            oracle.hash_list(orcs[0], orcs[1:], lo, hi)
        assert np.all((cp[1:] > cp[:-1]) | ((cp[1:] == cp[:-1]) & (ck[1:] > ck[:-1])))
        cand.free()
    kmers = [bytes(base[s0:s0 + k]).decode() for s0 in rng.integers(0, len(base) - k, 200)]
    kmers += ["".join(rng.choice(list("ACGT"), k)) for _ in range(50)]
    want = "".join(f"{oracle.jf_decode(key, k)} {c}\n" for key, c in oracle.query(orcs[-1], kmers))
    assert tools.jellyfish_query(files[-1], kmers) == want
    for f in files:
        f.records.free()


# ------------------------------------------------------------------------------------------------
# filter
# ------------------------------------------------------------------------------------------------
def test_testrun_filter_matches_reference_binary(ctx, testrun, tmp_path):
    exp = testrun["expected"]
    hl = tmp_path / "Child.HashList"
    hl.write_text(testrun["hashlist"])
    for m in (1, 2):
        (tmp_path / f"m{m}.fq").write_bytes(testrun["Child"][m - 1])
    stub = str(tmp_path / "out")
    n = tools.rufus_filter(ctx, str(hl), str(tmp_path / "m1.fq"), str(tmp_path / "m2.fq"), stub, 25, 15, 1, 4)
    assert n == 26
    for m in (1, 2):   # reference at 1 thread writes in input order, as we do: byte-identical files
        data = open(f"{stub}.Mutations.Mate{m}.fastq", "rb").read()
        assert hashlib.sha256(data).hexdigest() == exp["filter_paired_sha256"][str(m)]
    n = tools.rufus_filter_single(ctx, str(hl), str(tmp_path / "m1.fq"), stub, 25, 15, 1)
    data = open(f"{stub}.Mutations.fastq", "rb").read()
    assert hashlib.sha256(data).hexdigest() == exp["filter_single_sha256"]
    assert n == len(exp["filter_single_names"])


@pytest.mark.parametrize("k,minq,thresh", [(25, 15, 1), (25, 0, 2), (31, 30, 1), (12, 15, 3)])
def test_synthetic_filter_matches_oracle(ctx, small_trio, k, minq, thresh):
    child = small_trio["child"]
    m1, m2 = fastq_bytes(child, 1), fastq_bytes(child, 2)
    # hash list: k-mers of a handful of child reads (forward or reverse-complement spelling), mixed line formats
    rng = np.random.default_rng(k)
    lines = []
    for i in rng.integers(0, len(child), 12):
        s = child.s[int(rng.integers(0, 2))][i].tobytes().decode()
        j = int(rng.integers(0, len(s) - k - 3))
        for d in range(3):   # three consecutive windows so thresholds 2 and 3 can be met
            km = s[j + d:j + d + k]
            if "N" in km:
                continue
            lines.append(rng.choice([f"{km} 7", f"{km}\t9", f"1 2 3 {km}", km]))
    text = ("\n".join(lines) + "\n").encode()
    fs = oracle.FilterSet(text)
    keys = capi.hashlist_keys(text, k)
    mset = capi.MutantSet(ctx, keys, k)
    pulled = np.zeros(len(child), bool)
    for data, mate in ((m1, 0), (m2, 1)):
        h, s, p, q = tools.parse_fastq4(data)
        blk = ctx.upload(capi.PackedReads.from_reads(s, q, minq, capi.PACK_FILTER))
        hits, mask, nh = mset.filter(blk, thresh, True)
        want = np.array([fs.scan(a, b, k, minq) for a, b in zip(s, q)], dtype=np.uint32)
        assert np.array_equal(hits, want)
        bits = tools._mask_bits(mask, len(s))
        assert np.array_equal(bits, want >= thresh) and nh == int(bits.sum())
        # single-end loop bound (last base examined)
        hits1, _, _ = mset.filter(blk, thresh, False)
        want1 = np.array([fs.scan(a, b, k, minq, single_end=True) for a, b in zip(s, q)], dtype=np.uint32)
        assert np.array_equal(hits1, want1)
        pulled |= bits
        blk.free()
    mset.free()
    assert np.array_equal(np.flatnonzero(pulled), fs.pairs(m1, m2, k, minq, thresh))
    assert pulled.any()


@pytest.mark.parametrize("seed", _more_seeds([11, 12, 13, 14, 15, 16, 17, 18, 19]))
def test_filter_matches_oracle_on_random_configurations(ctx, seed):
    """Randomised: k, MinQ, threshold, set size (LDS bitmap vs HBM probe), read shapes (short, N, low
    quality, lower case, homopolymer), both loop bounds -- per-read hit counts against the oracle."""
    rng = np.random.default_rng(seed)
    k = int(rng.choice([5, 12, 19, 21, 25, 31, 32]))
    minq = int(rng.choice([0, 2, 15, 30]))
    thresh = int(rng.choice([1, 1, 2, 5]))
    genome = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(rng.integers(500, 20000)))]
    reads, quals = [], []
    for _ in range(int(rng.integers(50, 1500))):
        n = int(rng.choice([1, k - 1, k, k + 1, 2 * k, 101, 150, 251]))
        s0 = int(rng.integers(0, max(1, len(genome) - n)))
        r = genome[s0:s0 + n].copy()
        q = rng.choice(np.frombuffer(b"#(5?IJ", np.uint8), len(r), p=[.03, .03, .04, .1, .3, .5])
        if rng.random() < 0.15 and len(r):
            r[rng.integers(0, len(r))] = rng.choice(np.frombuffer(b"NnRacgt", np.uint8))
        if rng.random() < 0.03:
            r[:] = ord("ACGT"[int(rng.integers(0, 4))])
        reads.append(bytes(r))
        quals.append(bytes(q))
    n_set = int(rng.choice([1, 20, 400, 8000, 30000, 45000]))   # (both orientations: 90 000 entries are k_filter_q's again)
    kmers = []
    for _ in range(n_set):
        s0 = int(rng.integers(0, len(genome) - k)) if len(genome) > k else 0
        km = bytes(genome[s0:s0 + k]).decode()
        kmers.append(km if rng.random() < 0.7 else "".join(rng.choice(list("ACGT"), k)))
    text = ("\n".join(f"{x} {int(rng.integers(1, 99))}" for x in kmers) + "\n").encode()
    fs = oracle.FilterSet(text)
    mset = capi.MutantSet(ctx, capi.hashlist_keys(text, k), k)
    blk = ctx.upload(capi.PackedReads.from_reads(reads, quals, minq, capi.PACK_FILTER))
    for skipped, single in ((True, False), (False, True)):
        hits, mask, nh = mset.filter(blk, thresh, skipped)
        want = np.array([fs.scan(a, b, k, minq, single_end=single) for a, b in zip(reads, quals)], dtype=np.uint32)
        assert np.array_equal(hits, want), (k, minq, thresh, n_set, np.flatnonzero(hits != want)[:5])
        bits = tools._mask_bits(mask, len(reads))
        assert np.array_equal(bits, want >= thresh) and nh == int(bits.sum())
    blk.free()
    mset.free()


@pytest.mark.parametrize("n_random,k", [(60_000, 25), (126_000, 25), (110_000, 31)])
def test_filter_mask_only_equals_counts_on_large_sets(ctx, n_random, k, monkeypatch):
    """Sets of > 10^5 keys keep the pair filter's candidate queue draining all the time.  With thresh = 1 and nobody asking for
    counts the hits set the mask's bits themselves (k_filter_p MASK): that mask must be the counting mode's (count >= 1), the
    same in every run, and k_filter_q's -- on a uniform (compact) block of millions of reads with few of them hit, where
    round 6's 64-bit atomic ORs on the mask word set bits of reads WITHOUT a hit and lost a true one now and then (a self-check
    at 300x coverage found the pulled pairs differing from run to run), and on a ragged block."""
    from rufus_amd import wgs
    monkeypatch.setenv("RFX_FILTER_PAIR_MAX_LOG2", "18")   # (since the fix such sets go to k_filter_q: here the pair filter keeps them)
    rng = np.random.default_rng(n_random + k)
    sy = capi.Synth.sample(30_000_000, 0, n_snv=20, seed=77)
    blk = wgs.make_sample(ctx, sy, 1_500_000, 1 << 24, 15, want_good=True, compact=True)[0]       # 3 M reads, one block
    genome = np.frombuffer(sy.genome(0, 30_000_000), np.uint8)
    loci = rng.integers(0, len(genome) - 200, 60)
    own = [bytes(genome[p0 + i:p0 + i + k]) for p0 in loci for i in range(100)]                   # 6000 k-mers of the genome
    rnd = ["".join(x).encode() for x in rng.choice(list("ACGT"), (n_random, k))]
    text = b"".join(km + b" 7\n" for km in own + rnd)
    keys = capi.hashlist_keys(text, k)
    mset = capi.MutantSet(ctx, keys, k)
    hits, mask_c, n_c = mset.filter(blk, 1, True, want_hits=True, want_mask=True)
    want = tools._mask_bits(mask_c, blk.n)
    assert np.array_equal(want, hits >= 1) and 500 < int(want.sum()) == n_c < blk.n // 20
    for _ in range(16):     # (the old build failed in about one run of three)
        _, mask_m, n_m = mset.filter(blk, 1, True, want_hits=False, want_mask=True)
        got = tools._mask_bits(mask_m, blk.n)
        assert int((got & ~want).sum()) == 0 and int((want & ~got).sum()) == 0 and n_m == n_c
    (m_many, n_many), = mset.filter_many([blk], 1, last_base_skipped=True)
    assert np.array_equal(tools._mask_bits(m_many, blk.n), want) and n_many == n_c
    mset.free()
    monkeypatch.delenv("RFX_FILTER_PAIR_MAX_LOG2")     # the default choice for a set of this size: k_filter_q
    mq = capi.MutantSet(ctx, keys, k)
    hq, _, n_q = mq.filter(blk, 1, True, want_hits=True, want_mask=True)
    assert np.array_equal(hq, hits) and n_q == n_c
    _, mask_q, n_q2 = mq.filter(blk, 1, True, want_hits=False, want_mask=True)
    assert np.array_equal(tools._mask_bits(mask_q, blk.n), want) and n_q2 == n_c
    mq.free()
    blk.free()


def test_filter_edge_cases(ctx):
    k = 5
    text = b"ACGTA 3\nTTTTT 9\n"
    fs = oracle.FilterSet(text)
    mset = capi.MutantSet(ctx, capi.hashlist_keys(text, k), k)
    seqs = [b"ACGTAC", b"ACGTA", b"TACGT", b"ACGT", b"A", b"AAAAANAAAAA", b"TTTTTT", b"ACGTXACGTAA", b"acgtaA",
            b"ACGTA" * 50, b"TACGTAG", b"ACNTAACGTAT"]
    quals = [b"JJJJJJ", b"JJJJJ", b"JJJJJ", b"JJJJ", b"J", b"JJJJJJJJJJJ", b"JJJ#JJ", b"JJJJJJJJJJJ", b"JJJJJJ",
             b"J" * 250, b"JJJ", b"J" * 11]
    blk = ctx.upload(capi.PackedReads.from_reads(seqs, quals, 15, capi.PACK_FILTER))
    for skipped, single in ((True, False), (False, True)):
        hits, mask, _ = mset.filter(blk, 1, skipped)
        want = [fs.scan(a, b, k, 15, single_end=single) for a, b in zip(seqs, quals)]
        assert hits.tolist() == want
    blk.free()
    # a large set is probed from HBM instead of LDS
    rng = np.random.default_rng(3)
    kmers = ["".join(rng.choice(list("ACGT"), 21)) for _ in range(6000)]
    text = ("\n".join(f"{x} 5" for x in kmers) + "\n").encode()
    reads = [("".join(rng.choice(list("ACGT"), 40)) + kmers[i] + "ACGTACGT").encode() for i in range(0, 6000, 37)]
    reads += [("".join(rng.choice(list("ACGT"), 100))).encode() for _ in range(100)]
    quals = [b"J" * len(r) for r in reads]
    fs = oracle.FilterSet(text)
    big = capi.MutantSet(ctx, capi.hashlist_keys(text, 21), 21)
    blk = ctx.upload(capi.PackedReads.from_reads(reads, quals, 15, capi.PACK_FILTER))
    hits, _, _ = big.filter(blk, 1, True)
    assert hits.tolist() == [fs.scan(a, b, 21, 15) for a, b in zip(reads, quals)]
    blk.free(); big.free(); mset.free()


# ------------------------------------------------------------------------------------------------
# full-size properties (BASELINE.json configs[1]: synthetic 1M x 150 bp trio, k=25)
# ------------------------------------------------------------------------------------------------
@pytest.mark.skipif(os.environ.get("RFX_SKIP_S1") == "1", reason="RFX_SKIP_S1=1")
def test_s1_full_size_properties(ctx):
    """At the size the oracle no longer finishes in seconds, check size-independent properties:
    total = sum over the histogram, the planted SNVs are exactly the mutant k-mers (25 per SNV),
    records strictly sorted, child-only k-mers absent from both parents, and a 40k-read slice of
    the same input still matches the oracle bit for bit."""
    trio = make_trio()   # 5 Mb genome, 0.5 M pairs/sample, 20 SNVs, seed 12345
    k, size = 25, 8 << 30
    files = {}
    for name in ("child", "mother", "father"):
        s = trio[name]
        t = capi.CountTable(ctx, k, size, capacity=1 << 26)
        for m in (0, 1):
            off = np.arange(len(s) + 1, dtype=np.uint64) * np.uint64(150)
            blk = ctx.upload(capi.PackedReads(s.s[m].tobytes(), off))
            t.add(blk)
            blk.free()
        rec_all, h_all = t.finish(1, want_histo=True)
        n_valid_windows = int(sum(int(x) * i for i, x in enumerate(h_all)))   # no count reaches 10001 at 30x
        # every ACGT-only window was counted exactly once
        want = 0
        for m in (0, 1):
            isn = (s.s[m] == ord("N"))
            run = np.zeros(len(s), dtype=np.int64)
            tot = 0
            for j in range(150):
                run = np.where(isn[:, j], 0, run + 1)
                tot += int((run >= k).sum())
            want += tot
        assert n_valid_windows == want
        rec_all.free()
        rec = t.finish(2)
        files[name] = tools.JhashFile(rec, capi.jf_matrix(33, k), True)
        keys, counts, pos = rec.get()
        assert np.all((pos[1:] > pos[:-1]) | ((pos[1:] == pos[:-1]) & (keys[1:] > keys[:-1])))
        assert int(counts.min()) >= 2
        t.free()
    hl = tools.hash_list(ctx, files["child"], [files["mother"], files["father"]], 5, 1200)
    kmers = [ln.split()[0] for ln in hl.splitlines()]
    assert len(kmers) == 25 * 20
    for other in ("mother", "father"):
        assert not tools.jellyfish_query(files[other], kmers).replace(" 0\n", "\n").count(" ")
    # slice parity against the oracle
    sl = [trio["child"].s[0][i].tobytes() for i in range(40000)]
    jf = tools.jellyfish_count(ctx, [b"".join(b">r\n" + r + b"\n" for r in sl)], k, size, lower=2)
    assert jf.records.payload() == oracle.count(None, k, size, lower=2, reads=sl).payload()
    jf.records.free()
    for f in files.values():
        f.records.free()


# ------------------------------------------------------------------------------------------------
# multi-GPU path on the real HIP backend: two ranks share the one GPU, exchange over gloo
# ------------------------------------------------------------------------------------------------
def _dist_worker(rank, world, port, q, shard_by):
    import torch.distributed as dist
    from rufus_amd import dist as rdist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests.synth import flat_reads
        c = capi.Context(0)
        trio = make_trio(genome_len=40_000, n_pairs=3000, n_snv=5, seed=8, read_seed=50 + rank)
        blocks = {}
        for n in ("child", "mother", "father"):
            seq, qual, off = flat_reads(trio[n])
            blocks[n] = c.upload(capi.PackedReads(seq, off, qual, 15, capi.PACK_COUNT | capi.PACK_FILTER))
        shard = rdist.TrioShard(c, 25, 8 << 30, 2, 5, 1200, 1, group=dist.group.WORLD, shard_by=shard_by)
        res = shard.run(blocks["child"], [blocks["mother"], blocks["father"]], keep_records=True)
        q.put((rank, [tuple(x.tolist() for x in r.get()) for r in res["records"]], [h.tolist() for h in res["histos"]],
               res["mutant_keys"].tolist(), res["pulled"].tolist(), res["n_pulled"]))
        c.close()
    finally:
        dist.destroy_process_group()


def test_collectives_on_a_one_rank_rccl_group():
    """The exchange code issued against the real RCCL backend (one rank: a 1-GPU box cannot do more):
    all_to_all_single with device split tensors, the async variant, all_reduce, all_gather."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "rccl_selftest.py")], capture_output=True, text=True,
                       timeout=240, cwd=os.path.dirname(here))
    assert r.returncode == 0 and "rccl self-test ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("shard_by", ["minimizer", "pos"])
def test_two_ranks_exchange_on_the_hip_backend(shard_by):
    """minimizer: ranks exchange super-k-mer records by bin owner and count complete bins (shards
    interleave to the single-GPU result); pos: ranks exchange (key,count) partials by pos owner (slices
    concatenate to it)."""
    import socket
    import torch.multiprocessing as mp
    from rufus_amd import dist as rdist
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_dist_worker, args=(r, world, port, q, shard_by)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    parts = [make_trio(genome_len=40_000, n_pairs=3000, n_snv=5, seed=8, read_seed=50 + r) for r in range(world)]
    recs = []
    for n in ("child", "mother", "father"):
        reads = [x.tobytes() for t in parts for m in (0, 1) for x in t[n].s[m]]
        recs.append(oracle.count(None, 25, 8 << 30, lower=2, reads=reads))
    for i in range(3):
        shards = [(np.array(g[1][i][0], np.uint64), np.array(g[1][i][1], np.uint64), np.array(g[1][i][2], np.uint64))
                  for g in got]
        if shard_by == "pos":
            keys_, counts_, pos_ = (np.concatenate([s_[j] for s_ in shards]) for j in range(3))
        else:
            assert all(len(s_[0]) > 0 for s_ in shards)
            keys_, counts_, pos_ = rdist.merge_shards(shards)
        assert np.array_equal(keys_, recs[i].keys) and np.array_equal(counts_, recs[i].counts)
        assert np.array_equal(pos_, recs[i].pos)
        assert got[0][2][i] == got[1][2][i] == oracle.histo(recs[i].counts, full=True)[0].tolist()
    hl = oracle.hash_list(recs[0], recs[1:], 5, 1200)
    want_keys = [oracle.jf_encode(ln.split()[0]) for ln in hl.splitlines()]
    assert got[0][3] == got[1][3] == want_keys and want_keys
    fs = oracle.FilterSet(hl.encode())
    tot = 0
    for r in range(world):
        c = parts[r]["child"]
        from tests.synth import fastq_bytes as fqb
        want = np.zeros(len(c), bool)
        want[fs.pairs(fqb(c, 1), fqb(c, 2), 25, 15, 1)] = True
        assert got[r][4] == want.tolist()
        tot += int(want.sum())
    assert got[0][5] == got[1][5] == tot and tot > 0


def test_filter_hash_list_entries_longer_than_k(ctx):
    """Util::HashToLong packs up to 32 bases: a list entry longer than K has bits above 2K and matches no window --
    unless its extra bases encode 00 ('A').  The loader keeps that quirk."""
    k = 25
    core = b"ACGTTGCAAGGCTTAACCGGATATC"
    reads = [b"GG" + core + b"TTGA", b"GG" + core + b"ATGA", b"CCCC" + core[::-1] + b"AAAA"]
    quals = [b"I" * len(r) for r in reads]
    for extra in (b"C", b"A", b"AA", b"AC", b""):
        text = (core + extra).decode() + " 7\n"
        fs = oracle.FilterSet(text.encode())
        mset = capi.MutantSet(ctx, capi.hashlist_keys(text.encode(), k), k)
        blk = ctx.upload(capi.PackedReads.from_reads(reads, quals, 15, capi.PACK_FILTER))
        hits, _, _ = mset.filter(blk, 1, True)
        want = [fs.scan(a, b, k, 15) for a, b in zip(reads, quals)]
        assert hits.tolist() == want, (extra, hits.tolist(), want)
        blk.free()
        mset.free()
