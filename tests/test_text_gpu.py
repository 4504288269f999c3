"""GPU: K1 on the device (rfx_text_*, rufus_amd/csrc/rfx_text.hip) against the host packer rfx_pack_reads, which the
oracle-checked parity tests of the count and the filter already hold to the reference's tables
(jf/include/jellyfish/mer_dna.hpp:46-63, src/Util.cpp:51-84, src/RUFUS.Filter.cpp:205): the same records must give
the same block, bit for bit, for both packings; text that is not strict 4-line FASTQ must be refused, not guessed at."""
import numpy as np
import pytest

from rufus_amd import capi

pytestmark = pytest.mark.gpu


def _records(rng, n, ragged=True):
    alpha = np.frombuffer(b"ACGTACGTACGTACGTacgtNnRYKM.-*\r", np.uint8)
    seqs, quals, names = [], [], []
    for i in range(n):
        if ragged:
            L = int(rng.choice([0, 1, 5, 24, 25, 31, 32, 33, 63, 64, 65, 100, 150, 151, 250, 301]))
        else:
            L = 150
        s = alpha[rng.integers(0, len(alpha), L)] if rng.random() < 0.3 else alpha[rng.integers(0, 16, L)]
        q = rng.integers(33, 75, L).astype(np.uint8)
        q[rng.random(L) < 0.05] = ord("#")
        q[rng.random(L) < 0.01] = 10 + 128  # (a byte that is negative as a signed char)
        q[q == 10] = 11                      # (never a newline inside the line)
        seqs.append(s.tobytes())
        quals.append(q.tobytes())
        names.append(b"@r%d/1 some text + @ more" % i)
    return names, seqs, quals


def _fastq(names, seqs, quals, plus=b"+"):
    return b"".join(n + b"\n" + s + b"\n" + plus + b"\n" + q + b"\n" for n, s, q in zip(names, seqs, quals))


@pytest.mark.parametrize("seed,n,ragged", [(1, 1, True), (2, 777, True), (3, 60_000, False), (4, 20_000, True)])
@pytest.mark.parametrize("flags", [capi.PACK_COUNT, capi.PACK_FILTER])
def test_device_parse_equals_host_pack(ctx, seed, n, ragged, flags):
    rng = np.random.default_rng(seed)
    names, seqs, quals = _records(rng, n, ragged)
    text = _fastq(names, seqs, quals, plus=b"+" if seed % 2 else b"+the name again")
    arena = capi.TextArena(ctx, len(text) + 1000)
    # in pieces, as the ingest appends them (record-aligned)
    cuts = sorted(set([0, len(text)] + ([text.index(b"\n@r%d/" % i) + 1 for i in rng.integers(1, n, 5)] if n > 1 else [])))
    for a, b in zip(cuts, cuts[1:]):
        arena.append(text[a:b])
    blk = arena.parse(flags, 15)
    assert blk is not None and blk.n == n
    ref = capi.PackedReads.from_reads(seqs, quals, 15, flags)
    got = blk.get(want_good=flags == capi.PACK_FILTER, want_acgt=flags == capi.PACK_COUNT)
    nw = int(ref.word_off[-1])
    assert np.array_equal(got["word_off"], ref.word_off) and np.array_equal(got["len"][:n], ref.len[:n])
    assert np.array_equal(got["codes"][:nw], ref.codes[:nw])
    if flags == capi.PACK_COUNT:
        assert np.array_equal(got["acgt"][:nw], ref.acgt[:nw])
    else:
        assert np.array_equal(got["good"][:nw], ref.good[:nw])
    assert blk.bases == sum(len(s) for s in seqs)
    # the arena can take the next block's text
    arena.reset()
    arena.append(text[:cuts[1]])
    blk2 = arena.parse(flags, 15)
    assert blk2 is not None and blk2.n >= 1
    assert arena.fetch() == text[:cuts[1]]
    blk.free()
    blk2.free()
    arena.close()


@pytest.mark.parametrize("case", ["blank_line", "no_final_newline", "multi_line", "no_plus", "short_quality", "not_at", "three_lines",
                                  "empty"])
def test_text_that_is_not_strict_fastq_is_refused(ctx, case):
    rng = np.random.default_rng(9)
    names, seqs, quals = _records(rng, 50, ragged=False)
    text = _fastq(names, seqs, quals)
    if case == "blank_line":
        i = text.index(b"\n@r20/") + 1
        text = text[:i] + b"\n" + text[i:]
    elif case == "no_final_newline":
        text = text[:-1]
    elif case == "multi_line":
        i = text.index(seqs[7]) + 70
        text = text[:i] + b"\n" + text[i:]
    elif case == "no_plus":
        text = text.replace(b"\n+\n", b"\n-\n", 1)
    elif case == "short_quality":
        i = text.index(quals[3])
        text = text[:i] + text[i + 1:]
    elif case == "not_at":
        text = b">" + text[1:]
    elif case == "three_lines":
        text = text[:text.rindex(quals[-1])]
    elif case == "empty":
        text = b""
    arena = capi.TextArena(ctx, 1 << 20)
    if text:
        arena.append(text)
    assert arena.parse(capi.PACK_COUNT, 0) is None
    assert arena.fetch() == text
    arena.close()


def test_count_of_a_device_parsed_block_is_the_oracle_count(ctx):
    """End to end through the new entry: text -> device parse -> count == the oracle's count of the same FASTQ."""
    import oracle
    from tests.synth import make_trio, fastq_bytes
    trio = make_trio(genome_len=30000, n_pairs=2500, n_snv=3, seed=5)
    fq = [fastq_bytes(trio["child"], m) for m in (1, 2)]
    arena = capi.TextArena(ctx, sum(map(len, fq)) + 16)
    for f in fq:
        arena.append(f)
    blk = arena.parse(capi.PACK_COUNT, 0)
    assert blk is not None
    t = capi.CountTable(ctx, 25, 1 << 22)
    t.add(blk)
    rec = t.finish(2)
    assert rec.payload() == oracle.count(fq, 25, 1 << 22, lower=2).payload()
    for x in (rec, t, blk):
        x.free()
    arena.close()
