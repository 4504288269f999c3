"""CPU-only: pins the oracle against the reference's own known answers."""
import hashlib

import numpy as np

import oracle


def mt_sequence(seed, lengths):
    """bin/generate_sequence of jellyfish 2.2.5 (jellyfish/generate_sequence.cc:30-47 over
    jellyfish/mersenne.cpp): MT19937 init_genrand(seed), 37 outputs discarded, 16 letters per
    32-bit output taken from the low bits up."""
    bg = np.random.MT19937()
    bg._legacy_seeding(seed & 0xFFFFFFFF)
    bg.random_raw(37)
    out = []
    for n in lengths:
        raw = bg.random_raw((n + 15) // 16).astype(np.uint32)
        sh = (np.arange(16, dtype=np.uint32) * 2)[None, :]
        codes = ((raw[:, None] >> sh) & 3).reshape(-1)[:n]
        out.append(np.frombuffer(b"ACGT", dtype=np.uint8)[codes].tobytes())
    return out


def test_matrix_probe_values():
    # SURVEY.md 8a-F': production matrices of `jellyfish count -s 8G` (r=33) for k=25 and k=31
    m = oracle.jf_matrix(33, 25)
    assert m[:4].tolist() == [2614319272, 7343572143, 6131638113, 1979759137]
    assert m[-2:].tolist() == [6332333856, 8294244289]
    m = oracle.jf_matrix(33, 31)
    assert m[:4].tolist() == [6002715614, 2001800797, 1638428665, 5771199851]
    assert m[-2:].tolist() == [7994934477, 5934309508]


def test_jellyfish_md5_kats():
    """tests/parallel_hashing.sh and tests/multi_file.sh inside jellyfish-2.2.5.tar.gz: md5 of
    `jellyfish histo` after `count -m 15 -C` on the seeded sequences of tests/generate_sequence.sh."""
    seq10m, = mt_sequence(3141592653, [10_000_000])
    assert seq10m[:20] == b"GAACCTCATGGTACAGTCAG"
    rec = oracle.count(None, 15, 2 << 20, reads=[seq10m])
    txt = oracle.histo(rec.counts)[1]
    assert hashlib.md5(txt.encode()).hexdigest() == "864c0b0826854bdc72a85d170549b64b"
    # `jellyfish stats` of the same database (tests/parallel_hashing.sh: ${pref}_m15.stats)
    assert hashlib.md5(oracle.stats_text(rec.counts).encode()).hexdigest() == "41fd8408dde0ea14bec7425b1a877140"
    rec = oracle.count(None, 15, 2 << 20, lower=2, upper=3, reads=[seq10m])
    txt = oracle.histo(rec.counts)[1]
    # (the same md5 pins both the plain -L2 -U3 run and the --disk automerge run of parallel_hashing.sh)
    assert hashlib.md5(txt.encode()).hexdigest() == "94625cd2d59e278f08421a673eb0926a"
    seq1m = mt_sequence(1040104553, [1_000_000] * 5)
    rec = oracle.count(None, 15, 2 << 20, reads=seq1m[:3] + [seq10m] + seq1m[3:])
    txt = oracle.histo(rec.counts)[1]
    assert hashlib.md5(txt.encode()).hexdigest() == "d93b7678037814c256d1d9120a0e6422"


def test_jellyfish_md5_kat_of_a_subset_count():
    """tests/subset_hashing.sh inside jellyfish-2.2.5.tar.gz: `count -m 10 -C --if seq1m_0.fa --if seq1m_2.fa seq1m_1.fa
    seq1m_0.fa seq1m_3.fa seq1m_2.fa` counts, over the four inputs, only the k-mers that occur in the --if files
    (jf/sub_commands/count_main.cc: the --if k-mers are loaded with count 0, the counting pass only updates what is
    there); md5 of its `jellyfish histo`.  (-s 6M asks for more slots than 4^10: jellyfish cuts the table to 2k bits;
    the histogram does not depend on it.  The script's other line, -m 35, is beyond the 32 bases of a 64-bit key.)"""
    seq1m = mt_sequence(1040104553, [1_000_000] * 5)
    everything = oracle.count(None, 10, 1 << 20, reads=[seq1m[1], seq1m[0], seq1m[3], seq1m[2]])
    subset = oracle.count(None, 10, 1 << 20, reads=[seq1m[0], seq1m[2]])
    keep = np.isin(everything.keys, subset.keys)
    assert 0 < int(keep.sum()) < len(everything.keys)
    txt = oracle.histo(everything.counts[keep])[1]
    assert hashlib.md5(txt.encode()).hexdigest() == "8eb6d4a50aeba178e4847c2da71dbb70"


def test_testrun_count_merge_hashlist(testrun):
    exp = testrun["expected"]
    recs = {}
    for s in ("Child", "Mother", "Father"):
        r = oracle.count(testrun[s], 25, 100_000_000, lower=2)
        e = exp["samples"][s]["s100M"]
        assert len(r.keys) == e["records"]
        assert hashlib.sha256(r.payload()).hexdigest() == e["payload_sha256"]
        assert int(r.counts.sum()) == e["sum_counts"] and int(r.counts.max()) == e["max_count"]
        # records really are in (pos, key) order
        order = np.lexsort((r.keys, r.pos))
        assert np.array_equal(order, np.arange(len(order)))
        recs[s] = r
    # SURVEY probe values of the compiled reference
    assert [len(recs[s].keys) for s in ("Child", "Mother", "Father")] == [18356, 18364, 17390]
    assert recs["Child"].payload()[:11].hex() == "0000000000000030000000"   # poly-A x48
    merge = oracle.merge_unique_text([recs["Child"], recs["Mother"], recs["Father"]])
    assert merge == testrun["merge"] and merge.count("\n") == 411
    hl = oracle.hash_list(recs["Child"], [recs["Mother"], recs["Father"]], 5, 140)
    assert hl == testrun["hashlist"] and hl.count("\n") == 50


def test_testrun_filter_matches_reference_binary(testrun):
    """Pulled read names recorded from the real RUFUS.Filter / RUFUS.Filter.single binaries."""
    exp = testrun["expected"]
    fs = oracle.FilterSet(testrun["hashlist"].encode())
    assert len(fs) == 100
    m1, m2 = testrun["Child"]
    pulled = fs.pairs(m1, m2, 25, 15, 1)
    names = m1.decode().split("\n")[0::4]
    assert [names[i] for i in pulled] == exp["filter_paired_names"]
    # single-end: full length scanned, ":MH<n>" appended
    fs1 = oracle.FilterSet(testrun["hashlist"].encode(), single_end=True)
    lines = m1.split(b"\n")
    got = []
    for i in range(len(lines) // 4):
        n = fs1.scan(lines[4 * i + 1], lines[4 * i + 3], 25, 15, single_end=True)
        if n >= 1:
            got.append(lines[4 * i].decode() + f":MH{n}")
    assert got == exp["filter_single_names"]


def test_header_roundtrip():
    r = oracle.count(None, 25, 1 << 20, reads=[b"ACGTACGTACGTACGTACGTACGTACGTAAAA"])
    blob = oracle.header_bytes(r) + r.payload()
    assert (len(oracle.header_bytes(r)) % 8) == 0
    hdr, payload = oracle.parse_jhash(blob)
    assert hdr["key_len"] == 50 and hdr["size"] == 1 << 20 and hdr["format"] == "binary/sorted"
    r2 = oracle.records_from_payload(hdr, payload)
    assert np.array_equal(r2.keys, r.keys) and np.array_equal(r2.counts, r.counts)


def test_rufus_codec_quirks():
    # src/Util.cpp:51-84: A=00 C=(0,1) G=(1,0) T=(1,1) little endian => A0 G1 C2 T3 at shift 2i
    assert oracle.hash_to_long(b"A") == 0 and oracle.hash_to_long(b"G") == 1
    assert oracle.hash_to_long(b"C") == 2 and oracle.hash_to_long(b"T") == 3
    assert oracle.hash_to_long(b"AC") == 2 << 2
    assert oracle.hash_to_long(b"AXT") == 3 << 4      # unknown characters leave 00
