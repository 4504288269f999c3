"""CPU-only: the C-ABI library loads, exports every declared symbol, and its host-only helpers agree
with the oracle.  No device compute is attempted here."""
import json
import os
import re

import numpy as np
import pytest

import oracle
from rufus_amd import capi, tools
from tests.conftest import ROOT


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "rufus_hip.h")).read()
    declared = set(re.findall(r"\b(rfx_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    L = capi.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/rufus_hip.h but not exported"
    assert declared == set(capi.SIGNATURES), declared ^ set(capi.SIGNATURES)
    assert b"gfx950" in L.rfx_version()


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.RufusError):
        capi.Context(0)


def test_product_never_imports_oracle():
    for dp, _, files in os.walk(os.path.join(ROOT, "rufus_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "librufus_oracle" not in src, f


@pytest.mark.parametrize("lsize,k", [(33, 25), (33, 31), (27, 25), (21, 15), (10, 25), (50, 25), (62, 31)])
def test_matrix_matches_oracle(lsize, k):
    a, b = capi.jf_matrix(lsize, k), oracle.jf_matrix(lsize, k)
    assert np.array_equal(a, b)
    rng = np.random.default_rng(lsize * 100 + k)
    for key in rng.integers(0, 1 << (2 * k), 50, dtype=np.uint64):
        assert capi.jf_pos(a, k, lsize, int(key)) == oracle.jf_pos(b, int(key), lsize)


def test_pack_layout_and_masks():
    seqs = [b"ACGTNacgtRX", b"", b"T" * 33, b"G" * 64]
    quals = [b"JJJJJ#JJJJJ", b"", b"J" * 33, b"5" * 10]        # last one short: missing quality = bad
    p = capi.PackedReads.from_reads(seqs, flags=capi.PACK_COUNT)
    assert p.word_off.tolist() == [0, 1, 1, 3, 5] and p.len[:4].tolist() == [11, 0, 33, 64]
    codes = [(int(p.codes[0]) >> (2 * i)) & 3 for i in range(11)]
    assert codes == [0, 1, 2, 3, 0, 0, 1, 2, 3, 0, 0]                    # N, R, X pack as 0
    assert int(p.acgt[0]) == 0b00111101111                                 # ACGT acgt valid, N R X not
    assert int(p.codes[2]) == 3 and int(p.acgt[2]) == 1 and int(p.acgt[1]) == 0xFFFFFFFF
    with pytest.raises(capi.RufusError):                                   # lower-case c/g/t: encodings differ
        capi.PackedReads.from_reads(seqs, quals, 15, capi.PACK_COUNT | capi.PACK_FILTER)
    f = capi.PackedReads.from_reads(seqs, quals, 15, capi.PACK_FILTER)
    codes = [(int(f.codes[0]) >> (2 * i)) & 3 for i in range(11)]
    assert codes == [0, 1, 2, 3, 0, 0, 0, 0, 0, 0, 0]                    # Util::HashToLong: only upper-case ACGT
    assert int(f.good[0]) == 0b11111001111                                 # 'N' and '#' (Q2 < 15) are bad
    assert int(f.good[3]) == (1 << 10) - 1 and int(f.good[4]) == 0          # '5' = Q20 ok; missing quals bad


def _pack_rules(seq: bytes, qual, min_q: int, want_count: bool, want_filter: bool):
    """The packing rules restated base by base (rufus_amd/csrc/rfx_host.cpp states them once for its scalar and
    its 32-bases-per-step implementation)."""
    nw = (len(seq) + 31) // 32
    codes, acgt, good = [0] * nw, [0] * nw, [0] * nw
    for i, ch in enumerate(seq):
        c = bytes([ch])
        if want_filter:
            code = b"ACGT".find(c) if c in b"ACGT" else 0
            q = qual[i] if qual is not None else 0
            q = q - 256 if q > 127 else q                                   # plain (signed) char
            good[i // 32] |= int(not (q - 33 < min_q or c == b"N")) << (i % 32)
        else:
            code = b"ACGT".find(c.upper()) if c.upper() in b"ACGT" and c.isalpha() else 0
        codes[i // 32] |= code << (2 * (i % 32))
        if c in b"ACGTacgt":
            acgt[i // 32] |= 1 << (i % 32)
    return codes, acgt, good


@pytest.mark.parametrize("scalar", [False, True])
def test_pack_every_byte_value_and_length(scalar):
    """Both implementations of the packer (RFX_PACK_SCALAR=1 picks the scalar one at load time, hence the child
    process) against the rules above: all 256 byte values as bases and as qualities, lengths around the word size."""
    if scalar:
        import subprocess, sys
        env = dict(os.environ, RFX_PACK_SCALAR="1")
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", __file__, "-k",
                            "test_pack_every_byte_value_and_length and False"], env=env, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        return
    rng = np.random.default_rng(77)
    alphabet = np.frombuffer(b"ACGT" * 12 + b"acgtNn" + bytes(range(256)), np.uint8)
    upper_only = np.frombuffer(b"ACGT" * 12 + b"aNn" + bytes(x for x in range(256) if x not in b"cgt"), np.uint8)
    lens = list(range(0, 70)) + [95, 96, 97, 127, 128, 129, 150, 151, 250, 1000]
    for min_q in (15, 0, -40, 95, 200):
        seqs = [bytes(rng.choice(alphabet, n)) for n in lens]
        quals = [bytes(rng.integers(0, 256, n, dtype=np.uint8)) for n in lens]
        p = capi.PackedReads.from_reads(seqs, flags=capi.PACK_COUNT)
        f = capi.PackedReads.from_reads(seqs, quals, min_q, capi.PACK_FILTER)
        seqs_u = [bytes(rng.choice(upper_only, n)) for n in lens]
        b = capi.PackedReads.from_reads(seqs_u, quals, min_q, capi.PACK_COUNT | capi.PACK_FILTER)
        for r, n in enumerate(lens):
            w0, w1 = int(p.word_off[r]), int(p.word_off[r + 1])
            assert w1 - w0 == (n + 31) // 32
            c, a, _ = _pack_rules(seqs[r], None, 0, True, False)
            assert p.codes[w0:w1].tolist() == c and p.acgt[w0:w1].tolist() == a, (r, n)
            c, _, g = _pack_rules(seqs[r], quals[r], min_q, False, True)
            assert f.codes[w0:w1].tolist() == c and f.good[w0:w1].tolist() == g, (r, n, min_q)
            c, a, g = _pack_rules(seqs_u[r], quals[r], min_q, True, True)
            assert b.codes[w0:w1].tolist() == c and b.acgt[w0:w1].tolist() == a and b.good[w0:w1].tolist() == g
    for bad in (b"c", b"g", b"t"):                                           # wherever it sits in a word
        for at in (0, 5, 31, 32, 40, 63, 64, 70):
            seq = b"A" * at + bad + b"A" * 3
            with pytest.raises(capi.RufusError):
                capi.PackedReads.from_reads([seq], [b"J" * len(seq)], 15, capi.PACK_COUNT | capi.PACK_FILTER)


def test_hashlist_keys_match_oracle_set(testrun):
    k = 25
    texts = [testrun["hashlist"], testrun["merge"], "ACGTACGTACGTACGTACGTACGTA\n",
             "1 2 3 TTTTTTTTTTTTTTTTTTTTTTTTT\n", "ACGTNCGTACGTACGTACGTACGTA 7\n\nA B C\n"]
    for single in (False, True):
        for t in texts:
            keys = capi.hashlist_keys(t.encode(), k, single)
            fs = oracle.FilterSet(t.encode(), single_end=single)
            assert len(set(keys.tolist())) == len(fs)
            # every key, converted back to RUFUS's little-endian A0 G1 C2 T3 word, is in the reference set
            for key in set(keys.tolist()):
                s = tools.keys_to_text(np.array([key], dtype=np.uint64), k)[0].encode()
                assert fs.scan(s + b"A", b"J" * 26, k, 0, single_end=False) == 1


def test_header_is_jellyfish_readable():
    cols = capi.jf_matrix(27, 25)
    h = capi.jhash_header(25, 27, cols, True, 4, ["jellyfish", "count", "-m", "25", "it's"])
    assert len(h) % 8 == 0 and h[:9].isdigit() and int(h[:9]) == len(h) - 9
    js = json.loads(h[9:].rstrip(b"\0"))
    assert list(js) == sorted(js)                      # Json::FastWriter order (std::map)
    assert js["matrix1"]["columns"] == [int(x) for x in cols] and js["matrix1"]["r"] == 27
    assert js["size"] == 1 << 27 and js["key_len"] == 50 and js["counter_len"] == 4 and js["val_len"] == 7
    assert js["max_reprobe"] == 126 and js["reprobes"][:4] == [1, 1, 3, 6] and js["cmdline"][-1] == "it's"
    ho, _ = oracle.parse_jhash(h)
    ref = json.loads(oracle.header_bytes(oracle.Records(25, 27, cols, np.zeros(0, np.uint64), np.zeros(0, np.uint64),
                                                        None))[9:].rstrip(b"\0"))
    for key in ("alignment", "canonical", "counter_len", "format", "key_len", "matrix1", "max_reprobe", "reprobes",
                "size", "val_len"):
        assert ho[key] == ref[key], key
    small = json.loads(capi.jhash_header(25, 10, capi.jf_matrix(10, 25))[9:].rstrip(b"\0"))
    assert small["max_reprobe"] == 44                  # reprobe_limit_t shrinks until its offset < size


def test_parse_sequences_like_jellyfish():
    fq = b"@a\nACGT\n+\nJJJJ\n@b\nAC\nGT\n+b\nJJ\nJJ\n"
    assert tools.parse_sequences(fq) == [b"ACGT", b"ACGT"]
    fa = b">x desc\nACGT\nAC\n>y\n\nGG\n"
    assert tools.parse_sequences(fa) == [b"ACGTAC", b"GG"]
    assert tools.parse_sequences(b"") == []
    with pytest.raises(ValueError):
        tools.parse_sequences(b"ACGT\n")
    assert tools.keys_to_text(np.array([tools.text_to_key("ACGTTGCA")], dtype=np.uint64), 8) == ["ACGTTGCA"]
