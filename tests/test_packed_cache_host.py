"""CPU-only: the packed-read cache `jellyfish count --sam .. --keep-packed FILE` leaves (SURVEY 8(f) row N2,
rufus_amd/csrc/host/rfx_packed_cache.hpp), produced by the tool's own main() over the host stand-ins of
tests/host/jellyfish_harness.cpp and read back here field by field: every record's line span in the stream, its name
hash, the bases and the "good" mask exactly as src/PassThroughSamCheck.stranded.cpp:188-223 would print them and
src/RUFUS.Filter.cpp:205 would judge them (reverse-strand records reverse-complemented, qualities reversed), records
that cannot be packed exactly flagged for the text route, the chromosome runs of the feeder's log.
tests/test_cli_gpu.py::test_subject_is_parsed_once_and_filtered_from_the_packed_cache runs the consumer on a GPU."""
import os
import struct
import subprocess

import numpy as np
import pytest

from tests.test_cli_host import make_sam
from tests.test_jellyfish_host import _build, _payload, sh

FILE_MAGIC = 0x31484341434B5052
CHUNK_MAGIC = 0x314B4E5548434B50
DONE_MAGIC = 0x454E4F4448434143      # "CACHDONE": written last, with the chunk count and the stream's length
COMP = bytes.maketrans(b"ACGTN", b"TGCAN")


@pytest.fixture(scope="module")
def jf(tmp_path_factory):
    return _build(str(tmp_path_factory.mktemp("jfk") / "jellyfish"), ["-O2"])


def pad8(x):
    return (x + 7) & ~7


def read_cache(blob):
    """-> (min_q, [chunk dict]) in stream order; the layout of rfx_packed_cache.hpp:36-53."""
    magic, min_q, _, done, n_chunks, stream_bytes = struct.unpack_from("<QiIQQQ", blob, 0)
    assert magic == FILE_MAGIC and done == DONE_MAGIC
    at, chunks = 64, []
    while at + 48 <= len(blob):
        magic, seq, stream_off, nbytes, n, n_words, runs_bytes, _ = struct.unpack_from("<QQQQIIII", blob, at)
        if magic != CHUNK_MAGIC:
            break
        o = at + 48
        c = {"seq": seq, "stream_off": stream_off, "n": n}
        for name, dt, cnt in (("hash", "<u8", n), ("line_off", "<u4", n), ("line_len", "<u4", n), ("flags", "u1", n),
                              ("len", "<u4", n), ("word_off", "<u4", n + 1), ("codes", "<u8", n_words), ("good", "<u4", n_words)):
            c[name] = np.frombuffer(blob, dtype=dt, count=cnt, offset=o)
            o += pad8(cnt * np.dtype(dt).itemsize)
        c["runs"] = blob[o:o + runs_bytes]
        o += pad8(runs_bytes)
        assert o == at + nbytes
        chunks.append(c)
        at += nbytes
    assert not any(blob[at:]), "bytes behind the last chunk"
    chunks.sort(key=lambda c: c["seq"])
    assert [c["seq"] for c in chunks] == list(range(len(chunks))) and len(chunks) == n_chunks
    read_cache.stream_bytes = stream_bytes
    return min_q, chunks


def unpack(codes, good, length):
    seq = bytes(b"ACGT"[(int(codes[i // 32]) >> (2 * (i % 32))) & 3] for i in range(length))
    ok = [(int(good[i // 32]) >> (i % 32)) & 1 for i in range(length)]
    return seq, ok


@pytest.mark.parametrize("min_q,piece", [(15, "30000"), (3, "7000")])
def test_keep_packed_holds_every_record_as_the_filter_would_see_it(jf, tmp_path, min_q, piece):
    d = str(tmp_path)
    f = [ln.split(b"\t") for ln in make_sam(1200, seed=5).split(b"\n") if ln]
    rev = [j for j in range(len(f)) if len(f[j]) > 10 and int(f[j][1]) & 16]
    fwd = [j for j in range(len(f)) if len(f[j]) > 10 and not int(f[j][1]) & 16]
    f[rev[3]][9] = f[rev[3]][9][:70] + b"R" + f[rev[3]][9][71:]          # a base the feeder's reverse complement drops
    f[fwd[5]][9] = f[fwd[5]][9][:10] + b"n" + f[fwd[5]][9][11:]          # lower case: not what HashToLong packs
    f[fwd[7]][10] = f[fwd[7]][10][:90]                                     # fewer qualities than bases
    f[fwd[9]][9] = f[fwd[9]][9][:20] + b"N" + f[fwd[9]][9][21:]          # N is plain: packed, never good
    f.append(list(f[fwd[11]]))                                             # a name a third time
    sam = b"".join(b"\t".join(x) + b"\n" for x in f)
    open(f"{d}/in.sam", "wb").write(sam)
    cmd = [jf, "count", "--disk", "-m", "25", "-L", "2", "-s", "100M", "-t", "4", "-C"]
    r = sh(cmd + ["--sam", "a.chr", "--keep-packed", "cache.bin", "--keep-minq", str(min_q), "-o", "a.Jhash", "in.sam"], d,
           env={"RFX_INGEST_PIECE": piece})
    assert r.returncode == 0, r.stderr
    r2 = sh(cmd + ["--sam", "b.chr", "-o", "b.Jhash", "in.sam"], d)
    assert r2.returncode == 0 and _payload(f"{d}/a.Jhash") == _payload(f"{d}/b.Jhash")
    assert open(f"{d}/a.chr", "rb").read() == open(f"{d}/b.chr", "rb").read()

    got_q, chunks = read_cache(open(f"{d}/cache.bin", "rb").read())
    assert got_q == min_q and len(chunks) >= 4
    assert read_cache.stream_bytes == len(sam)          # the header's last word: how long a stream the chunks describe
    recs = [x for x in f if len(x) > 10]
    i = 0
    by_name = {}
    runs = b""
    always = 0
    for c in chunks:
        runs += c["runs"]
        for j in range(c["n"]):
            x = recs[i]
            i += 1
            a = c["stream_off"] + int(c["line_off"][j])
            line = sam[a:a + int(c["line_len"][j])]
            assert line.rstrip(b"\n") == b"\t".join(x)
            by_name.setdefault(x[0], set()).add(int(c["hash"][j]))
            plain = set(x[9]) <= set(b"ACGTN") and len(x[9]) == len(x[10]) and len(x[9]) > 0
            assert bool(c["flags"][j] & 1) == (not plain)
            if not plain:
                always += 1
                assert c["len"][j] == 0 and c["word_off"][j + 1] == c["word_off"][j]
                continue
            seq, qual = x[9], x[10]
            if int(x[1]) & 16:
                seq, qual = seq.translate(COMP)[::-1], qual[::-1]
            assert c["len"][j] == len(seq) and c["word_off"][j + 1] - c["word_off"][j] == (len(seq) + 31) // 32
            w0, w1 = int(c["word_off"][j]), int(c["word_off"][j + 1])
            got_seq, got_ok = unpack(c["codes"][w0:w1], c["good"][w0:w1], len(seq))
            assert got_seq == seq.replace(b"N", b"A")                                    # src/Util.cpp:51-84: not ACGT -> A
            assert got_ok == [int(b != 78 and q - 33 >= min_q) for b, q in zip(seq, qual)]   # src/RUFUS.Filter.cpp:205
    assert i == len(recs) and always == 3
    assert all(len(h) == 1 for h in by_name.values())                                   # one hash per name ..
    assert len({next(iter(h)) for h in by_name.values()}) == len(by_name)               # .. and no collision on 600 names
    # the runs of the chromosome log: the names in stream order, a new line where the name changes
    want = []
    for x in recs:
        if not want or want[-1] != x[2]:
            want.append(x[2])
    got = [r for r in runs.split(b"\n") if r]
    merged = [r for k, r in enumerate(got) if k == 0 or got[k - 1] != r]                # a run cut by a piece boundary comes twice
    assert merged == want


def test_keep_packed_is_refused_where_it_cannot_work(jf, tmp_path):
    d = str(tmp_path)
    open(f"{d}/in.sam", "wb").write(make_sam(50, seed=1))
    open(f"{d}/in.fq", "wb").write(b"@r\nACGTACGTACGTACGTACGTACGTACGTACGT\n+\n" + b"I" * 32 + b"\n")
    cmd = [jf, "count", "--disk", "-m", "25", "-L", "2", "-s", "100M", "-t", "2", "-C", "-o", "x.Jhash"]
    r = sh(cmd + ["--keep-packed", "c.bin", "in.fq"], d)                                 # without --sam
    assert r.returncode != 0 and b"--keep-packed" in r.stderr
    r = sh(cmd + ["--sam", "x.chr", "--keep-packed", "c.bin", "in.sam", "in.sam"], d)    # two inputs
    assert r.returncode != 0 and b"--keep-packed" in r.stderr
    r = subprocess.run(cmd + ["--sam", "x.chr", "--keep-packed", "c.bin", "/dev/stdin"], cwd=d,   # a pipe without --spool:
                       input=open(f"{d}/in.sam", "rb").read(), stdout=subprocess.PIPE, stderr=subprocess.PIPE)   # the cache would
    assert r.returncode != 0 and b"--spool" in r.stderr                                   # point into nothing
