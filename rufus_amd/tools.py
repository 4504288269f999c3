"""Host-side mirror of the reference's command-line operators for the hot path, on top of the C-ABI.

Each function keeps the name, argument meaning and file formats of the tool it mirrors:

=====================  ==========================================================================
``jellyfish_count``    ``jellyfish count -m K -s SIZE [-C] [-L n] [-U n] -o OUT IN...``
                       (scripts/RunJellyForRUFUS.sh:29)
``jellyfish_histo``    ``jellyfish histo [-f] DB``  (scripts/RunJellyForRUFUS.sh:37)
``jellyfish_dump``     ``jellyfish dump -c DB``  (scripts/Overlap.shorter.sh:247)
``jellyfish_query``    ``jellyfish query -s FASTA DB``  (scripts/CheckJellyHashList.sh:12)
``rufus_merge``        RUFUS's modified ``jellyfish merge F1 F2 ...``  (runRufus.sh:925)
``check_jelly_hash_list``  scripts/CheckJellyHashList.sh:12  (query + MinCov/MaxCov awk filters)
``rufus_filter``       ``RUFUS.Filter HashList M1 M2 STUB K MinQ Thresh Threads``  (runRufus.sh:967)
``rufus_filter_single``  ``RUFUS.Filter.single HashList FQ STUB K MinQ Thresh Threads``
=====================  ==========================================================================

Text parsing and file I/O happen here on the host; all arithmetic (k-mer extraction, hashing,
counting, sorting, set difference, read scan) runs in the HIP kernels behind ``rufus_amd.capi``.
"""
from __future__ import annotations

import json

import numpy as np

from . import capi

_CODES = np.full(256, 255, dtype=np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _CODES[_c] = _i
_LETTERS = np.frombuffer(b"ACGT", dtype=np.uint8)


# ---------------------------------------------------------------------------------------------------
# k-mer text <-> jellyfish key (jf/include/jellyfish/mer_dna.hpp: first base most significant)
# ---------------------------------------------------------------------------------------------------
def keys_to_text(keys: np.ndarray, k: int) -> list:
    keys = np.asarray(keys, dtype=np.uint64)
    shifts = (2 * (k - 1 - np.arange(k))).astype(np.uint64)
    codes = ((keys[:, None] >> shifts[None, :]) & np.uint64(3)).astype(np.intp)
    rows = _LETTERS[codes]
    return [r.tobytes().decode() for r in rows]


def text_to_key(kmer: str) -> int:
    v = 0
    for ch in kmer.upper().encode():
        c = int(_CODES[ch])
        if c > 3:
            raise ValueError(f"Invalid mer '{kmer}'")
        v = (v << 2) | c
    return v


def revcomp_key(key: int, k: int) -> int:
    r = 0
    for _ in range(k):
        r = (r << 2) | (3 - (key & 3))
        key >>= 2
    return r


# ---------------------------------------------------------------------------------------------------
# sequence files
# ---------------------------------------------------------------------------------------------------
def parse_sequences(data: bytes) -> list:
    """Sequences of a FASTA/FASTQ file the way jellyfish's parser sees them
    (jf/include/jellyfish/mer_overlap_sequence_parser.hpp:124-251): type sniffed from the first byte,
    multi-line records joined, qualities skipped by length."""
    if not data:
        return []
    lines = data.split(b"\n")
    if data[:1] == b">":
        out, cur = [], None
        for ln in lines:
            if ln[:1] == b">":
                if cur is not None:
                    out.append(b"".join(cur))
                cur = []
            elif cur is not None:
                cur.append(ln)
        if cur is not None:
            out.append(b"".join(cur))
        return out
    if data[:1] != b"@":
        raise ValueError("Unsupported format")
    # fast path: strict 4-line records
    n4 = len(lines) - (1 if lines[-1] == b"" else 0)
    if n4 % 4 == 0 and all(l[:1] == b"+" for l in lines[2:n4:4][:64]):
        seqs, quals, plus = lines[1:n4:4], lines[3:n4:4], lines[2:n4:4]
        if all(p[:1] == b"+" for p in plus) and all(len(s) == len(q) for s, q in zip(seqs, quals)):
            return seqs
    out, i, n = [], 0, len(lines)
    while i < n:
        if lines[i] == b"":
            i += 1
            continue
        if lines[i][:1] != b"@":
            raise ValueError("Invalid fastq sequence")
        i += 1
        seq = []
        while i < n and lines[i][:1] != b"+":
            seq.append(lines[i])
            i += 1
        s = b"".join(seq)
        i += 1
        q = 0
        while i < n and q < len(s):
            q += len(lines[i])
            i += 1
        if q != len(s):
            raise ValueError("Invalid fastq sequence")
        out.append(s)
    return out


def parse_fastq4(data: bytes):
    """Strict 4-line FASTQ as RUFUS.Filter reads it (src/RUFUS.Filter.cpp:162-175): (headers, seqs, plus, quals)."""
    lines = data.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    n = len(lines) // 4 * 4
    return lines[0:n:4], lines[1:n:4], lines[2:n:4], lines[3:n:4]


# ---------------------------------------------------------------------------------------------------
# .Jhash files
# ---------------------------------------------------------------------------------------------------
class JhashFile:
    """A ``binary/sorted`` jellyfish database held as device records plus its header facts."""

    def __init__(self, records: capi.Records, cols: np.ndarray, canonical: bool, counter_len: int = 4):
        self.records, self.cols, self.canonical, self.counter_len = records, cols, canonical, counter_len

    @property
    def k(self):
        return self.records.k

    @property
    def lsize(self):
        return self.records.lsize

    def write(self, path: str, argv=()):
        hdr = capi.jhash_header(self.k, self.lsize, self.cols, self.canonical, self.counter_len, argv)
        with open(path, "wb") as f:
            f.write(hdr)
            f.write(self.records.payload(self.counter_len))

    @classmethod
    def read(cls, ctx: capi.Context, path: str) -> "JhashFile":
        blob = open(path, "rb").read()
        digits = blob[:9]
        if not digits.isdigit() or blob[9:10] != b"{":
            raise ValueError(f"Failed to parse header of file '{path}'")
        hlen = int(digits)
        hdr = json.loads(blob[9:9 + hlen].rstrip(b"\0"))
        if hdr.get("format") != "binary/sorted":
            raise ValueError(f"Unsupported format '{hdr.get('format')}'")
        k = hdr["key_len"] // 2
        cols = np.array(hdr["matrix1"]["columns"], dtype=np.uint64)
        lsize = capi.ceil_log2(hdr["size"])
        rec = capi.Records.load(ctx, k, lsize, cols, blob[9 + hlen:], hdr["counter_len"])
        return cls(rec, cols, bool(hdr.get("canonical", False)), hdr["counter_len"])


def jellyfish_count(ctx: capi.Context, inputs, k: int, size: int, canonical: bool = True, lower: int = 0,
                    upper: int = 2**64 - 1, out: str | None = None, capacity: int = 0, argv=(),
                    mode: int = capi.COUNT_AUTO) -> JhashFile:
    """Count the k-mers of FASTA/FASTQ files (paths or bytes)."""
    table = capi.CountTable(ctx, k, size, canonical, capacity, mode=mode)
    try:
        for src in inputs:
            data = src if isinstance(src, (bytes, bytearray)) else open(src, "rb").read()
            seqs = parse_sequences(bytes(data))
            block = ctx.upload(capi.PackedReads.from_reads(seqs, flags=capi.PACK_COUNT))
            try:
                table.add(block)
            finally:
                block.free()
        rec = table.finish(lower, upper)
    finally:
        table.free()
    jf = JhashFile(rec, capi.jf_matrix(capi.ceil_log2(size), k), canonical)
    if out:
        jf.write(out, argv)
    return jf


def histo_text(h: np.ndarray, full: bool = False) -> str:
    """jf/sub_commands/histo_main.cc:82-84: ``count n`` rows, empty bins only with -f."""
    return "".join(f"{i} {int(v)}\n" for i, v in enumerate(h) if full or v > 0)


def jellyfish_histo(db: JhashFile, full: bool = False) -> str:
    return histo_text(db.records.histo(), full)


def jellyfish_dump(db: JhashFile) -> str:
    keys, counts, _ = db.records.get()
    return "".join(f"{t} {int(c)}\n" for t, c in zip(keys_to_text(keys, db.k), counts))


def jellyfish_query(db: JhashFile, kmers) -> str:
    """``KMER COUNT`` per query k-mer; canonicalised when the database is (query_main.cc:115)."""
    keys = []
    for km in kmers:
        key = text_to_key(km)
        if db.canonical:
            key = min(key, revcomp_key(key, db.k))
        keys.append(key)
    keys = np.array(keys, dtype=np.uint64)
    counts = db.records.query(keys)
    return "".join(f"{t} {int(c)}\n" for t, c in zip(keys_to_text(keys, db.k), counts))


def rufus_merge(ctx: capi.Context, files) -> str:
    """stdout of RUFUS's modified ``jellyfish merge``: ``KMER\\tCOUNT`` for keys in exactly one input, count >= 5."""
    keys, counts = capi.merge_unique(ctx, [f.records for f in files], 5)
    return "".join(f"{t}\t{int(c)}\n" for t, c in zip(keys_to_text(keys, files[0].k), counts))


def check_jelly_hash_list(db: JhashFile, merge_text: str, min_cov: int, max_cov: int) -> str:
    kmers = [ln.split()[0] for ln in merge_text.splitlines() if ln.strip()]
    out = []
    for ln in jellyfish_query(db, kmers).splitlines():
        c = int(ln.split()[1])
        if min_cov <= c <= max_cov:
            out.append(ln + "\n")
    return "".join(out)


def hash_list(ctx: capi.Context, subject: JhashFile, others, min_cov: int, max_cov: int) -> str:
    """Fused runRufus.sh:925-926: same text as rufus_merge | check_jelly_hash_list, one device pass."""
    keys, counts = capi.unique_to_subject(ctx, subject.records, [o.records for o in others], min_cov, max_cov)
    return "".join(f"{t} {int(c)}\n" for t, c in zip(keys_to_text(keys, subject.k), counts))


# ---------------------------------------------------------------------------------------------------
# RUFUS.Filter
# ---------------------------------------------------------------------------------------------------
def _mask_bits(mask: np.ndarray, n: int) -> np.ndarray:
    bits = np.unpackbits(mask.view(np.uint8), bitorder="little")[:n]
    return bits.astype(bool)


def rufus_filter(ctx: capi.Context, hashlist: str, mate1: str, mate2: str, stub: str, k: int, min_q: int, thresh: int,
                 threads: int = 1) -> int:
    """Writes ``STUB.Mutations.Mate1.fastq`` / ``Mate2``; returns the number of pulled pairs.  Pairs
    are written in input order (the reference's order depends on OpenMP scheduling)."""
    keys = capi.hashlist_keys(open(hashlist, "rb").read(), k, single_end=False)
    h1, s1, p1, q1 = parse_fastq4(open(mate1, "rb").read())
    h2, s2, p2, q2 = parse_fastq4(open(mate2, "rb").read())
    if any(len(s) == 0 for s in s1) or any(len(s) == 0 for s in s2):
        raise ValueError("empty sequence line (undefined behaviour in the reference; rejected)")
    n = len(s1)
    mset = capi.MutantSet(ctx, keys, k)
    pulled = np.zeros(n, dtype=bool)
    try:
        for seqs, quals in ((s1, q1), (s2[:n], q2[:n])):
            blk = ctx.upload(capi.PackedReads.from_reads(seqs, quals, min_q, capi.PACK_FILTER))
            try:
                _, mask, _ = mset.filter(blk, thresh, last_base_skipped=True, want_hits=False)
            finally:
                blk.free()
            pulled[:len(seqs)] |= _mask_bits(mask, len(seqs))
    finally:
        mset.free()
    with open(stub + ".Mutations.Mate1.fastq", "wb") as f1, open(stub + ".Mutations.Mate2.fastq", "wb") as f2:
        for i in np.flatnonzero(pulled):
            f1.write(h1[i] + b"\n" + s1[i] + b"\n" + p1[i] + b"\n" + q1[i] + b"\n")
            if i < len(s2):
                f2.write(h2[i] + b"\n" + s2[i] + b"\n" + p2[i] + b"\n" + q2[i] + b"\n")
    return int(pulled.sum())


def rufus_filter_single(ctx: capi.Context, hashlist: str, fastq: str, stub: str, k: int, min_q: int, thresh: int,
                        threads: int = 1) -> int:
    """``STUB.Mutations.fastq`` with ``:MH<hits>`` appended to each header (src/RUFUS.Filter.ss.cpp:198)."""
    keys = capi.hashlist_keys(open(hashlist, "rb").read(), k, single_end=True)
    h, s, p, q = parse_fastq4(open(fastq, "rb").read())
    mset = capi.MutantSet(ctx, keys, k)
    try:
        blk = ctx.upload(capi.PackedReads.from_reads(s, q, min_q, capi.PACK_FILTER))
        try:
            hits, mask, _ = mset.filter(blk, thresh, last_base_skipped=False)
        finally:
            blk.free()
    finally:
        mset.free()
    pulled = _mask_bits(mask, len(s))
    with open(stub + ".Mutations.fastq", "wb") as f:
        for i in np.flatnonzero(pulled):
            f.write(h[i] + b":MH%d\n" % int(hits[i]) + s[i] + b"\n" + p[i] + b"\n" + q[i] + b"\n")
    return int(pulled.sum())
