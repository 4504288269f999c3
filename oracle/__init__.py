"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the RUFUS hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package, and only as the checker.  The product (``rufus_amd``) never imports it.

The heavy loops live in ``rufus_oracle.cpp`` (built by ``make -C oracle``); the byte/integer glue
(file header, histogram, merge, query, hash list) is numpy / plain Python here.  Citations are
``path:line`` under ``/root/reference``; ``jf/`` = ``src/modifiedJellyfish/``.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build() -> str:
    """Compile the C++ restatement (idempotent)."""
    so = os.path.join(_HERE, "librufus_oracle.so")
    src = os.path.join(_HERE, "rufus_oracle.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "librufus_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        u64p = C.POINTER(C.c_uint64)
        L.orc_jf_matrix.argtypes = [C.c_int, C.c_int, u64p]
        L.orc_jf_times.argtypes = [u64p, C.c_int, C.c_uint64]
        L.orc_jf_times.restype = C.c_uint64
        L.orc_count_new.argtypes = [C.c_int, C.c_int]
        L.orc_count_new.restype = C.c_void_p
        L.orc_count_free.argtypes = [C.c_void_p]
        L.orc_count_add_read.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.orc_count_add_text.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.orc_count_add_text.restype = C.c_long
        L.orc_count_finish.argtypes = [C.c_void_p, C.c_int, u64p, C.c_uint64, C.c_uint64]
        L.orc_count_finish.restype = C.c_size_t
        L.orc_count_total.argtypes = [C.c_void_p]
        L.orc_count_total.restype = C.c_uint64
        L.orc_count_get.argtypes = [C.c_void_p, u64p, u64p, u64p]
        L.orc_merge_unique.restype = C.c_size_t
        L.orc_count_cas.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, u64p,
                                    C.c_uint64, C.c_uint64]
        L.orc_count_cas.restype = C.c_size_t
        L.orc_hash_to_long.argtypes = [C.c_char_p, C.c_size_t]
        L.orc_hash_to_long.restype = C.c_uint64
        L.orc_revcomp.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
        L.orc_revcomp.restype = C.c_size_t
        L.orc_filter_set_new.argtypes = [C.c_char_p, C.c_size_t, C.c_int]
        L.orc_filter_set_new.restype = C.c_void_p
        L.orc_filter_set_free.argtypes = [C.c_void_p]
        L.orc_filter_set_size.argtypes = [C.c_void_p]
        L.orc_filter_set_size.restype = C.c_size_t
        L.orc_filter_scan.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int, C.c_int,
                                      C.c_int]
        L.orc_filter_pairs.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int, C.c_int,
                                       C.c_int, C.POINTER(C.c_uint32), C.c_size_t]
        L.orc_filter_pairs.restype = C.c_long
        _LIB = L
    return _LIB


def _u64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


# ------------------------------------------------------------------------------------------------
# jellyfish k-mer text <-> key  (jf/include/jellyfish/mer_dna.hpp:46-63, :460-471)
# ------------------------------------------------------------------------------------------------
_JF = "ACGT"


def jf_encode(kmer: str) -> int:
    v = 0
    for ch in kmer:
        v = (v << 2) | _JF.index(ch.upper())
    return v


def jf_decode(key: int, k: int) -> str:
    return "".join(_JF[(int(key) >> (2 * (k - 1 - i))) & 3] for i in range(k))


def jf_revcomp_key(key: int, k: int) -> int:
    r = 0
    key = int(key)
    for _ in range(k):
        r = (r << 2) | (3 - (key & 3))
        key >>= 2
    return r


def jf_canonical(key: int, k: int) -> int:
    return min(int(key), jf_revcomp_key(key, k))


def ceil_log2(x: int) -> int:
    """jf/include/jellyfish/misc.hpp ceilLog2: table size is rounded up to a power of two."""
    return max(0, (int(x) - 1).bit_length())


def jf_matrix(lsize: int, k: int) -> np.ndarray:
    """Hash matrix of a (2^lsize)-slot table for k-mers (jf/.../large_hash_array.hpp:942-950)."""
    cols = np.zeros(2 * k, dtype=np.uint64)
    if lib().orc_jf_matrix(lsize, 2 * k, _u64p(cols)) != 0:
        raise ValueError("bad matrix size")
    return cols


def jf_pos(cols: np.ndarray, key: int, lsize: int) -> int:
    cols = np.ascontiguousarray(cols, dtype=np.uint64)
    return int(lib().orc_jf_times(_u64p(cols), len(cols), int(key))) & ((1 << lsize) - 1)


def multiset_checksum(keys, counts) -> tuple:
    with np.errstate(over="ignore"):
        x = np.asarray(keys, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
        a = (x * np.asarray(counts, dtype=np.uint64)).sum(dtype=np.uint64)
        b = x.sum(dtype=np.uint64)
    return int(a), int(b)


# ------------------------------------------------------------------------------------------------
# count  (jf/sub_commands/count_main.cc:148-180, :214-353)
# ------------------------------------------------------------------------------------------------
class Records:
    """Sorted binary/sorted payload: keys, counts (and pos) in (pos, key) order."""

    def __init__(self, k, lsize, cols, keys, counts, pos, canonical=True, total=0):
        self.k, self.lsize, self.cols = k, lsize, cols
        self.keys, self.counts, self.pos = keys, counts, pos
        self.canonical, self.total = canonical, total

    def payload(self, counter_len: int = 4) -> bytes:
        """jf/include/jellyfish/binary_dumper.hpp:44-48: ceil(2k/8) low bytes of the key word,
        then min(count, 2^(8*counter_len)-1) little endian."""
        kb = (2 * self.k + 7) // 8
        n = len(self.keys)
        out = np.zeros((n, kb + counter_len), dtype=np.uint8)
        kbytes = self.keys.astype("<u8").view(np.uint8).reshape(n, 8)
        out[:, :kb] = kbytes[:, :kb]
        cap = (1 << (8 * counter_len)) - 1
        v = np.minimum(self.counts, np.uint64(cap)).astype("<u8").view(np.uint8).reshape(n, 8)
        out[:, kb:] = v[:, :counter_len]
        return out.tobytes()

    def checksum(self) -> tuple:
        """include/rufus_hip.h rfx_records_checksum restated: (sum of mix(key) * count, sum of mix(key)) mod 2^64,
        mix = the splitmix64 finaliser.  (Not a reference function: the shard-independent fingerprint of the record
        multiset that the full-size runs compare between different cuts of the work.)"""
        return multiset_checksum(self.keys, self.counts)

    def dump_text(self) -> str:
        """``jellyfish dump -c``: ``KMER COUNT`` per record in file order."""
        return "".join(f"{jf_decode(k, self.k)} {int(c)}\n" for k, c in zip(self.keys, self.counts))


def count(texts, k: int, size: int, lower: int = 0, upper: int = 2**64 - 1, canonical: bool = True,
          reads=None) -> Records:
    """``jellyfish count -m k -s size [-C] [-L lower] [-U upper]`` over FASTA/FASTQ file contents
    (``texts``: iterable of bytes) and/or bare read sequences (``reads``: iterable of bytes)."""
    L = lib()
    h = L.orc_count_new(k, int(canonical))
    if not h:
        raise ValueError("k out of range")
    try:
        for t in texts or ():
            if L.orc_count_add_text(h, t, len(t)) < 0:
                raise ValueError("malformed sequence file")
        for r in reads or ():
            L.orc_count_add_read(h, r, len(r))
        lsize = ceil_log2(size)
        cols = jf_matrix(lsize, k)
        n = L.orc_count_finish(h, lsize, _u64p(cols), lower, upper)
        keys = np.zeros(n, dtype=np.uint64)
        vals = np.zeros(n, dtype=np.uint64)
        pos = np.zeros(n, dtype=np.uint64)
        L.orc_count_get(h, _u64p(keys), _u64p(vals), _u64p(pos))
        return Records(k, lsize, cols, keys, vals, pos, canonical, int(L.orc_count_total(h)))
    finally:
        L.orc_count_free(h)


def count_reads_matrix(seq: np.ndarray, k: int, size: int, lower: int = 0, upper: int = 2**64 - 1,
                       canonical: bool = True, threads: int = 1, table_bits: int = 0) -> Records:
    """The CPU-baseline count: all threads insert into ONE lock-free open-addressed table (the way
    ``jellyfish count -t`` works, ``orc_count_cas``).  ``seq``: uint8 matrix (n_reads, L) of bases.  Same
    records as :func:`count`."""
    L = lib()
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    n, rl = seq.shape
    if not table_bits:     # distinct k-mers <= windows; errors make ~15 % of the windows distinct at 30x
        table_bits = max(16, ceil_log2(max(1, n * max(rl - k + 1, 1)) // 2))
    h = L.orc_count_new(k, int(canonical))
    try:
        lsize = ceil_log2(size)
        cols = jf_matrix(lsize, k)
        while True:
            m = L.orc_count_cas(h, seq.ctypes.data, n, rl, table_bits, threads, lsize, _u64p(cols), lower, upper)
            if m != 2**64 - 1:
                break
            table_bits += 1
        keys, vals, pos = np.zeros(m, np.uint64), np.zeros(m, np.uint64), np.zeros(m, np.uint64)
        L.orc_count_get(h, _u64p(keys), _u64p(vals), _u64p(pos))
        return Records(k, lsize, cols, keys, vals, pos, canonical, int(L.orc_count_total(h)))
    finally:
        L.orc_count_free(h)


# ------------------------------------------------------------------------------------------------
# .Jhash container  (jf/include/jellyfish/generic_file_header.hpp:96-121, file_header.hpp:33-110)
# ------------------------------------------------------------------------------------------------
# jf/include/jellyfish/large_hash_array.hpp quadratic_reprobes: i*(i+1)/2
def quadratic_reprobes(n: int):
    return [1 if i == 0 else i * (i + 1) // 2 for i in range(n + 1)]


def header_bytes(rec: Records, counter_len: int = 4, val_len: int = 7, max_reprobe: int = 126, extra=None) -> bytes:
    root = {
        "alignment": 8,
        "canonical": bool(rec.canonical),
        "cmdline": [],
        "counter_len": counter_len,
        "exe_path": "",
        "format": "binary/sorted",
        "hostname": "",
        "key_len": 2 * rec.k,
        "matrix1": {"c": 2 * rec.k, "columns": [int(x) for x in rec.cols], "r": rec.lsize},
        "max_reprobe": max_reprobe,
        "pwd": "",
        "reprobes": quadratic_reprobes(max_reprobe),
        "size": 1 << rec.lsize,
        "time": "",
        "val_len": val_len,
    }
    if extra:
        root.update(extra)
    js = json.dumps(root, sort_keys=True, separators=(",", ":")).encode()
    hlen = len(js)
    pad = (9 + len(js)) % 8
    if pad:
        hlen += 8 - pad
    return b"%09d" % hlen + js + b"\0" * (hlen - len(js))


def parse_jhash(blob: bytes):
    """-> (header dict, payload bytes).  generic_file_header.hpp:123-150."""
    hlen = int(blob[:9])
    js = blob[9:9 + hlen].rstrip(b"\0")
    return json.loads(js), blob[9 + hlen:]


def records_from_payload(hdr: dict, payload: bytes) -> Records:
    k = hdr["key_len"] // 2
    kb = (hdr["key_len"] + 7) // 8
    cl = hdr["counter_len"]
    n = len(payload) // (kb + cl)
    raw = np.frombuffer(payload, dtype=np.uint8).reshape(n, kb + cl)
    kk = np.zeros((n, 8), dtype=np.uint8)
    kk[:, :kb] = raw[:, :kb]
    cc = np.zeros((n, 8), dtype=np.uint8)
    cc[:, :cl] = raw[:, kb:]
    keys = kk.view("<u8").reshape(n).astype(np.uint64)
    cnts = cc.view("<u8").reshape(n).astype(np.uint64)
    cols = np.array(hdr["matrix1"]["columns"], dtype=np.uint64)
    lsize = ceil_log2(hdr["size"])
    pos = np.array([jf_pos(cols, int(x), lsize) for x in keys], dtype=np.uint64) if n <= 200000 else None
    return Records(k, lsize, cols, keys, cnts, pos, hdr.get("canonical", False))


# ------------------------------------------------------------------------------------------------
# histo  (jf/sub_commands/histo_main.cc:33-89; defaults low=1 high=10000 inc=1)
# ------------------------------------------------------------------------------------------------
def histo(counts: np.ndarray, low: int = 1, high: int = 10000, inc: int = 1, full: bool = False):
    base = 0 if inc >= low else low - inc
    ceil = high + inc
    nb = (ceil + inc - base) // inc
    h = np.zeros(nb, dtype=np.uint64)
    c = np.asarray(counts, dtype=np.uint64)
    idx = np.where(c < base, 0, np.where(c > ceil, nb - 1, (np.maximum(c, base) - base) // inc)).astype(np.int64)
    np.add.at(h, idx, 1)
    rows = [(base + i * inc, int(h[i])) for i in range(nb) if full or h[i] > 0]
    return h, "".join(f"{a} {b}\n" for a, b in rows)


# ------------------------------------------------------------------------------------------------
# set difference
# ------------------------------------------------------------------------------------------------
def stats_text(counts: np.ndarray, low: int = 0, high: int = 2**64 - 1) -> str:
    """``jellyfish stats`` (jf/sub_commands/stats_main.cc:33-46,:73-77): Unique / Distinct / Total / Max_count of the
    records with low <= count <= high."""
    c = np.asarray(counts, dtype=np.uint64)
    c = c[(c >= np.uint64(low)) & (c <= np.uint64(high))]
    return (f"Unique:    {int((c == 1).sum())}\nDistinct:  {len(c)}\nTotal:     {int(c.sum())}\n"
            f"Max_count: {int(c.max()) if len(c) else 0}\n")


def merge_unique(files, min_count: int = 5, with_file: bool = False):
    """RUFUS's modified ``jellyfish merge`` (jf/jellyfish/merge_files.cc:69-155): k-way merge in
    (pos, key) order; a key present in exactly one input with count >= 5 is printed ``KMER\\tCOUNT``.
    ``files``: list of Records sharing k / lsize / matrix (:193-203)."""
    f0 = files[0]
    for f in files[1:]:
        if f.k != f0.k or f.lsize != f0.lsize or not np.array_equal(f.cols, f0.cols):
            raise ValueError("Can't merge hash with different hash function")
    nf = len(files)
    P = C.POINTER(C.c_uint64)
    arr = lambda xs: (P * nf)(*[_u64p(np.ascontiguousarray(x, dtype=np.uint64)) for x in xs])
    keep = [[np.ascontiguousarray(getattr(f, a), dtype=np.uint64) for f in files] for a in ("keys", "counts", "pos")]
    ns = (C.c_size_t * nf)(*[len(f.keys) for f in files])
    tot = sum(len(f.keys) for f in files)
    ok, ov = np.zeros(max(tot, 1), np.uint64), np.zeros(max(tot, 1), np.uint64)
    of = np.zeros(max(tot, 1), np.int32)
    n = lib().orc_merge_unique(nf, (P * nf)(*[_u64p(x) for x in keep[0]]), (P * nf)(*[_u64p(x) for x in keep[1]]),
                               (P * nf)(*[_u64p(x) for x in keep[2]]), ns, C.c_uint64(min_count), _u64p(ok),
                               _u64p(ov), of.ctypes.data_as(C.POINTER(C.c_int)))
    if with_file:
        return ok[:n].tolist(), ov[:n].tolist(), of[:n].tolist()
    return list(zip(ok[:n].tolist(), ov[:n].tolist()))


def merge_unique_text(files, min_count: int = 5) -> str:
    k = files[0].k
    return "".join(f"{jf_decode(key, k)}\t{cnt}\n" for key, cnt in merge_unique(files, min_count))


def query(rec: Records, kmers):
    """``jellyfish query -s`` (jf/sub_commands/query_main.cc:44-51): every k-mer of the query
    sequences, canonicalised when the database header says so (:115), printed ``KMER COUNT``."""
    table = dict(zip(rec.keys.tolist(), rec.counts.tolist()))
    out = []
    for km in kmers:
        key = jf_encode(km)
        if rec.canonical:
            key = jf_canonical(key, rec.k)
        out.append((key, table.get(key, 0)))
    return out


def hash_list(subject: Records, others, min_cov: int, max_cov: int) -> str:
    """Net effect of runRufus.sh:925-926 + scripts/CheckJellyHashList.sh:12: merge-unique over
    [subject]+others, re-query in the subject (keys unique to another input read 0 there), keep
    min_cov <= count <= max_cov; ``KMER COUNT`` in merge order."""
    keys, vals, which = merge_unique([subject] + list(others), with_file=True)
    out = []
    for key, v, f in zip(keys, vals, which):
        c = v if f == 0 else 0          # query of the subject database
        if min_cov <= c <= max_cov:
            out.append(f"{jf_decode(key, subject.k)} {c}\n")
    return "".join(out)


# ------------------------------------------------------------------------------------------------
# filter  (src/RUFUS.Filter.cpp, src/RUFUS.Filter.ss.cpp)
# ------------------------------------------------------------------------------------------------
class FilterSet:
    def __init__(self, hashlist_text: bytes, single_end: bool = False):
        self._h = lib().orc_filter_set_new(hashlist_text, len(hashlist_text), int(single_end))

    def __len__(self):
        return lib().orc_filter_set_size(self._h)

    def scan(self, seq: bytes, qual: bytes, k: int, minq: int, single_end: bool = False) -> int:
        return lib().orc_filter_scan(self._h, seq, len(seq), qual, len(qual), k, minq, int(single_end))

    def pairs(self, mate1: bytes, mate2: bytes, k: int, minq: int, thresh: int) -> np.ndarray:
        cap = mate1.count(b"\n") // 4 + 2
        out = np.zeros(cap, dtype=np.uint32)
        n = lib().orc_filter_pairs(self._h, mate1, len(mate1), mate2, len(mate2), k, minq, thresh,
                                   out.ctypes.data_as(C.POINTER(C.c_uint32)), cap)
        return out[:n].copy()

    def __del__(self):
        try:
            lib().orc_filter_set_free(self._h)
        except Exception:
            pass


def hash_to_long(kmer: bytes) -> int:
    return int(lib().orc_hash_to_long(kmer, len(kmer)))
