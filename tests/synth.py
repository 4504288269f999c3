"""Deterministic synthetic trio generator: random genome, child = genome + planted
heterozygous SNVs, parents = genome; paired 150 bp reads, insert U[250,400], 0.5 % substitution
errors, quality 'J' except 2 % of bases at '#', 0.1 % 'N'.  Shared by tests, smoke() and bench.py.
"""
from __future__ import annotations

import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGTN", b"TGCAN"):
    _COMP[a] = b


class Sample:
    """Mate matrices: seq/qual are uint8 arrays of shape (n_pairs, L) per mate."""

    def __init__(self, name, s1, q1, s2, q2, pos1=None, pos2=None):
        self.name, self.s = name, (s1, s2)
        self.q = (q1, q2)
        self.pos = (pos1, pos2)    # 0-based leftmost genome coordinate of each mate

    def __len__(self):
        return self.s[0].shape[0]


def _reads_from(genome: np.ndarray, alt: np.ndarray | None, n_pairs: int, L: int, rng, err, lowq, nrate):
    G = len(genome)
    insert = rng.integers(250, 401, n_pairs)
    start = rng.integers(0, G - 400, n_pairs)
    idx1 = start[:, None] + np.arange(L)[None, :]
    idx2 = (start + insert - L)[:, None] + np.arange(L)[None, :]
    if alt is not None:
        hap = rng.random(n_pairs) < 0.5
        src1 = np.where(hap[:, None], alt[idx1], genome[idx1])
        src2 = np.where(hap[:, None], alt[idx2], genome[idx2])
    else:
        src1, src2 = genome[idx1], genome[idx2]
    s1 = src1.copy()
    s2 = _COMP[src2[:, ::-1]]
    out = []
    for s in (s1, s2):
        e = rng.random(s.shape) < err
        # substitute with one of the three other bases
        code = np.searchsorted(ACGT, s[e])
        s[e] = ACGT[(code + rng.integers(1, 4, code.shape)) & 3]
        s[rng.random(s.shape) < nrate] = ord("N")
        q = np.full(s.shape, ord("J"), dtype=np.uint8)
        q[rng.random(s.shape) < lowq] = ord("#")
        out.append((s, q))
    return out[0][0], out[0][1], out[1][0], out[1][1], start, start + insert - L


def make_trio(genome_len=5_000_000, n_pairs=500_000, n_snv=20, seed=12345, L=150, err=0.005, lowq=0.02,
              nrate=0.001, read_seed=None):
    """``seed`` fixes the genome and the planted SNVs; ``read_seed`` (default: continue the same
    stream) draws the reads, so several read blocks of one trio can be generated independently."""
    rng = np.random.default_rng(seed)
    genome = ACGT[rng.integers(0, 4, genome_len)]
    alt = genome.copy()
    pos = np.sort(rng.choice(np.arange(1000, genome_len - 1000), n_snv, replace=False))
    alt[pos] = ACGT[(np.searchsorted(ACGT, genome[pos]) + rng.integers(1, 4, n_snv)) & 3]
    if read_seed is not None:
        rng = np.random.default_rng(read_seed)
    trio = {}
    for name, a in (("child", alt), ("mother", None), ("father", None)):
        trio[name] = Sample(name, *_reads_from(genome, a, n_pairs, L, rng, err, lowq, nrate))
    trio["snv_pos"] = pos
    return trio


def fastq_bytes(sample: Sample, mate: int) -> bytes:
    """4-line FASTQ text of one mate file; headers ``@<name>.<index>``, fixed width."""
    s, q = sample.s[mate - 1], sample.q[mate - 1]
    n, L = s.shape
    tag = ("@" + sample.name[:1]).encode()
    W = len(tag) + 9
    rec = np.empty((n, W + 1 + L + 1 + 2 + L + 1), dtype=np.uint8)
    rec[:, :len(tag)] = np.frombuffer(tag, dtype=np.uint8)
    idx = np.arange(n)
    for d in range(9):
        rec[:, len(tag) + 8 - d] = (idx // 10**d) % 10 + ord("0")
    rec[:, W] = ord("\n")
    rec[:, W + 1:W + 1 + L] = s
    rec[:, W + 1 + L] = ord("\n")
    rec[:, W + 2 + L] = ord("+")
    rec[:, W + 3 + L] = ord("\n")
    rec[:, W + 4 + L:W + 4 + 2 * L] = q
    rec[:, W + 4 + 2 * L] = ord("\n")
    return rec.tobytes()


def flat_reads(sample: Sample, mates=(1, 2)):
    """Concatenated bases/qualities + offsets of the selected mate files, in file order
    (all of mate 1, then all of mate 2) -- the order the FASTQ route feeds jellyfish."""
    seq = b"".join(sample.s[m - 1].tobytes() for m in mates)
    qual = b"".join(sample.q[m - 1].tobytes() for m in mates)
    n = sum(sample.s[m - 1].shape[0] for m in mates)
    L = sample.s[0].shape[1]
    off = (np.arange(n + 1, dtype=np.uint64) * np.uint64(L))
    return seq, qual, off


# ---------------------------------------------------------------------------------------------------
# Counter-based workload (include/rufus_hip.h `rfx_synth`, rufus_amd/csrc/rfx_synth.h): an independent
# numpy restatement of the generator, used to pin the C++ host twin and the device kernel.
# ---------------------------------------------------------------------------------------------------
_U = np.uint64
_PHI, _STEP = _U(0x9E3779B97F4A7C15), _U(0xD6E8FEB86659FD93)


def _mix64(z):
    with np.errstate(over="ignore"):
        z = np.asarray(z, dtype=np.uint64) + _PHI
        z = (z ^ (z >> _U(30))) * _U(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> _U(27))) * _U(0x94D049BB133111EB)
        return z ^ (z >> _U(31))


def _scale32(r, rng):
    return ((r >> _U(32)) * _U(rng)) >> _U(32)


def synth_text_np(sy, first_pair: int, n_pairs: int):
    """(seq, qual) uint8 matrices (2*n_pairs, read_len) of a `capi.Synth` sample, numpy only."""
    L, G = int(sy.read_len), int(sy.genome_len)
    with np.errstate(over="ignore"):
        pair = np.arange(first_pair, first_pair + n_pairs, dtype=np.uint64)
        key = _mix64(_U(sy.read_seed) ^ (pair * _PHI + _U(1)))
        k1 = _mix64(key + _U(1))
        start = _scale32(key, G - (sy.insert_lo + sy.insert_span))
        end = start + _U(sy.insert_lo) + _scale32(k1, sy.insert_span)
        hap = (k1 & _U(1)).astype(bool)
        seq = np.zeros((2 * n_pairs, L), dtype=np.uint8)
        qual = np.zeros_like(seq)
        j = np.arange(L, dtype=np.uint64)
        st = (G - 2000) // sy.n_snv if sy.n_snv else 0
        for mate in (0, 1):
            mk = _mix64(key + _U(2 + mate))
            x = (end[:, None] - _U(1) - j[None, :]) if mate else (start[:, None] + j[None, :])
            gw = _mix64(_U(sy.genome_seed) ^ ((x >> _U(5)) * _PHI))
            b = ((gw >> (_U(2) * (x & _U(31)))) & _U(3)).astype(np.int64)
            if sy.carrier and sy.n_snv:
                i = np.minimum((np.maximum(x, _U(1000)) - _U(1000)) // _U(st), _U(sy.n_snv - 1))
                h = _mix64(_U(sy.snv_seed) + i)
                spos = _U(1000) + i * _U(st) + _scale32(h, st - 64)
                alt = (b + 1 + ((h & _U(0xFFFF)) % _U(3)).astype(np.int64)) & 3
                b = np.where((spos == x) & hap[:, None], alt, b)
            if mate:
                b = 3 - b
            r = _mix64(mk[:, None] + ((j[None, :] >> _U(1)) + _U(1)) * _STEP)
            bits = np.where((j[None, :] & _U(1)) == 1, r >> _U(32), r & _U(0xFFFFFFFF)).astype(np.int64)
            err = (bits & 1023) < sy.err_1024
            b = np.where(err, (b + 1 + ((bits >> 10) & 15) % 3) & 3, b)
            lowq = ((bits >> 14) & 255) < sy.lowq_256
            is_n = (bits >> 22) < sy.n_1024
            s = ACGT[b]
            s[is_n] = ord("N")
            seq[mate::2] = s
            qual[mate::2] = np.where(lowq, ord("#"), ord("J"))
    return seq, qual


def synth_flat(seq: np.ndarray, qual: np.ndarray):
    """(seq bytes, qual bytes, off) of read matrices, for capi.PackedReads."""
    n, L = seq.shape
    return seq.tobytes(), qual.tobytes(), np.arange(n + 1, dtype=np.uint64) * np.uint64(L)


def synth_fastq(seq: np.ndarray, qual: np.ndarray, first_read: int = 0, tag: bytes = b"@r") -> bytes:
    """4-line FASTQ text of read matrices."""
    out = []
    for i in range(seq.shape[0]):
        out.append(tag + str(first_read + i).encode() + b"\n" + seq[i].tobytes() + b"\n+\n" + qual[i].tobytes() + b"\n")
    return b"".join(out)
