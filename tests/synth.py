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
