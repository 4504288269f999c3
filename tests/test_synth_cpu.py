"""Host side of the synthetic WGS workload (no GPU): the C++ generator twin against the independent
numpy restatement, its statistics, the CAS hash-table count port against the sort-based oracle, and the
pass planner / pair mask helpers of rufus_amd/wgs.py."""
import numpy as np
import pytest

import oracle
from rufus_amd import capi, wgs
from tests.synth import synth_text_np


@pytest.mark.parametrize("which,first,n", [(0, 0, 700), (0, 123_456_789_012, 300), (1, 5, 257), (2, 99, 64)])
def test_host_twin_matches_numpy_restatement(which, first, n):
    sy = capi.Synth.sample(400_000, which, n_snv=40, seed=31337)
    s, q = sy.text(first, n)
    s2, q2 = synth_text_np(sy, first, n)
    assert np.array_equal(s, s2) and np.array_equal(q, q2)


def test_generator_statistics_and_snvs():
    """Error / low-quality / N rates of SURVEY 8(d); the child's haplotype-1 pairs carry the alt allele of
    every SNV they cover, parents never do; SNVs sit one per stratum, >= 64 bases apart."""
    G = 300_000
    child, mother = capi.Synth.sample(G, 0, n_snv=30), capi.Synth.sample(G, 1, n_snv=30)
    s, q = child.text(0, 30_000)
    assert abs((s == ord("N")).mean() - 1 / 1024) < 2e-4
    assert abs((q == ord("#")).mean() - 5 / 256) < 1e-3
    genome = np.frombuffer(child.genome(0, G), dtype=np.uint8)
    snvs = child.snvs()
    pos = np.array([p for p, _, _ in snvs])
    assert np.all(np.diff(pos) >= 64) and pos[0] >= 1000 and pos[-1] < G - 1000
    for p, ref, alt in snvs:
        assert genome[p:p + 1].tobytes() == ref and alt != ref
    # locate mate-1 reads by exact search of their error-free prefix is overkill: use the geometry instead
    from tests.synth import _PHI, _U, _mix64, _scale32
    pair = np.arange(30_000, dtype=np.uint64)
    with np.errstate(over="ignore"):
        key = _mix64(_U(child.read_seed) ^ (pair * _PHI + _U(1)))
        hap = (_mix64(key + _U(1)) & _U(1)).astype(bool)
    start = _scale32(key, G - 401).astype(np.int64)
    m1 = s[0::2]
    ref_rows = genome[start[:, None] + np.arange(150)[None, :]]
    mism = (m1 != ref_rows) & (m1 != ord("N"))
    assert abs(mism.mean() - 5 / 1024) < 6e-4           # substitution errors (+ the few SNV bases)
    covered = carried = 0
    for p, ref, alt in snvs:
        rows = np.flatnonzero((start <= p) & (p < start + 150))
        for r in rows:
            b = m1[r, p - start[r]:p - start[r] + 1].tobytes()
            if hap[r]:
                covered += 1
                carried += b == alt
            else:
                assert b != alt or mism[r, p - start[r]]   # only a sequencing error can fake it
    assert covered > 50 and carried >= 0.97 * covered
    sm, _ = mother.text(0, 2000)
    assert not np.array_equal(sm, s[:4000])


def test_cas_count_port_matches_sort_port():
    sy = capi.Synth.sample(100_000, 0, n_snv=4)
    seq, _ = sy.text(0, 8000)
    a = oracle.count(None, 25, 1 << 30, lower=2, reads=[r.tobytes() for r in seq])
    for threads in (1, 4):
        b = oracle.count_reads_matrix(seq, 25, 1 << 30, lower=2, threads=threads)
        assert a.payload() == b.payload() and np.array_equal(a.pos, b.pos) and a.total == b.total
    c = oracle.count_reads_matrix(seq, 25, 1 << 30, lower=0, threads=2, table_bits=14)   # forces table growth
    assert c.payload() == oracle.count(None, 25, 1 << 30, reads=[r.tobytes() for r in seq]).payload()


def test_pass_planner_and_pair_mask():
    # full WGS trio: 138 GB of reads resident on a 288 GB part -> several passes; a 1/8 share -> one
    assert 3 <= wgs.plan_passes(620_000_000, 150, 25, 138 << 30, 288 << 30) <= 8
    assert wgs.plan_passes(77_000_000, 150, 25, 17 << 30, 288 << 30) == 1
    m = np.zeros(2, dtype=np.uint64)
    for r in (0, 1, 5, 64, 127):          # pairs 0 (both mates), 2, 32, 63
        m[r // 64] |= np.uint64(1) << np.uint64(r % 64)
    assert wgs.pulled_pairs(m, 128) == 4
