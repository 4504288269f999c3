"""Host-only test of the parallel FASTQ ingest pipeline of the drop-in `jellyfish count`
(rufus_amd/csrc/host/rfx_ingest.hpp) through tests/host/ingest_harness.cpp: many threads, tiny staging blocks
and tiny pieces, so that every block hand-over path runs thousands of times -- on a mapped file and on a pipe."""
import os
import subprocess

import numpy as np
import pytest

from rufus_amd import capi
from tests.conftest import ROOT

HARNESS_SRC = os.path.join(ROOT, "tests", "host", "ingest_harness.cpp")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("ingest") / "ingest_harness")
    lib = os.path.join(ROOT, "rufus_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", out, HARNESS_SRC, f"-L{lib}", "-lrufus_hip",
                           f"-Wl,-rpath,{lib}"])
    return out


def _fnv(h, v):
    for i in range(8):
        h ^= (v >> (8 * i)) & 255
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _expected(seqs):
    p = capi.PackedReads.from_reads(seqs)
    total = 0
    for r, s in enumerate(seqs):
        h = _fnv(0xCBF29CE484222325, len(s))
        for w in range(int(p.word_off[r]), int(p.word_off[r + 1])):
            h = _fnv(_fnv(h, int(p.codes[w])), int(p.acgt[w]))
        total = (total + h) & 0xFFFFFFFFFFFFFFFF
    return total


@pytest.mark.parametrize("threads,cap_reads,piece", [(8, 500, 20_000), (3, 64, 3_000), (16, 4000, 100_000)])
def test_ingest_pipeline_delivers_every_read_once(harness, tmp_path, threads, cap_reads, piece):
    rng = np.random.default_rng(threads)
    seqs = []
    for i in range(6000):
        n = int(rng.choice([1, 25, 31, 32, 33, 100, 150, 151]))
        s = np.frombuffer(b"ACGTN", np.uint8)[rng.choice(5, n, p=[.24, .25, .25, .25, .01])].tobytes()
        seqs.append(s)
    fq = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, s, b"I" * len(s)) for i, s in enumerate(seqs))
    path = tmp_path / "x.fq"
    path.write_bytes(fq)
    want = f"ok {len(seqs)} {sum(map(len, seqs))} {_expected(seqs)}"
    for rep in range(3):
        # ranges read by the workers / the mapped file, its parsed ranges dropped from the page table by the workers
        # (rfx_cli.hpp drop_mapped: pieces far smaller than a page here, pieces of many pages with the third parameter set) or kept
        for env in ({}, {"INGEST_MMAP": "1"}, {"INGEST_MMAP": "1", "RFX_KEEP_PTES": "1"}):
            out = subprocess.run([harness, str(threads), str(cap_reads), str(cap_reads * 6), str(piece), str(path)],
                                 stdout=subprocess.PIPE, timeout=120, check=True, env={**os.environ, **env}).stdout.decode()
            assert out.rsplit(" ", 1)[0] == want, out
            assert int(out.split()[-1]) >= len(seqs) // cap_reads
        out = subprocess.run([harness, str(threads), str(cap_reads), str(cap_reads * 6), str(piece), "-"], input=fq,
                             stdout=subprocess.PIPE, timeout=120, check=True).stdout.decode()
        assert out.rsplit(" ", 1)[0] == want, out


def test_ingest_rejects_what_is_not_4_line_fastq(harness, tmp_path):
    (tmp_path / "a.fa").write_bytes(b">x\nACGT\n>y\nGGCC\n")
    out = subprocess.run([harness, "2", "100", "600", "1000", str(tmp_path / "a.fa")], stdout=subprocess.PIPE,
                         timeout=60, check=True).stdout.decode()
    assert out.startswith("not4line")
    (tmp_path / "w.fq").write_bytes(b"@x\nACGT\nACGT\n+\nIIII\nIIII\n")
    out = subprocess.run([harness, "2", "100", "600", "1000", str(tmp_path / "w.fq")], stdout=subprocess.PIPE,
                         timeout=60, check=True).stdout.decode()
    assert out.startswith("not4line")


def test_line_scanners_agree_with_memchr(tmp_path):
    """skip_lines / index_lines / find_nl (32 or 16 bytes per step) against their memchr versions: random text,
    empty lines, every buffer alignment, with and without a final newline (tests/host/lines_harness.cpp)."""
    out = str(tmp_path / "lines_harness")
    lib = os.path.join(ROOT, "rufus_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", out,
                           os.path.join(ROOT, "tests", "host", "lines_harness.cpp"), f"-L{lib}", "-lrufus_hip",
                           f"-Wl,-rpath,{lib}"])
    r = subprocess.run([out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith(b"ok "), r.stdout + r.stderr
