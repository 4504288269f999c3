"""CPU-only: the host side of `RUFUS.Filter --sam` (filter half) through tests/host/filter_sam_harness.cpp --
the tool's own main() with the device entry points replaced by host stand-ins -- so that the reader, the helper
threads, the recycling of pieces and the pairing by QNAME across pieces run without a GPU.  Expected bytes: the
two-process route of runRufus.sh:964-967, i.e. the stranded feeder's two FASTQ streams put through the oracle's filter
(and, where the reference filter can read the input, the reference binaries under oracle/_ref).  The real tool is
compared with the same routes in tests/test_cli_gpu.py::test_filter_sam_equals_feeder_plus_filter."""
import os
import subprocess

import numpy as np
import pytest

import oracle
from tests.conftest import ROOT
from tests.test_cli_host import BIN, REF, make_sam

HARNESS_SRC = os.path.join(ROOT, "tests", "host", "filter_sam_harness.cpp")
HOST_SRC = os.path.join(ROOT, "rufus_amd", "csrc", "rfx_host.cpp")
K, MIN_Q, THRESH = 25, 15, 1


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("samf") / "filter_sam_harness")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", out, HARNESS_SRC, HOST_SRC])
    return out


def _shuffled_sam(rng, n=6000):
    """Records in random order, names seen two, three and four times, IUPAC / lower-case bases, qualities shorter than
    the read: thousands of reads wait across many pieces."""
    alphabet = np.frombuffer(b"ACGT" * 8 + b"Nacgtn" + b"RY", np.uint8)
    recs = []
    for i in range(n):
        times = 2 if i % 97 else (3 if i % 2 else 4)
        for t in range(times):
            L = int(rng.integers(30, 151))
            seq = bytes(rng.choice(alphabet, L))
            qual = rng.integers(55, 75, L if i % 53 else max(1, L - 7), dtype=np.uint8)
            qual[rng.random(len(qual)) < 0.03] = 35                # '#': below MinQ
            flag = int(rng.choice([99, 147, 83, 163, 16, 0, 1040, 65]))
            recs.append(b"\t".join([b"q%d" % i, b"%d" % flag, b"chr%d" % (1 + i % 5), b"%d" % (i + t), b"60", b"*", b"=",
                                     b"1", b"0", seq, bytes(qual), b"XS:i:%d" % t]) + b"\n")
    return b"".join(recs[j] for j in rng.permutation(len(recs)))


def _hash_list(rng, sam, n_lines):
    lines = [ln.split(b"\t") for ln in sam.split(b"\n") if ln.count(b"\t") >= 10]
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    kmers = []
    for j in rng.choice(len(lines), n_lines, replace=False):
        sq = lines[j][9]
        ok = [a for a in range(len(sq) - K + 1) if set(sq[a:a + K]) <= set(b"ACGT")]
        for a in (rng.choice(ok, min(3, len(ok)), replace=False) if ok else ()):
            km = sq[int(a):int(a) + K]
            kmers.append(km if rng.random() < 0.5 else km.translate(comp)[::-1])
    assert len(kmers) > 40
    return b"".join(km + b" 9\n" for km in kmers)


def _records(fq):
    ln = fq.split(b"\n")
    assert ln[-1] == b"" and (len(ln) - 1) % 4 == 0
    return [b"\n".join(ln[i:i + 4]) + b"\n" for i in range(0, len(ln) - 1, 4)]


@pytest.mark.parametrize("shape", ["sorted", "shuffled"])
def test_filter_sam_host_side_equals_feeder_plus_filter(harness, tmp_path, shape):
    d = str(tmp_path)
    rng = np.random.default_rng(5)
    sam = make_sam(3000, seed=8) if shape == "sorted" else _shuffled_sam(rng)
    hl = _hash_list(rng, sam, 200 if shape == "sorted" else 1500)
    open(f"{d}/in.sam", "wb").write(sam)
    open(f"{d}/hl", "wb").write(hl)
    # the two-process route: the drop-in feeder (byte-identical to the reference's, tests/test_cli_host.py) ...
    r = subprocess.run(f"{BIN}/PassThroughSamCheck.stranded two.chr two < in.sam > two.log", shell=True, cwd=d,
                       stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr
    m1, m2 = (open(f"{d}/two.mate{m}.fastq", "rb").read() for m in (1, 2))
    # ... and the oracle's filter over its two streams
    pulled = oracle.FilterSet(hl).pairs(m1, m2, K, MIN_Q, THRESH)
    r1, r2 = _records(m1), _records(m2)
    want = [b"".join(r[i] for i in pulled) for r in (r1, r2)]
    assert len(pulled) >= 15
    chr_want = open(f"{d}/two.chr", "rb").read()

    def check(stub):
        for m in (1, 2):
            assert open(f"{d}/{stub}.Mutations.Mate{m}.fastq", "rb").read() == want[m - 1], (stub, m)
        assert open(f"{d}/{stub}.chr", "rb").read() == chr_want, stub

    small = dict(os.environ, RFX_INGEST_PIECE="16384")   # hundreds of pieces: waiting records outlive theirs
    runs = [("pipe1", small, "stdin", "1"), ("pipe5", small, "stdin", "5"), ("pipe3big", dict(os.environ), "stdin", "3"),
            ("file3", small, "in.sam", "3"), ("read3", dict(small, RFX_FILTER_NO_MMAP="1"), "in.sam", "3")]
    for stub, env, src, threads in runs:
        r = subprocess.run([harness, "--sam", f"{stub}.chr", "hl", src, stub, str(K), str(MIN_Q), str(THRESH), threads], cwd=d,
                           env=env, input=sam if src == "stdin" else None, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           timeout=300)
        assert r.returncode == 0, r.stderr
        check(stub)
    if os.path.exists(f"{REF}/RUFUS.Filter") and shape == "sorted":     # (the reference filter needs whole quality lines)
        r = subprocess.run(f"{REF}/PassThroughSamCheck.stranded ref.chr ref < in.sam > ref.log && "
                           f"{REF}/RUFUS.Filter hl ref.mate1.fastq ref.mate2.fastq ref {K} {MIN_Q} {THRESH} 1 > ref.flog",
                           shell=True, cwd=d, stderr=subprocess.PIPE, timeout=300)
        assert r.returncode == 0, r.stderr
        for m in (1, 2):
            assert open(f"{d}/ref.Mutations.Mate{m}.fastq", "rb").read() == want[m - 1]
        assert open(f"{d}/ref.chr", "rb").read() == chr_want


def test_filter_sam_host_side_many_threads_repeated(harness, tmp_path):
    """The same stream ten times with more helpers than cores and pieces of 4 KB: the order of completion of the
    pieces changes from run to run, the outputs must not."""
    d = str(tmp_path)
    rng = np.random.default_rng(9)
    sam = _shuffled_sam(rng, 1500)
    open(f"{d}/hl", "wb").write(_hash_list(rng, sam, 3000))
    env = dict(os.environ, RFX_INGEST_PIECE="4096", RFX_HOST_THREADS="12")
    first = None
    for rep in range(10):
        r = subprocess.run([harness, "--sam", "x.chr", "hl", "stdin", "x", str(K), str(MIN_Q), str(THRESH), "12"], cwd=d, env=env,
                           input=sam, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert r.returncode == 0, r.stderr
        got = tuple(open(f"{d}/{f}", "rb").read() for f in ("x.Mutations.Mate1.fastq", "x.Mutations.Mate2.fastq", "x.chr"))
        assert got[0].count(b"\n") >= 40
        first = first or got
        assert got == first


@pytest.mark.parametrize("route", ["files", "fifos"])
def test_paired_filter_host_side(harness, tmp_path, route):
    """The paired tool's own threading (two mate readers in lock step, pieces, ordered output: src/RUFUS.Filter.cpp:162-277)
    over the host stand-ins: the pairs the oracle pulls, from files and from two named pipes fed by the stranded feeder."""
    d = str(tmp_path)
    rng = np.random.default_rng(21)
    sam = make_sam(4000, seed=12)
    hl = _hash_list(rng, sam, 300)
    open(f"{d}/in.sam", "wb").write(sam)
    open(f"{d}/hl", "wb").write(hl)
    r = subprocess.run(f"{BIN}/PassThroughSamCheck.stranded two.chr two < in.sam > two.log", shell=True, cwd=d,
                       stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr
    m1, m2 = (open(f"{d}/two.mate{m}.fastq", "rb").read() for m in (1, 2))
    pulled = oracle.FilterSet(hl).pairs(m1, m2, K, MIN_Q, THRESH)
    want = [b"".join(r[i] for i in pulled) for r in (_records(m1), _records(m2))]
    assert len(pulled) >= 15
    env = dict(os.environ, RFX_INGEST_PIECE="8192")
    if route == "files":
        r = subprocess.run([harness, "hl", "two.mate1.fastq", "two.mate2.fastq", "out", str(K), str(MIN_Q), str(THRESH), "4"],
                           cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    else:
        os.mkfifo(f"{d}/ff.mate1.fastq")
        os.mkfifo(f"{d}/ff.mate2.fastq")
        feeder = subprocess.Popen(f"{BIN}/PassThroughSamCheck.stranded ff.chr ff < in.sam > ff.log", shell=True, cwd=d)
        r = subprocess.run([harness, "hl", "ff.mate1.fastq", "ff.mate2.fastq", "out", str(K), str(MIN_Q), str(THRESH), "4"],
                           cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert feeder.wait(timeout=60) == 0
    assert r.returncode == 0, r.stderr
    for m in (1, 2):
        assert open(f"{d}/out.Mutations.Mate{m}.fastq", "rb").read() == want[m - 1]
