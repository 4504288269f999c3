"""CPU-only: the sequential FASTA/FASTQ parser of the drop-in `jellyfish` (rfx_cli.hpp parse_sequences) against the oracle's
own reading of jf/include/jellyfish/mer_overlap_sequence_parser.hpp:124-251 -- the k-mers of the sequences the harness
yields must be the k-mers the oracle counts in the file, on wrapped FASTA, multi-line FASTQ, empty records, blank lines,
CRLF-free odd spacing, no final newline, buffers of a few bytes (every line straddles a refill); malformed files are
refused by both.  The harness is built with -fsanitize=address,undefined when the compiler has it."""
import os
import subprocess

import numpy as np
import pytest

import oracle
from tests.conftest import ROOT

SRC = [os.path.join(ROOT, "tests", "host", "sequences_harness.cpp"), os.path.join(ROOT, "rufus_amd", "csrc", "rfx_host.cpp")]


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("sq") / "sequences_harness")
    base = ["g++", "-O1", "-g", "-std=c++17", "-pthread", "-o", out] + SRC
    if subprocess.run(base + ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"], stderr=subprocess.DEVNULL).returncode != 0:
        subprocess.check_call(base)
    return out


def _parse(harness, data, buf):
    r = subprocess.run([harness, "stdin", str(buf)], input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0 and b"Sanitizer" not in r.stderr and b"runtime error" not in r.stderr, r.stderr[-2000:]
    lines = r.stdout.split(b"\n")
    if lines[-2] == b"malformed":
        return None
    assert lines[-2].startswith(b"ok ") and int(lines[-2][3:]) == len(lines) - 2
    return lines[:-2]


def _text(rng, kind):
    alphabet = np.frombuffer(b"ACGTacgtNn", np.uint8)
    out = []
    for i in range(int(rng.integers(1, 40))):
        L = int(rng.choice([0, 1, 4, 5, 6, 30, 61, 150, 400]))
        seq = bytes(rng.choice(alphabet, L, p=[.22, .22, .22, .22, .02, .02, .02, .02, .02, .02]))
        width = int(rng.choice([1, 7, 60, 1000]))
        rows = [seq[j:j + width] for j in range(0, L, width)] or ([b""] if rng.random() < 0.5 else [])
        if kind == "fasta":
            out.append(b">r%d some text\n" % i + b"".join(r + b"\n" for r in rows))
        else:
            qual = bytes(rng.integers(33, 74, L, dtype=np.uint8)).replace(b"@", b"A").replace(b"+", b"B")
            qrows = [qual[j:j + width] for j in range(0, L, width)] or ([b""] if rows else [])
            out.append(b"@r%d\n" % i + b"".join(r + b"\n" for r in rows) + b"+\n" + b"".join(q + b"\n" for q in qrows))
            if rng.random() < 0.2:
                out.append(b"\n")                                    # a blank line between records
    data = b"".join(out)
    return data[:-1] if data.endswith(b"\n") and rng.random() < 0.3 else data


@pytest.mark.parametrize("kind", ["fasta", "fastq"])
def test_parser_yields_what_the_oracle_counts(harness, kind):
    rng = np.random.default_rng(3 if kind == "fasta" else 4)
    agreed = refused = 0
    for it in range(60):
        data = _text(rng, kind)
        k = int(rng.choice([4, 5, 12, 25]))
        size = 1 << min(20, 2 * k)
        try:
            want = oracle.count([data], k, size)
        except ValueError:
            want = None
        for buf in (16, 64, 1 << 16):
            seqs = _parse(harness, data, buf)
            if want is None:
                assert seqs is None, data[:200]
                refused += 1
                continue
            assert seqs is not None, data[:200]
            got = oracle.count(None, k, size, reads=seqs)
            assert got.payload() == want.payload(), (it, buf, data[:300])
            agreed += 1
    assert agreed >= 100


def test_parser_refuses_or_survives_garbage(harness):
    rng = np.random.default_rng(8)
    for it in range(40):
        data = bytearray(_text(rng, "fastq" if it % 2 else "fasta"))
        for _ in range(int(rng.integers(1, 12))):
            p = int(rng.integers(0, len(data)))
            op = int(rng.integers(0, 3))
            if op == 0:
                data[p] = int(rng.integers(0, 256))
            elif op == 1:
                del data[p:p + int(rng.integers(1, 40))]
            else:
                data[p:p] = bytes(rng.choice(np.frombuffer(b"@+>\n\n ACGT", np.uint8), int(rng.integers(1, 8))))
            if not data:
                data = bytearray(b"@")
        data = bytes(data)
        try:
            want = oracle.count([data], 5, 1 << 10)
        except ValueError:
            want = None
        seqs = _parse(harness, data, 32)
        if want is not None and seqs is not None:
            assert oracle.count(None, 5, 1 << 10, reads=seqs).payload() == want.payload(), data[:300]
