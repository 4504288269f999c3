"""CPU-only race / memory-error detection: the threaded host pipelines of the drop-in tools built with -fsanitize=thread
(and once more with -fsanitize=address,undefined) and run on small inputs with many threads and tiny pieces -- the count ingest (rfx_ingest.hpp, tests/host/ingest_harness.cpp), the
stranded feeder (pass_through_main.cpp) and `RUFUS.Filter --sam` (tests/host/filter_sam_harness.cpp: the tool's main()
over host stand-ins for the device calls).  A report of ThreadSanitizer fails the test; outputs must equal those of
the uninstrumented run of tests/test_filter_sam_host.py's generator."""
import os
import subprocess

import numpy as np
import pytest

from tests.conftest import ROOT
from tests.test_filter_sam_host import _hash_list, _shuffled_sam

HOST = os.path.join(ROOT, "rufus_amd", "csrc")
FLAGS = ["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=thread"]


@pytest.fixture(scope="module", params=["thread", "address,undefined"])
def tsan_bins(request, tmp_path_factory):
    d = tmp_path_factory.mktemp("san")
    flags = FLAGS[:-1] + ["-fsanitize=" + request.param] + (["-fno-sanitize-recover=all"] if "undefined" in request.param else [])
    probe = d / "probe.cpp"
    probe.write_text("int main() { return 0; }\n")
    if subprocess.run(flags + ["-o", str(d / "probe"), str(probe)], stderr=subprocess.DEVNULL).returncode != 0 or \
            subprocess.run([str(d / "probe")]).returncode != 0:
        pytest.skip(f"no usable -fsanitize={request.param} here")
    host = os.path.join(HOST, "rfx_host.cpp")
    jobs = [subprocess.Popen(flags + ["-o", str(d / "filter"), os.path.join(ROOT, "tests", "host", "filter_sam_harness.cpp"), host]),
            subprocess.Popen(flags + ["-o", str(d / "ingest"), os.path.join(ROOT, "tests", "host", "ingest_harness.cpp"), host]),
            subprocess.Popen(flags + ["-DPTS_MODE=1", "-o", str(d / "feeder"), os.path.join(HOST, "host", "pass_through_main.cpp")])]
    assert all(j.wait() == 0 for j in jobs)
    return d


def _run(cmd, cwd, env, stdin=None):
    # (the tools leave without their teardown once the outputs are closed -- rfx_cli.hpp leave() --: no leak report under
    # ASan; under the sanitizers they take the orderly way, so that threads are joined and the teardown is raced too)
    r = subprocess.run(cmd, cwd=cwd, env=dict(os.environ, RFX_CLEAN_EXIT="1", TSAN_OPTIONS="halt_on_error=0 exitcode=66", ASAN_OPTIONS="detect_leaks=0",
                                              **env), input=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert b"Sanitizer" not in r.stderr and b"runtime error" not in r.stderr and r.returncode == 0, r.stderr[-3000:]
    return r.stdout


def test_host_pipelines_are_race_free(tsan_bins, tmp_path):
    d = str(tmp_path)
    rng = np.random.default_rng(9)
    sam = _shuffled_sam(rng, 1500)
    open(f"{d}/in.sam", "wb").write(sam)
    open(f"{d}/hl", "wb").write(_hash_list(rng, sam, 3000))
    small = {"RFX_INGEST_PIECE": "4096", "RFX_HOST_THREADS": "6"}
    _run([str(tsan_bins / "filter"), "--sam", "a.chr", "hl", "stdin", "a", "25", "15", "1", "6"], d, small, sam)
    _run([str(tsan_bins / "filter"), "--sam", "b.chr", "hl", "in.sam", "b", "25", "15", "1", "6"], d, small)
    for f in ("Mutations.Mate1.fastq", "Mutations.Mate2.fastq", "chr"):
        assert open(f"{d}/a.{f}", "rb").read() == open(f"{d}/b.{f}", "rb").read()
    assert open(f"{d}/a.Mutations.Mate1.fastq", "rb").read().count(b"\n") >= 40
    _run([str(tsan_bins / "feeder"), "p.chr", "p"], d, {"RFX_PTS_THREADS": "4"}, sam)
    # the paired tool on the feeder's two streams (two mate readers in lock step, helpers, ordered writer)
    _run([str(tsan_bins / "filter"), "hl", "p.mate1.fastq", "p.mate2.fastq", "c", "25", "15", "1", "6"], d,
         {"RFX_INGEST_PIECE": "4096"})
    assert open(f"{d}/c.Mutations.Mate1.fastq", "rb").read() == open(f"{d}/a.Mutations.Mate1.fastq", "rb").read()
    seqs = [np.frombuffer(b"ACGTN", np.uint8)[rng.choice(5, int(rng.choice([1, 25, 31, 32, 33, 100, 150, 151])),
                                                          p=[.24, .25, .25, .25, .01])].tobytes() for _ in range(6000)]
    fq = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, s, b"I" * len(s)) for i, s in enumerate(seqs))
    open(f"{d}/x.fq", "wb").write(fq)
    outs = [_run([str(tsan_bins / "ingest"), "8", "500", "3000", "20000", "x.fq"], d, {}),
            _run([str(tsan_bins / "ingest"), "8", "500", "3000", "20000", "x.fq"], d, {"INGEST_MMAP": "1"}),
            _run([str(tsan_bins / "ingest"), "8", "500", "3000", "20000", "-"], d, {}, fq)]
    assert len({o.rsplit(b" ", 1)[0] for o in outs}) == 1 and outs[0].startswith(b"ok 6000 ")


def _mutate(b, rng, cap=60000):
    b = bytearray(b[:int(rng.integers(1, cap))])
    for _ in range(int(rng.integers(1, 40))):
        op, p = int(rng.integers(0, 5)), int(rng.integers(0, len(b)))
        if op == 0:
            b[p] = int(rng.integers(0, 256))
        elif op == 1:
            del b[p:p + int(rng.integers(1, 300))]
        elif op == 2:
            b[p:p] = bytes(rng.integers(0, 256, int(rng.integers(1, 50)), dtype=np.uint8))
        elif op == 3:
            b[p:p] = b"\n" * int(rng.integers(1, 4))
        else:
            b[p:p] = (b"\t", b" ")[int(rng.integers(0, 2))] * int(rng.integers(1, 12))
        if not b:
            b = bytearray(b"x")
    return bytes(b)


def test_host_parsers_survive_malformed_input(tsan_bins, tmp_path):
    """Truncated, spliced and byte-flipped SAM / FASTQ / hash-list text into the ingest, the feeder and both filter
    front ends: a tool may refuse its input (exit code 1 and a message), it may not crash, hang or trip a sanitizer."""
    d = str(tmp_path)
    rng = np.random.default_rng(4)
    sam = _shuffled_sam(rng, 400)
    hl = _hash_list(rng, sam, 800)
    open(f"{d}/hl", "wb").write(hl)
    _run([str(tsan_bins / "feeder"), "p.chr", "p"], d, {}, sam)
    m1, m2 = (open(f"{d}/p.mate{m}.fastq", "rb").read() for m in (1, 2))
    fq = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, b"ACGTN" * 20, b"I" * 100) for i in range(500))
    seen = set()
    for it in range(10):
        open(f"{d}/hl_m", "wb").write(_mutate(hl, rng, len(hl)))
        open(f"{d}/m1_m", "wb").write(_mutate(m1, rng))
        open(f"{d}/m2_m", "wb").write(_mutate(m2, rng))
        bad_sam = _mutate(sam, rng)
        small = {"RFX_INGEST_PIECE": "2048"}
        for cmd, stdin, env in (
                ([str(tsan_bins / "filter"), "--sam", "f.chr", "hl", "stdin", "f", "25", "15", "1", "3"], bad_sam, small),
                ([str(tsan_bins / "filter"), "--sam", "g.chr", "hl_m", "stdin", "g", "25", "15", "1", "3"], sam, small),
                ([str(tsan_bins / "filter"), "hl_m", "m1_m", "m2_m", "k", str(int(rng.integers(1, 33))), "15", "1", "3"], None, small),
                ([str(tsan_bins / "feeder"), "q.chr", "q"], bad_sam, {"RFX_PTS_THREADS": "3"}),
                ([str(tsan_bins / "ingest"), "4", "200", "1500", "3000", "-"], _mutate(fq, rng), {})):
            # (a tool that refuses its input exits with its helper threads still running: no thread-leak report)
            r = subprocess.run(cmd, cwd=d, input=stdin,
                               env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0", TSAN_OPTIONS="report_thread_leaks=0", **env),
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
            assert b"Sanitizer" not in r.stderr and b"runtime error" not in r.stderr and r.returncode in (0, 1), \
                (cmd, r.returncode, r.stderr[-2000:])
            seen.add(r.returncode)
    assert 0 in seen
