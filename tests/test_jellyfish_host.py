"""CPU-only: the drop-in `jellyfish` (count / histo / merge / query / dump) through
tests/host/jellyfish_harness.cpp -- the tool's own main() over host stand-ins for the device entry points -- so that its
argument handling, the parallel ingest, the .Jhash reader / writer, the position-range walks of merge and query and the
several-device plumbing (RUFUS_GPUS) run without a GPU: against the golden fixtures of the reference's testRun trio, the
oracle, and jellyfish's own md5 known answers.  tests/test_cli_gpu.py runs the same chain through the real executable."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

import oracle
from tests.conftest import ROOT

SRC = [os.path.join(ROOT, "tests", "host", "jellyfish_harness.cpp"), os.path.join(ROOT, "rufus_amd", "csrc", "rfx_host.cpp")]


def _build(out, extra):
    subprocess.check_call(["g++", "-std=c++17", "-pthread", "-o", out] + extra + SRC)
    return out


@pytest.fixture(scope="module")
def jf(tmp_path_factory):
    return _build(str(tmp_path_factory.mktemp("jfh") / "jellyfish"), ["-O2"])


@pytest.fixture(scope="module", params=["thread", "address,undefined"])
def jf_san(request, tmp_path_factory):
    d = tmp_path_factory.mktemp("jfs")
    probe = d / "probe.cpp"
    probe.write_text("int main() { return 0; }\n")
    flags = ["-O1", "-g", "-fsanitize=" + request.param]
    if subprocess.run(["g++"] + flags + ["-o", str(d / "probe"), str(probe)], stderr=subprocess.DEVNULL).returncode != 0 or \
            subprocess.run([str(d / "probe")]).returncode != 0:
        pytest.skip(f"no usable -fsanitize={request.param} here")
    return _build(str(d / "jellyfish"), flags)


def sh(cmd, cwd, env=None, timeout=600, **kw):
    r = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0", TSAN_OPTIONS="report_thread_leaks=0", **(env or {})), **kw)
    assert b"Sanitizer" not in r.stderr and b"runtime error" not in r.stderr, r.stderr[-3000:]
    return r


def _payload(path):
    blob = open(path, "rb").read()
    return blob[9 + int(blob[:9]):]


def _golden_chain(jf, testrun, d):
    exp = testrun["expected"]
    for s in ("Child", "Mother", "Father"):
        open(f"{d}/{s}.fq", "wb").write(testrun[s][0] + testrun[s][1])
        r = sh([jf, "count", "--disk", "-m", "25", "-L", "2", "-s", "100M", "-t", "4", "-o", f"{s}.Jhash", "-C", f"{s}.fq"], d)
        assert r.returncode == 0, r.stderr
        assert hashlib.sha256(_payload(f"{d}/{s}.Jhash")).hexdigest() == exp["samples"][s]["s100M"]["payload_sha256"]
        r = sh([jf, "histo", "-f", "-o", f"{s}.Jhash.histo", f"{s}.Jhash"], d)
        assert r.returncode == 0, r.stderr
        h = open(f"{d}/{s}.Jhash.histo", "rb").read()
        assert hashlib.md5(h).hexdigest() == exp["samples"][s]["s100M"]["histo_full_md5"]
        r = sh([jf, "histo", "-f", f"{s}.Jhash"], d, env={"RFX_HISTO_SLICE_RECORDS": "999"})      # the database in 19 pieces
        assert r.stdout == h
    r = sh([jf, "count", "--disk", "-m", "25", "-L", "2", "-s", "100M", "-t", "4", "-o", "side.Jhash", "-C", "Child.fq"], d,
           env={"RFX_COUNT_HISTO": "1"})
    assert r.returncode == 0
    assert hashlib.md5(open(f"{d}/side.Jhash.histo", "rb").read()).hexdigest() == exp["samples"]["Child"]["s100M"]["histo_full_md5"]
    r = sh([jf, "merge", "Child.Jhash", "Mother.Jhash", "Father.Jhash"], d)
    assert r.returncode == 0 and r.stdout.decode() == testrun["merge"]
    assert os.path.getsize(f"{d}/mer_counts_merged.jf") > 1000
    for slices in ("2", "7", "300"):
        r = sh([jf, "merge", "Child.Jhash", "Mother.Jhash", "Father.Jhash"], d, env={"RFX_MERGE_SLICES": slices})
        assert r.returncode == 0 and r.stdout.decode() == testrun["merge"], (slices, r.stderr)
    open(f"{d}/q.fa", "w").write("".join(f">{ln.split()[0]}\n{ln.split()[0]}\n" for ln in testrun["merge"].splitlines()))
    r = sh([jf, "query", "-s", "q.fa", "Child.Jhash"], d)
    assert r.returncode == 0, r.stderr
    hl = "".join(ln + "\n" for ln in r.stdout.decode().splitlines() if 5 <= int(ln.split()[1]) <= 140)
    assert hl == testrun["hashlist"]
    whole = r.stdout
    for per in ("1000", "37"):
        r = sh([jf, "query", "-s", "q.fa", "Child.Jhash"], d, env={"RFX_QUERY_SLICE_RECORDS": per})
        assert r.returncode == 0 and r.stdout == whole, r.stderr
    r = sh([jf, "query", "-s", "q.fa", "-o", "o1", "-o", "o2", "-o", "o3", "Child.Jhash", "Mother.Jhash", "Father.Jhash"], d,
           env={"RFX_QUERY_SLICE_RECORDS": "500"})
    assert r.returncode == 0 and open(f"{d}/o1", "rb").read() == whole, r.stderr
    for s, o in (("Mother", "o2"), ("Father", "o3")):
        assert open(f"{d}/{o}", "rb").read() == sh([jf, "query", "-s", "q.fa", f"{s}.Jhash"], d).stdout
    # few k-mers against a big database: only the records at the queried positions are read (one small sorted database
    # goes to the device) -- the same lines as the walk over position ranges, hits, misses and several databases alike
    few = [ln.split()[0] for ln in testrun["merge"].splitlines()][:60] + ["ACGTTGCA" * 3 + "A", "C" * 25, "G" * 24 + "A"]
    open(f"{d}/few.fa", "w").write("".join(f">{i}\n{km}\n" for i, km in enumerate(few)))
    a = sh([jf, "query", "-s", "few.fa", "Child.Jhash", "Mother.Jhash", "Father.Jhash"], d, env={"RFX_QUERY_SPARSE_RATIO": "64"})
    b = sh([jf, "query", "-s", "few.fa", "Child.Jhash", "Mother.Jhash", "Father.Jhash"], d, env={"RFX_QUERY_NO_SPARSE": "1"})
    assert a.returncode == 0 and b.returncode == 0 and a.stdout == b.stdout and a.stdout.count(b"\n") == len(few), a.stderr
    assert a.stdout.decode().splitlines()[:60] == [ln for ln in whole.decode().splitlines()[:60]
                                                   ] or all(x.split()[:2] == y.split()[:2] for x, y in
                                                            zip(a.stdout.decode().splitlines()[:60], whole.decode().splitlines()[:60]))
    r = sh([jf, "dump", "-c", "Child.Jhash"], d)
    lines = r.stdout.decode().splitlines()
    assert len(lines) == 18356 and lines[0] == "A" * 25 + " 48"
    r = sh([jf, "query", "Child.Jhash", "T" * 25, "ACGT"], d)
    assert r.stdout.decode() == "A" * 25 + " 48\n" and b"Invalid mer" in r.stderr
    sh([jf, "count", "-m", "25", "-s", "1M", "-o", "small.Jhash", "-C", "Father.fq"], d)
    r = sh([jf, "merge", "Child.Jhash", "small.Jhash"], d)
    assert r.returncode != 0 and b"different size" in r.stderr


def test_golden_chain(jf, testrun, tmp_path):
    _golden_chain(jf, testrun, str(tmp_path))


def test_golden_chain_under_sanitizers(jf_san, testrun, tmp_path):
    _golden_chain(jf_san, testrun, str(tmp_path))


@pytest.mark.parametrize("k,size,extra,kw", [(25, "100M", ["-C", "-L", "2"], dict(lower=2)),
                                             (31, "8G", ["-C", "-U", "30"], dict(upper=30)),
                                             (12, "1M", [], dict(canonical=False)),
                                             (25, "100M", ["-C", "--out-counter-len", "1"], dict())])
def test_count_routes_match_oracle(jf, small_trio, tmp_path, k, size, extra, kw):
    """Files, one pipe, wrapped FASTA, several devices (RUFUS_GPUS) and deferred shard passes: the oracle's payload."""
    from tests.synth import fastq_bytes
    d = str(tmp_path)
    fq = [fastq_bytes(small_trio["child"], m) for m in (1, 2)]
    for m in (0, 1):
        open(f"{d}/m{m}.fq", "wb").write(fq[m])
    n = {"100M": 100_000_000, "8G": 8 << 30, "1M": 1_000_000}[size]
    want = oracle.count(fq, k, n, **kw)
    clen = 1 if "--out-counter-len" in extra else 4
    ref = want.payload(clen) if clen != 4 else want.payload()
    base = [jf, "count", "-m", str(k), "-s", size, "-t", "4"] + extra
    assert sh(base + ["-o", "a.jf", "m0.fq", "m1.fq"], d).returncode == 0 and _payload(f"{d}/a.jf") == ref
    r = sh("cat m0.fq m1.fq | " + " ".join(base + ["-o", "b.jf", "/dev/stdin"]), d, shell=True)
    assert r.returncode == 0 and _payload(f"{d}/b.jf") == ref, r.stderr
    for gpus in ("0,0", "0-0,0,0,0,0"):
        r = sh(base + ["-o", "c.jf", "m0.fq", "m1.fq"], d, env={"RUFUS_GPUS": gpus})
        assert r.returncode == 0 and _payload(f"{d}/c.jf") == ref, r.stderr
    r = sh(base + ["-o", "e.jf", "m0.fq", "m1.fq"], d, env={"RFX_COUNT_PASSES": "3", "RFX_HOST_THREADS": "7"})
    assert r.returncode == 0 and _payload(f"{d}/e.jf") == ref, r.stderr
    r = sh(base + ["-o", "/dev/stdout", "m0.fq", "m1.fq"], d)
    assert r.returncode == 0 and r.stdout[9 + int(r.stdout[:9]):] == ref


def test_count_reproduces_jellyfish_own_md5_kats(jf, tmp_path):
    """tests/parallel_hashing.sh of jellyfish-2.2.5: `count -m 15 -C -s 2M` (+ `-L2 -U3 --disk`) on the seeded 10 Mb
    sequence, md5 of `histo` -- through the tool's ingest, writer, reader and histo."""
    from tests.test_oracle import mt_sequence
    d = str(tmp_path)
    seq10m, = mt_sequence(3141592653, [10_000_000])
    with open(f"{d}/seq10m.fa", "wb") as f:
        f.write(b">read0\n")
        for i in range(0, len(seq10m), 70):
            f.write(seq10m[i:i + 70] + b"\n")
    for extra, md5 in (([], "864c0b0826854bdc72a85d170549b64b"), (["-L2", "-U3", "--disk"], "94625cd2d59e278f08421a673eb0926a")):
        r = sh([jf, "count", "-t", "4", "-o", "m15.jf", "-s", "2M", "-C", "-m", "15"] + extra + ["seq10m.fa"], d)
        assert r.returncode == 0, r.stderr
        r = sh([jf, "histo", "m15.jf"], d)
        assert r.returncode == 0 and hashlib.md5(r.stdout).hexdigest() == md5


def _sam_of(testrun, rng):
    lines = []
    chrs = [b"chr1", b"chr1", b"chr2", b"chr10", b"chr1", b"chrX", b"*"]
    for m, text in enumerate(testrun["Child"]):
        recs = text.split(b"\n")
        for i in range(0, len(recs) - 1, 4):
            c = chrs[min(len(chrs) - 1, (i // 4) * len(chrs) // (len(recs) // 4))] if m == 0 else chrs[int(rng.integers(0, 3))]
            lines.append(b"\t".join([recs[i][1:], b"99", c, b"%d" % (i + 1), b"60", b"100M", b"=", b"1", b"0", recs[i + 1],
                                     recs[i + 3], b"NM:i:0"]))
    return lines


def _count_sam(jf, testrun, d):
    """`jellyfish count --sam X.chr [--spool FILE]` on SAM text = the oracle's count of field 10 of
    every line, the chromosome log of the reference's PassThroughSamCheck, the spool a copy of the stream."""
    lines = _sam_of(testrun, np.random.default_rng(11))
    sam = b"\n".join(lines) + b"\n"
    open(f"{d}/in.sam", "wb").write(sam)
    want = oracle.count(None, 25, 100_000_000, lower=2, reads=[ln.split(b"\t")[9] for ln in lines]).payload()
    env = {"RFX_INGEST_PIECE": "65536"}
    cmd = [jf, "count", "--sam", "b.chr", "--disk", "-m", "25", "-L", "2", "-s", "100M", "-t", "6", "-o", "b.Jhash", "-C"]
    for tail, kw in ((["/dev/stdin"], dict(input=sam)), (["in.sam"], {}), (["--spool", "spool.sam", "/dev/stdin"], dict(input=sam))):
        r = sh(cmd[:4] + tail[:-1] + cmd[4:] + tail[-1:], d, env=env, **kw)
        assert r.returncode == 0, r.stderr
        assert _payload(f"{d}/b.Jhash") == want and len(want) > 100_000
        assert open(f"{d}/b.chr").read().split()[:7] == ["notachr", "chr1", "chr2", "chr10", "chr1", "chrX", "*"]
    assert open(f"{d}/spool.sam", "rb").read() == sam
    ref = os.path.join(ROOT, "oracle", "_ref", "PassThroughSamCheck")
    if os.path.exists(ref):
        subprocess.run(f"{ref} ref.chr < in.sam > /dev/null", shell=True, cwd=d, check=True)
        assert open(f"{d}/ref.chr").read() == open(f"{d}/b.chr").read()
    r = sh(cmd + ["/dev/stdin"], d, env=env, input=b"@HD\tVN:1.6\n" + sam)
    assert r.returncode != 0 and b"--sam" in r.stderr


def test_count_sam_and_spool(jf, testrun, tmp_path):
    _count_sam(jf, testrun, str(tmp_path))


def test_count_sam_and_spool_under_sanitizers(jf_san, testrun, tmp_path):
    _count_sam(jf_san, testrun, str(tmp_path))


def test_output_pages_are_prepared_while_the_input_is_read(jf_san, testrun, tmp_path):
    """rfx_cli.hpp OutputPrealloc + write_jhash: a background thread allocates the output's pages and enters them into the
    shared mapping the payload is copied through, from a GUESS of the size -- far too large (the mapping is used, the
    file cut back), far too small (mapped again at the real size), mapping switched off, no preallocation at all: the
    same file every time, nothing left allocated past its end; under ThreadSanitizer / ASan+UBSan."""
    d = str(tmp_path)
    open(f"{d}/c.fq", "wb").write(testrun["Child"][0] + testrun["Child"][1])
    cmd = [jf_san, "count", "--disk", "-m", "25", "-L", "2", "-s", "100M", "-t", "4", "-C"]
    r = sh(cmd + ["-o", "plain.Jhash", "c.fq"], d, env={"RFX_NO_PREALLOC": "1"})
    assert r.returncode == 0, r.stderr
    want = _payload(f"{d}/plain.Jhash")
    assert hashlib.sha256(want).hexdigest() == testrun["expected"]["samples"]["Child"]["s100M"]["payload_sha256"]
    for name, env in (("big", {"RFX_PREALLOC_FRAC": "3.0"}), ("small", {"RFX_PREALLOC_FRAC": "0.0005"}),
                      ("nomap", {"RFX_PREALLOC_FRAC": "3.0", "RFX_NO_PREMAP": "1"}), ("exact", {"RFX_PREALLOC_FRAC": "0.19"})):
        r = sh(cmd + ["-o", f"{name}.Jhash", "c.fq"], d, env=dict(env, RFX_PREALLOC_MIN="0", RFX_CLI_TRACE="1"))
        assert r.returncode == 0, r.stderr
        blob = open(f"{d}/{name}.Jhash", "rb").read()
        assert blob[9 + int(blob[:9]):] == want and len(blob) == 9 + int(blob[:9]) + len(want), name
        assert os.stat(f"{d}/{name}.Jhash").st_blocks * 512 < len(blob) + (1 << 20), name
        assert b"preallocated" in r.stderr
    # a piped input: the guess follows the bytes that have come in (the file grows under a mapping of address space)
    blob = open(f"{d}/c.fq", "rb").read()
    for name, env in (("pipe_big", {"RFX_PREALLOC_FRAC": "3.0"}), ("pipe_small", {"RFX_PREALLOC_FRAC": "0.0005"}),
                      ("pipe_nomap", {"RFX_PREALLOC_FRAC": "3.0", "RFX_NO_PREMAP": "1"}), ("pipe_none", {"RFX_NO_PREALLOC": "1"})):
        r = sh(cmd + ["-o", f"{name}.Jhash", "/dev/stdin"], d, env=dict(env, RFX_PREALLOC_MIN="0", RFX_CLI_TRACE="1", RFX_INGEST_PIECE="65536"),
               input=blob)
        assert r.returncode == 0, r.stderr
        got = open(f"{d}/{name}.Jhash", "rb").read()
        assert got[9 + int(got[:9]):] == want and len(got) == 9 + int(got[:9]) + len(want), name
        assert os.stat(f"{d}/{name}.Jhash").st_blocks * 512 < len(got) + (1 << 20), name
        if name == "pipe_big":
            ready = int(r.stderr.split(b"bytes preallocated, ")[1].split(b" in the mapping")[0])
            assert ready >= 0, r.stderr                  # (how far the background thread got is a matter of timing)
    r = sh(cmd + ["-o", "/dev/stdout", "/dev/stdin"], d, env={"RFX_PREALLOC_MIN": "0"}, input=blob)   # nothing to prepare for a pipe
    assert r.returncode == 0 and r.stdout[9 + int(r.stdout[:9]):] == want


def test_a_count_that_dies_leaves_an_empty_output(jf, testrun, tmp_path):
    """The output is given a size (and mapped) while the input is still parsed; the reference scripts take a non-empty
    .Jhash for a finished one (`[ ! -s X.Jhash ]`, runRufus.sh:806,816: exit codes are not looked at).  A count that
    dies on its input (a SAM line with too few fields, far into the stream) or is killed (SIGTERM while it waits for a
    pipe) therefore cuts the file back to nothing (rfx_cli.hpp UnfinishedOutput)."""
    import signal
    import time
    d = str(tmp_path)
    rng = np.random.default_rng(8)
    sam = b"\n".join(_sam_of(testrun, rng)) + b"\n"
    bad = sam + b"r\t0\tchr1\n" + sam
    open(f"{d}/bad.sam", "wb").write(bad)
    cmd = [jf, "count", "--disk", "-m", "25", "-L", "2", "-s", "100M", "-t", "4", "-C"]
    env = {"RFX_PREALLOC_MIN": "0", "RFX_PREALLOC_FRAC": "3.0", "RFX_INGEST_PIECE": "65536"}
    r = sh(cmd + ["--sam", "x.chr", "-o", "file.Jhash", "bad.sam"], d, env=env)
    assert r.returncode != 0 and b"fewer than 10" in r.stderr
    assert os.path.getsize(f"{d}/file.Jhash") == 0
    r = sh(cmd + ["--sam", "x.chr", "-o", "pipe.Jhash", "/dev/stdin"], d, env=env, input=bad)
    assert r.returncode != 0 and os.path.getsize(f"{d}/pipe.Jhash") == 0
    # killed while the stream is still open: the file had grown under the mapping
    p = subprocess.Popen(cmd + ["--sam", "x.chr", "-o", "killed.Jhash", "/dev/stdin"], cwd=d, stdin=subprocess.PIPE,
                         stderr=subprocess.PIPE, env=dict(os.environ, **env))
    p.stdin.write(sam * 3)
    p.stdin.flush()
    for _ in range(100):
        if os.path.exists(f"{d}/killed.Jhash") and os.path.getsize(f"{d}/killed.Jhash") > 0:
            break
        time.sleep(0.05)
    grown = os.path.getsize(f"{d}/killed.Jhash")
    p.send_signal(signal.SIGTERM)
    p.wait(timeout=60)
    p.stdin.close()
    assert p.returncode != 0 and os.path.getsize(f"{d}/killed.Jhash") == 0, grown


@pytest.mark.parametrize("tool", ["plain", "san"])
def test_count_text_route_and_what_it_hands_back(jf, jf_san, small_trio, tmp_path, tool):
    """Round 6: a regular FASTQ file is not parsed on the host but appended to a text arena piece by piece (TextIngest) and
    parsed behind rfx_text_parse.  Same payload as the host route (RFX_HOST_PARSE=1) and as the oracle: with pieces of
    a few KB (many arenas' worth of tickets and buffer reuse), with a file that lacks its final newline, and with
    blank lines between records / a wrapped record -- which the text route refuses arena by arena and hands to the host
    parser (the reference's grammar, jf mer_overlap_sequence_parser.hpp:179-206)."""
    from tests.synth import fastq_bytes
    exe = jf if tool == "plain" else jf_san
    d = str(tmp_path)
    fq = fastq_bytes(small_trio["child"], 1)
    recs = fq.split(b"\n@")
    recs = [recs[0]] + [b"@" + r for r in recs[1:]]
    blank = b"\n".join(recs[:50]) + b"\n\n" + b"\n".join(recs[50:300]) + b"\n\n\n" + b"\n".join(recs[300:])
    h, s_, p_, q_ = recs[10].split(b"\n")[:4]
    wrapped = b"\n".join(recs[:10] + [h + b"\n" + s_[:70] + b"\n" + s_[70:] + b"\n+\n" + q_[:70] + b"\n" + q_[70:]] + recs[11:])
    cases = {"plain.fq": fq, "nonl.fq": fq.rstrip(b"\n"), "blank.fq": blank}
    want = oracle.count([fq], 25, 100_000_000, lower=2).payload()
    base = [exe, "count", "-m", "25", "-s", "100M", "-t", "4", "-C", "-L", "2"]
    for name, text in cases.items():
        open(f"{d}/{name}", "wb").write(text)
        for env in ({}, {"RFX_HOST_PARSE": "1"}, {"RFX_DEVICE_PARSE": "1", "RFX_INGEST_PIECE": "3000"},
                    {"RFX_DEVICE_PARSE": "1", "RFX_TEXT_PREAD": "1"}, {"RFX_DEVICE_PARSE": "1"}):
            r = sh(base + ["-o", "o.jf", name], d, env=env)
            assert r.returncode == 0 and _payload(f"{d}/o.jf") == want, (name, env, r.stderr[-400:])
    # a wrapped (multi-line) record in the middle of a file that starts like strict FASTQ: the reference parses it; the
    # parallel HOST reader says it cannot (RFX_HOST_THREADS=1 is its advice); the text route hands the text back and
    # the sequential parser counts it
    open(f"{d}/wrapped.fq", "wb").write(wrapped)
    r = sh(base + ["-o", "w.jf", "wrapped.fq"], d, env={"RFX_DEVICE_PARSE": "1"})
    assert r.returncode == 0 and _payload(f"{d}/w.jf") == want, r.stderr[-400:]
