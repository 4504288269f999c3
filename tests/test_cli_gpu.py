"""GPU: the drop-in executables (same argv as the reference tools) end to end on the reference's own
test trio -- count -> histo -> merge -> query/hash list -> filter -- against the golden fixtures,
including the named-pipe plumbing of runRufus.sh:964-967."""
import hashlib
import os
import subprocess
import threading

import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "rufus_amd", "bin")


def sh(args, cwd, stdin=None, timeout=180):
    return subprocess.run(args, cwd=cwd, input=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)


def test_jellyfish_and_filter_executables_reproduce_the_goldens(testrun, tmp_path):
    exp = testrun["expected"]
    d = str(tmp_path)
    for s in ("Child", "Mother", "Father"):
        # the FASTQ route feeds both mate files through one pipe (RunJellyForRUFUS.sh:28)
        open(f"{d}/{s}.fq", "wb").write(testrun[s][0] + testrun[s][1])
        r = sh([f"{BIN}/jellyfish", "count", "--disk", "-m", "25", "-L", "2", "-s", "100M", "-t", "4", "-o", f"{s}.Jhash",
                "-C", f"{s}.fq"], d)
        assert r.returncode == 0, r.stderr
        blob = open(f"{d}/{s}.Jhash", "rb").read()
        hlen = int(blob[:9])
        assert hashlib.sha256(blob[9 + hlen:]).hexdigest() == exp["samples"][s]["s100M"]["payload_sha256"]
        r = sh([f"{BIN}/jellyfish", "histo", "-f", "-o", f"{s}.Jhash.histo", f"{s}.Jhash"], d)
        assert r.returncode == 0, r.stderr
        h = open(f"{d}/{s}.Jhash.histo", "rb").read()
        assert hashlib.md5(h).hexdigest() == exp["samples"][s]["s100M"]["histo_full_md5"]
    # modified merge: stdout is the data channel, plus the header-only side file
    r = sh([f"{BIN}/jellyfish", "merge", "Child.Jhash", "Mother.Jhash", "Father.Jhash"], d)
    assert r.returncode == 0 and r.stdout.decode() == testrun["merge"]
    assert os.path.getsize(f"{d}/mer_counts_merged.jf") > 1000
    # scripts/CheckJellyHashList.sh:12
    open(f"{d}/q.fa", "w").write("".join(f">{ln.split()[0]}\n{ln.split()[0]}\n" for ln in testrun["merge"].splitlines()))
    r = sh([f"{BIN}/jellyfish", "query", "-s", "q.fa", "Child.Jhash"], d)
    assert r.returncode == 0, r.stderr
    hl = "".join(ln + "\n" for ln in r.stdout.decode().splitlines() if 5 <= int(ln.split()[1]) <= 140)
    assert hl == testrun["hashlist"]
    open(f"{d}/Child.HashList", "w").write(hl)
    # dump -c and a command-line query
    r = sh([f"{BIN}/jellyfish", "dump", "-c", "Child.Jhash"], d)
    lines = r.stdout.decode().splitlines()
    assert len(lines) == 18356 and lines[0] == "A" * 25 + " 48"
    r = sh([f"{BIN}/jellyfish", "query", "Child.Jhash", "T" * 25, "ACGT"], d)
    assert r.stdout.decode() == "A" * 25 + " 48\n" and b"Invalid mer" in r.stderr
    # merging databases of different table sizes is refused (merge_files.cc:193-203)
    sh([f"{BIN}/jellyfish", "count", "-m", "25", "-s", "1M", "-o", "small.Jhash", "-C", "Father.fq"], d)
    r = sh([f"{BIN}/jellyfish", "merge", "Child.Jhash", "small.Jhash"], d)
    assert r.returncode != 0 and b"different size" in r.stderr

    # RUFUS.Filter / RUFUS.Filter.single: byte-identical to the reference binaries' outputs
    for m in (1, 2):
        open(f"{d}/m{m}.fq", "wb").write(testrun["Child"][m - 1])
    r = sh([f"{BIN}/RUFUS.Filter", "Child.HashList", "m1.fq", "m2.fq", "out", "25", "15", "1", "6"], d)
    assert r.returncode == 0, r.stderr
    for m in (1, 2):
        data = open(f"{d}/out.Mutations.Mate{m}.fastq", "rb").read()
        assert hashlib.sha256(data).hexdigest() == exp["filter_paired_sha256"][str(m)]
    r = sh([f"{BIN}/RUFUS.Filter.single", "Child.HashList", "m1.fq", "se", "25", "15", "1", "6"], d)
    assert hashlib.sha256(open(f"{d}/se.Mutations.fastq", "rb").read()).hexdigest() == exp["filter_single_sha256"]
    # missing inputs: message on stdout, exit status 0 (the shell checks for empty outputs instead)
    r = sh([f"{BIN}/RUFUS.Filter", "nope", "m1.fq", "m2.fq", "x", "25", "15", "1", "6"], d)
    assert r.returncode == 0 and b"could not be opened" in r.stdout


def test_filter_reads_lock_step_named_pipes(testrun, tmp_path):
    """runRufus.sh:964-967: the feeder writes both mate pipes record by record while RUFUS.Filter
    reads them; reading one pipe ahead of the other would deadlock."""
    d = str(tmp_path)
    open(f"{d}/hl", "w").write(testrun["hashlist"])
    os.mkfifo(f"{d}/p.mate1.fastq")
    os.mkfifo(f"{d}/p.mate2.fastq")
    m1, m2 = (x.split(b"\n") for x in testrun["Child"])

    def feed():
        with open(f"{d}/p.mate1.fastq", "wb", buffering=0) as f1, open(f"{d}/p.mate2.fastq", "wb", buffering=0) as f2:
            for i in range(0, len(m1) - 1, 4):
                f1.write(b"\n".join(m1[i:i + 4]) + b"\n")
                f2.write(b"\n".join(m2[i:i + 4]) + b"\n")

    t = threading.Thread(target=feed, daemon=True)
    t.start()
    r = sh([f"{BIN}/RUFUS.Filter", "hl", "p.mate1.fastq", "p.mate2.fastq", "piped", "25", "15", "1", "4"], d, timeout=120)
    t.join(30)
    assert r.returncode == 0 and not t.is_alive()
    exp = testrun["expected"]
    for m in (1, 2):
        data = open(f"{d}/piped.Mutations.Mate{m}.fastq", "rb").read()
        assert hashlib.sha256(data).hexdigest() == exp["filter_paired_sha256"][str(m)]


def test_count_reads_a_named_pipe_and_several_files(testrun, tmp_path):
    d = str(tmp_path)
    os.mkfifo(f"{d}/gen.fq")

    def feed():
        with open(f"{d}/gen.fq", "wb") as f:
            f.write(testrun["Mother"][0] + testrun["Mother"][1])

    t = threading.Thread(target=feed, daemon=True)
    t.start()
    r = sh([f"{BIN}/jellyfish", "count", "--disk", "-m", "25", "-L", "2", "-s", "8G", "-t", "38", "-o", "M.Jhash", "-C",
            "gen.fq"], d)
    t.join(30)
    assert r.returncode == 0, r.stderr
    blob = open(f"{d}/M.Jhash", "rb").read()
    want = testrun["expected"]["samples"]["Mother"]["s8G"]["payload_sha256"]
    assert hashlib.sha256(blob[9 + int(blob[:9]):]).hexdigest() == want
    # two files on the command line: k-mers do not span files, same result
    for m in (1, 2):
        open(f"{d}/m{m}.fq", "wb").write(testrun["Mother"][m - 1])
    sh([f"{BIN}/jellyfish", "count", "-m", "25", "-L2", "-s", "8G", "-o", "M2.Jhash", "-C", "m1.fq", "m2.fq"], d)
    blob = open(f"{d}/M2.Jhash", "rb").read()
    assert hashlib.sha256(blob[9 + int(blob[:9]):]).hexdigest() == want
    assert sh([f"{BIN}/jellyfish", "count", "-m", "25", "-s", "1M", "nonexistent.fa"], d).returncode != 0
