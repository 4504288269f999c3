"""GPU: the drop-in executables (same argv as the reference tools) end to end on the reference's own
test trio -- count -> histo -> merge -> query/hash list -> filter -- against the golden fixtures,
including the named-pipe plumbing of runRufus.sh:964-967."""
import hashlib
import os
import subprocess
import threading

import numpy as np
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
        if s == "Child":      # the opt-in side output of count == what histo -f writes
            r = subprocess.run([f"{BIN}/jellyfish", "count", "--disk", "-m", "25", "-L", "2", "-s", "100M", "-t", "4", "-o",
                                "side.Jhash", "-C", f"{s}.fq"], cwd=d, env=dict(os.environ, RFX_COUNT_HISTO="1"))
            assert r.returncode == 0
            side = open(f"{d}/side.Jhash.histo", "rb").read()
            assert hashlib.md5(side).hexdigest() == exp["samples"][s]["s100M"]["histo_full_md5"]
        r = sh([f"{BIN}/jellyfish", "histo", "-f", "-o", f"{s}.Jhash.histo", f"{s}.Jhash"], d)
        assert r.returncode == 0, r.stderr
        h = open(f"{d}/{s}.Jhash.histo", "rb").read()
        assert hashlib.md5(h).hexdigest() == exp["samples"][s]["s100M"]["histo_full_md5"]
        r = subprocess.run([f"{BIN}/jellyfish", "histo", "-f", f"{s}.Jhash"], cwd=d, stdout=subprocess.PIPE,
                           env=dict(os.environ, RFX_HISTO_SLICE_RECORDS="999"))      # the database in 19 pieces
        assert r.stdout == h
    # modified merge: stdout is the data channel, plus the header-only side file
    r = sh([f"{BIN}/jellyfish", "merge", "Child.Jhash", "Mother.Jhash", "Father.Jhash"], d)
    assert r.returncode == 0 and r.stdout.decode() == testrun["merge"]
    assert os.path.getsize(f"{d}/mer_counts_merged.jf") > 1000
    # the join walks the inputs by position ranges (one range for small inputs; a 30x trio takes 8): any number of
    # ranges, empty ones included, prints the same list
    for slices in ("2", "7", "300"):
        r = subprocess.run([f"{BIN}/jellyfish", "merge", "Child.Jhash", "Mother.Jhash", "Father.Jhash"], cwd=d,
                           env=dict(os.environ, RFX_MERGE_SLICES=slices), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr
        assert r.stdout.decode() == testrun["merge"], slices
    # scripts/CheckJellyHashList.sh:12
    open(f"{d}/q.fa", "w").write("".join(f">{ln.split()[0]}\n{ln.split()[0]}\n" for ln in testrun["merge"].splitlines()))
    r = sh([f"{BIN}/jellyfish", "query", "-s", "q.fa", "Child.Jhash"], d)
    assert r.returncode == 0, r.stderr
    hl = "".join(ln + "\n" for ln in r.stdout.decode().splitlines() if 5 <= int(ln.split()[1]) <= 140)
    assert hl == testrun["hashlist"]
    whole_answer = r.stdout
    for per in ("1000", "37"):        # the database read in ranges of ~per records, only those that got a query
        r = subprocess.run([f"{BIN}/jellyfish", "query", "-s", "q.fa", "Child.Jhash"], cwd=d,
                           env=dict(os.environ, RFX_QUERY_SLICE_RECORDS=per), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0 and r.stdout == whole_answer, r.stderr
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


@pytest.mark.parametrize("route", ["mapped", "read", "pipe"])
def test_filter_pieces_and_ragged_records_match_the_reference_binary(tmp_path, route):
    """More records than one pipeline piece (65536), reads of every length from 26 up, N and lower-case bases, low
    qualities, no newline after the last record -- through the mapped-file, the read() and the pipe route of the
    reader; byte-identical to the reference's own binary (one thread: its output order is then input order)."""
    from tests.conftest import require_ref
    ref = require_ref("RUFUS.Filter")
    from rufus_amd import capi
    d = str(tmp_path)
    n_pairs, G = 70_000, 700_000
    sy = capi.Synth.sample(G, 0, n_snv=40, seed=7)
    seq, qual = sy.text(0, n_pairs)
    rng = np.random.default_rng(5)
    cut = rng.integers(26, 151, 2 * n_pairs)
    cut[rng.random(2 * n_pairs) < 0.5] = 150
    low = rng.random(2 * n_pairs) < 0.01
    mates = [[], []]
    for r in range(2 * n_pairs):
        s_, q_ = bytearray(seq[r, :cut[r]].tobytes()), bytearray(qual[r, :cut[r]].tobytes())
        if low[r]:
            s_[5:9] = bytes(s_[5:9]).lower()
            q_[10:14] = b"####"
        mates[r & 1].append(b"@r%d/%d\n%s\n+\n%s\n" % (r >> 1, (r & 1) + 1, bytes(s_), bytes(q_)))
    m1, m2 = b"".join(mates[0]), b"".join(mates[1])[:-1]        # mate 2: no final newline
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    with open(f"{d}/hl", "w") as f:
        for p_, _, alt in sy.snvs():
            c = bytearray(sy.genome(p_ - 24, 49))
            c[24:25] = alt
            for i in range(25):
                km = bytes(c[i:i + 25])
                f.write(min(km, km[::-1].translate(comp)).decode() + " 12\n")
    open(f"{d}/m1.fq", "wb").write(m1)
    open(f"{d}/m2.fq", "wb").write(m2)
    r = sh([ref, "hl", "m1.fq", "m2.fq", "ref", "25", "15", "1", "1"], d, timeout=600)
    assert r.returncode == 0
    env = dict(os.environ)
    args = [f"{BIN}/RUFUS.Filter", "hl", "m1.fq", "m2.fq", "out", "25", "15", "1", "5"]
    if route == "read":
        env["RFX_FILTER_NO_MMAP"] = "1"
    if route == "pipe":
        os.mkfifo(f"{d}/p1")
        os.mkfifo(f"{d}/p2")
        args[2:4] = ["p1", "p2"]
        feeders = [threading.Thread(target=lambda a=a, b=b: open(f"{d}/{a}", "wb").write(b), daemon=True)
                   for a, b in (("p1", m1), ("p2", m2))]
        for t in feeders:
            t.start()
    r = subprocess.run(args, cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr
    for m in (1, 2):
        got = open(f"{d}/out.Mutations.Mate{m}.fastq", "rb").read()
        want = open(f"{d}/ref.Mutations.Mate{m}.fastq", "rb").read()
        assert len(want) > 10_000 and got == want


def test_filter_with_a_hash_list_of_100k_kmers_matches_the_reference_binary(tmp_path):
    """Real hash lists hold 10^5 .. 10^6 k-mers (recurrent sequencing errors), not the 500 of the synthetic trio: the paired
    tool then runs the pair filter in its mask-only mode with the candidate queue draining all the time -- where round 6
    had set bits of pairs without a hit (tests/test_gpu_parity.py::test_filter_mask_only_equals_counts_on_large_sets).
    Byte-identical to the reference's own binary (one thread: its output order is then input order)."""
    from tests.conftest import require_ref
    ref = require_ref("RUFUS.Filter")
    from rufus_amd import capi
    d = str(tmp_path)
    n_pairs, G = 120_000, 6_000_000
    sy = capi.Synth.sample(G, 0, n_snv=20, seed=11)
    seq, qual = sy.text(0, n_pairs)
    m1 = b"".join(b"@r%d/1\n%s\n+\n%s\n" % (r, seq[2 * r].tobytes(), qual[2 * r].tobytes()) for r in range(n_pairs))
    m2 = b"".join(b"@r%d/2\n%s\n+\n%s\n" % (r, seq[2 * r + 1].tobytes(), qual[2 * r + 1].tobytes()) for r in range(n_pairs))
    rng = np.random.default_rng(12)
    genome = np.frombuffer(sy.genome(0, G), np.uint8)
    loci = rng.integers(0, G - 200, 40)
    kmers = [bytes(genome[p0 + i:p0 + i + 25]) for p0 in loci for i in range(100)]
    kmers += ["".join(x).encode() for x in rng.choice(list("ACGT"), (110_000, 25))]
    open(f"{d}/hl", "wb").write(b"".join(km + b" 9\n" for km in kmers))
    open(f"{d}/m1.fq", "wb").write(m1)
    open(f"{d}/m2.fq", "wb").write(m2)
    r = sh([ref, "hl", "m1.fq", "m2.fq", "ref", "25", "15", "1", "1"], d, timeout=900)
    assert r.returncode == 0
    for rep in range(3):
        r = subprocess.run([f"{BIN}/RUFUS.Filter", "hl", "m1.fq", "m2.fq", "out", "25", "15", "1", "5"], cwd=d,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert r.returncode == 0, r.stderr
        for m in (1, 2):
            got = open(f"{d}/out.Mutations.Mate{m}.fastq", "rb").read()
            want = open(f"{d}/ref.Mutations.Mate{m}.fastq", "rb").read()
            assert len(want) > 10_000 and got == want


@pytest.mark.parametrize("route", ["pipe", "file"])
def test_count_sam_input_equals_passthrough_plus_count(testrun, tmp_path, route):
    """`jellyfish count --sam X.chr` on SAM text = `PassThroughSamCheck X.chr | jellyfish count`
    (scripts/RunJellyForRUFUS.sh:28-29) without the FASTQ text in between: same .Jhash payload, same chromosome log
    (also the reference binary's), with the stream cut into many pieces handled by different threads."""
    d = str(tmp_path)
    rng = np.random.default_rng(11)
    lines = []
    chrs = [b"chr1", b"chr1", b"chr2", b"chr10", b"chr1", b"chrX", b"*"]
    for m, text in enumerate(testrun["Child"]):
        recs = text.split(b"\n")
        for i in range(0, len(recs) - 1, 4):
            c = chrs[min(len(chrs) - 1, (i // 4) * len(chrs) // (len(recs) // 4))] if m == 0 else chrs[int(rng.integers(0, 3))]
            lines.append(b"\t".join([recs[i][1:], b"99", c, b"%d" % (i + 1), b"60", b"100M", b"=", b"1", b"0", recs[i + 1],
                                     recs[i + 3], b"NM:i:0"]))
    sam = b"\n".join(lines) + b"\n"
    open(f"{d}/in.sam", "wb").write(sam)
    r = subprocess.run(f"{BIN}/PassThroughSamCheck a.chr < in.sam | {BIN}/jellyfish count --disk -m 25 -L 2 -s 100M -t 4 "
                       f"-o a.Jhash -C /dev/stdin", shell=True, cwd=d, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, RFX_INGEST_PIECE="65536")
    cmd = [f"{BIN}/jellyfish", "count", "--sam", "b.chr", "--disk", "-m", "25", "-L", "2", "-s", "100M", "-t", "6", "-o",
           "b.Jhash", "-C"]
    if route == "pipe":
        r = subprocess.run(cmd + ["/dev/stdin"], cwd=d, env=env, input=sam, stderr=subprocess.PIPE)
    else:
        r = subprocess.run(cmd + ["in.sam"], cwd=d, env=env, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr
    assert _payload(f"{d}/a.Jhash") == _payload(f"{d}/b.Jhash") and len(_payload(f"{d}/b.Jhash")) > 100_000
    # ... and directly the oracle's count of field 10 of every line (jf mer_overlap_sequence_parser semantics: the
    # reads are independent sequences), not only the repo's own two-process route
    import oracle
    want = oracle.count(None, 25, 100_000_000, lower=2, reads=[ln.split(b"\t")[9] for ln in lines])
    assert _payload(f"{d}/b.Jhash") == want.payload()
    assert open(f"{d}/a.chr").read() == open(f"{d}/b.chr").read()
    assert open(f"{d}/b.chr").read().split() == ["notachr", "chr1", "chr2", "chr10", "chr1", "chrX", "*"] + \
        open(f"{d}/b.chr").read().split()[7:]
    ref = os.path.join(ROOT, "oracle", "_ref", "PassThroughSamCheck")
    if os.path.exists(ref):
        subprocess.run(f"{ref} ref.chr < in.sam > /dev/null", shell=True, cwd=d, check=True)
        assert open(f"{d}/ref.chr").read() == open(f"{d}/b.chr").read()
    # a header line is not a record (the reference tool would read past the end of the line)
    r = subprocess.run(cmd + ["/dev/stdin"], cwd=d, env=env, input=b"@HD\tVN:1.6\n" + sam, stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"--sam" in r.stderr


@pytest.mark.parametrize("shape", ["sorted", "shuffled"])
def test_filter_sam_equals_feeder_plus_filter(tmp_path, shape):
    """SURVEY 8 row N1, filter half: `RUFUS.Filter --sam CHR HashList stdin STUB ...` on the SAM stream = the
    two-process route of runRufus.sh:964-967 (`PassThroughSamCheck.stranded CHR STUB.temp` -> two FASTQ streams ->
    `RUFUS.Filter`), byte for byte: Mutations.Mate1/2.fastq and the chromosome log -- against the drop-in pair AND against
    the reference binaries (oracle/_ref, one thread: input order).  "shuffled": records in random order, names seen
    three and four times, IUPAC / lower-case bases in reverse-strand reads, qualities shorter than the read, so that
    thousands of reads wait across many pieces (pieces of 16 KB: the waiting records outlive their piece)."""
    from tests.test_cli_host import make_sam
    d = str(tmp_path)
    rng = np.random.default_rng(5)
    if shape == "sorted":
        sam = make_sam(3000, seed=8)
    else:
        n = 6000
        alphabet = np.frombuffer(b"ACGT" * 8 + b"Nacgtn" + b"RY", np.uint8)
        recs = []
        for i in range(n):
            times = 2 if i % 97 else (3 if i % 2 else 4)
            for t in range(times):
                L = int(rng.integers(30, 151))
                seq = bytes(rng.choice(alphabet, L))
                qual = rng.integers(55, 75, L if i % 53 else max(1, L - 7), dtype=np.uint8)
                qual[rng.random(len(qual)) < 0.03] = 35                # '#': below MinQ
                qual = bytes(qual)
                flag = int(rng.choice([99, 147, 83, 163, 16, 0, 1040, 65]))
                recs.append(b"\t".join([b"q%d" % i, b"%d" % flag, b"chr%d" % (1 + i % 5), b"%d" % (i + t), b"60", b"*", b"=",
                                         b"1", b"0", seq, qual, b"XS:i:%d" % t]) + b"\n")
        sam = b"".join(recs[j] for j in rng.permutation(len(recs)))
    open(f"{d}/in.sam", "wb").write(sam)
    # hash list: k-mers of the (upper-case ACGT) stretches of some records, either strand
    lines = [ln.split(b"\t") for ln in sam.split(b"\n") if ln.count(b"\t") >= 10]
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    kmers = []
    for j in rng.choice(len(lines), 200 if shape == "sorted" else 1500, replace=False):
        sq = lines[j][9]
        ok = [a for a in range(len(sq) - 24) if set(sq[a:a + 25]) <= set(b"ACGT")]
        for a in (rng.choice(ok, min(3, len(ok)), replace=False) if ok else ()):
            km = sq[int(a):int(a) + 25]
            kmers.append(km if rng.random() < 0.5 else km.translate(comp)[::-1])
    assert len(kmers) > 40
    open(f"{d}/hl", "wb").write(b"".join(km + b" 9\n" for km in kmers))
    env = dict(os.environ, RFX_INGEST_PIECE="16384")
    r = subprocess.run(f"{BIN}/PassThroughSamCheck.stranded two.chr two < in.sam > two.log && "
                       f"{BIN}/RUFUS.Filter hl two.mate1.fastq two.mate2.fastq two 25 15 1 4", shell=True, cwd=d,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr
    for threads in ("1", "5"):
        r = subprocess.run([f"{BIN}/RUFUS.Filter", "--sam", "one.chr", "hl", "stdin", "one", "25", "15", "1", threads], cwd=d, env=env,
                           input=sam, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert r.returncode == 0, r.stderr
        for m in (1, 2):
            got = open(f"{d}/one.Mutations.Mate{m}.fastq", "rb").read()
            assert got == open(f"{d}/two.Mutations.Mate{m}.fastq", "rb").read() and got.count(b"\n") >= 4 * 15
        assert open(f"{d}/one.chr", "rb").read() == open(f"{d}/two.chr", "rb").read()
    # ... and from a regular file (mapped, cut at line ends without a copy; RFX_FILTER_NO_MMAP: read like a pipe)
    for e in (env, dict(env, RFX_FILTER_NO_MMAP="1")):
        r = subprocess.run([f"{BIN}/RUFUS.Filter", "--sam", "file.chr", "hl", "in.sam", "file", "25", "15", "1", "3"], cwd=d, env=e,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert r.returncode == 0, r.stderr
        for m in (1, 2):
            assert open(f"{d}/file.Mutations.Mate{m}.fastq", "rb").read() == open(f"{d}/two.Mutations.Mate{m}.fastq", "rb").read()
        assert open(f"{d}/file.chr", "rb").read() == open(f"{d}/two.chr", "rb").read()
    ref = os.path.join(ROOT, "oracle", "_ref")
    if os.path.exists(f"{ref}/RUFUS.Filter") and shape == "sorted":     # (the reference filter needs whole quality lines)
        r = subprocess.run(f"{ref}/PassThroughSamCheck.stranded ref.chr ref < in.sam > ref.log && "
                           f"{ref}/RUFUS.Filter hl ref.mate1.fastq ref.mate2.fastq ref 25 15 1 1 > ref.flog", shell=True, cwd=d,
                           stderr=subprocess.PIPE, timeout=300)
        assert r.returncode == 0, r.stderr
        for m in (1, 2):
            assert open(f"{d}/one.Mutations.Mate{m}.fastq", "rb").read() == open(f"{d}/ref.Mutations.Mate{m}.fastq", "rb").read()
        assert open(f"{d}/one.chr", "rb").read() == open(f"{d}/ref.chr", "rb").read()


@pytest.mark.parametrize("gpus,k,size", [("0,0", 25, "100M"), ("0,0,0,0,0", 31, "8G"), ("0-0,0,0", 25, "8G")])
def test_executables_spread_a_sample_over_several_devices(testrun, tmp_path, gpus, k, size):
    """Row E-cli (runRufus.sh:776-797 calls binaries: the GPUs of a node must be reachable from them): with
    RUFUS_GPUS naming n devices -- here n contexts on the box's one GPU, the same code path -- `jellyfish count`
    deals the read blocks to the devices in turn, each partitions its blocks, the owners of the minimizer bins pull
    their records, the survivors change hands by output position and the .Jhash is the devices' slices one after the
    other: the bytes of the one-device run (and of the golden payload).  `RUFUS.Filter` deals its pieces to the devices: the same Mutations.Mate1/2.fastq."""
    d = str(tmp_path)
    exp = testrun["expected"]
    open(f"{d}/c.fq", "wb").write(testrun["Child"][0] + testrun["Child"][1])
    # (small ingest pieces: the sample comes as several read blocks, dealt to the devices in turn)
    env = dict(os.environ, RUFUS_GPUS=gpus, RFX_COUNT_HISTO="1", RFX_INGEST_PIECE="200000")
    for name, e in (("one", dict(os.environ, RFX_COUNT_HISTO="1")), ("many", env), ("repl", dict(env, RFX_PEERS_REPLICATE="1"))):
        r = subprocess.run([f"{BIN}/jellyfish", "count", "--disk", "-m", str(k), "-L", "2", "-s", size, "-t", "4", "-o",
                            f"{name}.Jhash", "-C", "c.fq"], cwd=d, env=e, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr
    assert _payload(f"{d}/one.Jhash") == _payload(f"{d}/many.Jhash") and len(_payload(f"{d}/many.Jhash")) > 100_000
    assert _payload(f"{d}/one.Jhash") == _payload(f"{d}/repl.Jhash")      # (round 3's scheme: every device every block)
    assert open(f"{d}/one.Jhash.histo", "rb").read() == open(f"{d}/many.Jhash.histo", "rb").read()
    if (k, size) == (25, "100M"):
        assert hashlib.sha256(_payload(f"{d}/many.Jhash")).hexdigest() == exp["samples"]["Child"]["s100M"]["payload_sha256"]
    # a pipe as the sample (scripts/RunJellyForRUFUS.sh:23-31) and as the output
    r = subprocess.run(f"cat c.fq | {BIN}/jellyfish count --disk -m {k} -L 2 -s {size} -t 3 -o /dev/stdout -C /dev/stdin | cat > piped.Jhash",
                       shell=True, cwd=d, env=env, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr
    assert _payload(f"{d}/piped.Jhash") == _payload(f"{d}/one.Jhash")
    # a database that is a pipe (process substitution): opened once, header and payload in one pass
    h = subprocess.run(["bash", "-c", f"{BIN}/jellyfish histo -f <(cat one.Jhash)"], cwd=d, stdout=subprocess.PIPE).stdout
    assert h == open(f"{d}/one.Jhash.histo", "rb").read()
    q = subprocess.run(["bash", "-c", f"{BIN}/jellyfish query <(cat one.Jhash) " + "A" * k], cwd=d, stdout=subprocess.PIPE).stdout
    assert q == subprocess.run([f"{BIN}/jellyfish", "query", "one.Jhash", "A" * k], cwd=d, stdout=subprocess.PIPE).stdout and q
    if k == 25:
        open(f"{d}/hl", "w").write(testrun["hashlist"])
        for i, m in enumerate(testrun["Child"]):
            open(f"{d}/m{i + 1}.fq", "wb").write(m)
        for name, e in (("f1", os.environ), ("fn", dict(os.environ, RUFUS_GPUS=gpus))):
            r = subprocess.run([f"{BIN}/RUFUS.Filter", "hl", "m1.fq", "m2.fq", name, "25", "15", "1", "6"], cwd=d, env=e,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert r.returncode == 0, r.stderr
        for m in (1, 2):
            a = open(f"{d}/fn.Mutations.Mate{m}.fastq", "rb").read()
            assert a == open(f"{d}/f1.Mutations.Mate{m}.fastq", "rb").read() and a.count(b"\n") == 4 * 26


def test_query_looks_a_kmer_list_up_in_several_databases_at_once(testrun, tmp_path):
    """SURVEY row N3 (scripts/Overlap.shorter.sh:265-299: the same k-mer lists looked up in the subject's and every
    control's database, one `jellyfish query` process each): `jellyfish query -s FA -o O1 -o O2 -o O3 DB1 DB2 DB3`
    writes to Oi the bytes `jellyfish query -s FA DBi` prints -- with the databases cut into many position ranges too --,
    the counts are the oracle's, and without -o the counts come as columns."""
    import oracle
    d = str(tmp_path)
    names = ("Child", "Mother", "Father")
    for s in names:
        open(f"{d}/{s}.fq", "wb").write(testrun[s][0] + testrun[s][1])
        r = sh([f"{BIN}/jellyfish", "count", "-m", "25", "-L", "2", "-s", "100M", "-t", "4", "-o", f"{s}.Jhash", "-C", f"{s}.fq"], d)
        assert r.returncode == 0, r.stderr
    rng = np.random.default_rng(2)
    kmers = [ln.split()[0] for ln in testrun["merge"].splitlines()][:200]
    recs = testrun["Mother"][0].split(b"\n")
    kmers += [recs[4 * i + 1][10:35].decode() for i in range(300) if b"N" not in recs[4 * i + 1][10:35]]
    kmers += ["".join(rng.choice(list("ACGT"), 25)) for _ in range(200)]
    open(f"{d}/q.fa", "w").write("".join(f">{i}\n{km}\n" for i, km in enumerate(kmers)))
    single = {}
    for s in names:
        r = sh([f"{BIN}/jellyfish", "query", "-s", "q.fa", f"{s}.Jhash"], d)
        assert r.returncode == 0, r.stderr
        single[s] = r.stdout
        assert single[s].count(b"\n") == len(kmers)
    for env in (os.environ, dict(os.environ, RFX_QUERY_SLICE_RECORDS="777")):
        r = subprocess.run([f"{BIN}/jellyfish", "query", "-s", "q.fa", "-o", "o1", "-o", "o2", "-o", "o3"] +
                           [f"{s}.Jhash" for s in names], cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0 and r.stdout == b"", r.stderr
        for i, s in enumerate(names):
            assert open(f"{d}/o{i + 1}", "rb").read() == single[s]
    r = sh([f"{BIN}/jellyfish", "query", "-s", "q.fa"] + [f"{s}.Jhash" for s in names] + ["A" * 25], d)
    assert r.returncode == 0, r.stderr
    cols = [ln.split() for ln in r.stdout.decode().splitlines()]
    assert len(cols) == len(kmers) + 1 and all(len(c) == 4 for c in cols)
    for j, s in enumerate(names):
        want = [ln.split()[1] for ln in single[s].decode().splitlines()]
        assert [c[1 + j] for c in cols[:-1]] == want
        rec = oracle.count([testrun[s][0] + testrun[s][1]], 25, 100_000_000, lower=2)
        table = dict(zip(rec.keys.tolist(), rec.counts.tolist()))
        for c in cols[:-1]:
            key = min(oracle.jf_encode(c[0]), oracle.jf_encode(c[0].translate(str.maketrans("ACGT", "TGCA"))[::-1]))
            assert int(c[1 + j]) == table.get(key, 0)
    assert sum(int(c[1]) > 0 for c in cols) > 200
    # few k-mers against a big database: the records at the queried positions only (located on the host, looked up on the
    # device) -- the lines of the walk over position ranges
    open(f"{d}/few.fa", "w").write("".join(f">{i}\n{km}\n" for i, km in enumerate(kmers[:40] + kmers[-5:])))
    a = subprocess.run([f"{BIN}/jellyfish", "query", "-s", "few.fa"] + [f"{s}.Jhash" for s in names], cwd=d,
                       env=dict(os.environ, RFX_QUERY_SPARSE_RATIO="64"), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    b = subprocess.run([f"{BIN}/jellyfish", "query", "-s", "few.fa"] + [f"{s}.Jhash" for s in names], cwd=d,
                       env=dict(os.environ, RFX_QUERY_NO_SPARSE="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert a.returncode == 0 and b.returncode == 0 and a.stdout == b.stdout and a.stdout.count(b"\n") == 45, a.stderr
    assert [ln.split() for ln in a.stdout.decode().splitlines()] == cols[:40] + cols[-6:-1]
    # A database that is PIPED (loaded whole, never sliced) and has another table size, between two regular ones of one
    # size: the third must not be looked up with positions computed for a hash function it does not have (ADVICE r3:
    # the recompute test compared with the database before, which had never touched the positions).
    r = sh([f"{BIN}/jellyfish", "count", "-m", "25", "-L", "2", "-s", "1M", "-t", "4", "-o", "Mother.small.Jhash", "-C", "Mother.fq"], d)
    assert r.returncode == 0, r.stderr
    os.mkfifo(f"{d}/pipe.Jhash")
    feeder = subprocess.Popen(["sh", "-c", "cat Mother.small.Jhash > pipe.Jhash"], cwd=d)
    r = subprocess.run([f"{BIN}/jellyfish", "query", "-s", "q.fa", "-o", "p1", "-o", "p2", "-o", "p3", "Child.Jhash", "pipe.Jhash",
                        "Father.Jhash"], cwd=d, env=dict(os.environ, RFX_QUERY_SLICE_RECORDS="777"), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE)
    assert feeder.wait() == 0 and r.returncode == 0, r.stderr
    assert open(f"{d}/p1", "rb").read() == single["Child"] and open(f"{d}/p3", "rb").read() == single["Father"]
    assert open(f"{d}/p2", "rb").read() == single["Mother"]     # (counts do not depend on the table size)
    # an argument spelt like a mer stays a mer, even if a file of that name exists
    open(f"{d}/{'A' * 25}", "w").write("x")
    r = sh([f"{BIN}/jellyfish", "query", "Child.Jhash", "A" * 25], d)
    assert r.returncode == 0 and r.stdout.decode().split()[0] == "A" * 25, r.stderr


def test_subject_stream_is_ingested_once_through_a_spool(tmp_path):
    """the subject's generator runs ONCE.  `jellyfish count --sam CHR --spool FILE` counts the SAM pipe
    (scripts/RunJellyForRUFUS.sh:28) and leaves its bytes in FILE (written piece by piece by the parser threads);
    `RUFUS.Filter --sam CHR HashList FILE ...` then reads FILE instead of a second `samtools view` (runRufus.sh:966).
    The spool is the stream byte for byte, the count is the no-spool count, and the pulled pairs are those of the
    reference route (feeder -> two FASTQ streams -> filter) on a second run of the generator."""
    from tests.test_cli_host import make_sam
    d = str(tmp_path)
    sam = make_sam(4000, seed=21)
    open(f"{d}/in.sam", "wb").write(sam)
    env = dict(os.environ, RFX_INGEST_PIECE="20000")
    cmd = [f"{BIN}/jellyfish", "count", "--disk", "-m", "25", "-L", "2", "-s", "100M", "-t", "5", "-C"]
    r = subprocess.run(cmd + ["--sam", "a.chr", "--spool", "spool.sam", "-o", "a.Jhash", "/dev/stdin"], cwd=d, env=env, input=sam,
                       stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr
    assert open(f"{d}/spool.sam", "rb").read() == sam
    r = subprocess.run(cmd + ["--sam", "b.chr", "-o", "b.Jhash", "in.sam"], cwd=d, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr
    assert _payload(f"{d}/a.Jhash") == _payload(f"{d}/b.Jhash") and open(f"{d}/a.chr").read() == open(f"{d}/b.chr").read()
    lines = [ln.split(b"\t") for ln in sam.split(b"\n") if ln.count(b"\t") >= 10]
    rng = np.random.default_rng(4)
    kmers = [lines[j][9][30:55] for j in rng.choice(len(lines), 40, replace=False) if set(lines[j][9][30:55]) <= set(b"ACGT")]
    open(f"{d}/hl", "wb").write(b"".join(km + b" 9\n" for km in kmers))
    r = subprocess.run([f"{BIN}/RUFUS.Filter", "--sam", "f.chr", "hl", "spool.sam", "one", "25", "15", "1", "4"], cwd=d,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr
    # the reference route: the REFERENCE's stranded feeder and filter (oracle/_ref, one thread: input order)
    from tests.conftest import require_ref
    ref = os.path.dirname(require_ref("RUFUS.Filter"))
    r = subprocess.run(f"{ref}/PassThroughSamCheck.stranded two.chr two < in.sam > two.log && "
                       f"{ref}/RUFUS.Filter hl two.mate1.fastq two.mate2.fastq two 25 15 1 1", shell=True, cwd=d,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr
    for m in (1, 2):
        got = open(f"{d}/one.Mutations.Mate{m}.fastq", "rb").read()
        assert got == open(f"{d}/two.Mutations.Mate{m}.fastq", "rb").read() and got.count(b"\n") >= 4 * 15


def test_subject_is_parsed_once_and_filtered_from_the_packed_cache(tmp_path):
    """SURVEY 8 row N2, for real: `jellyfish count --sam CHR --spool SPOOL --keep-packed CACHE` packs every record once --
    as RUFUS.Filter would see it -- while it counts; `RUFUS.Filter --packed CACHE CHR HashList SPOOL ...` scans the
    cache and touches text only for the names with a hit (scripts/RunJellyForRUFUS.sh:28 + runRufus.sh:966 without the
    second parse).  Mutations.Mate1/2 and the chromosome log are byte for byte those of the text route (`--sam`) and of
    the reference route (stranded feeder -> two FASTQ files -> filter) -- on a stream with reverse-strand records,
    names that come three times, a record whose bases the feeder would drop (IUPAC), a short quality string, records
    out of coordinate order; the count is the count without the cache.  A cache packed for another MinQ is not used."""
    from tests.test_cli_host import make_sam
    d = str(tmp_path)
    rng = np.random.default_rng(11)
    lines = [ln for ln in make_sam(3000, seed=23).split(b"\n") if ln]
    f = [ln.split(b"\t") for ln in lines]
    # awkward records: an IUPAC base on a reverse-strand record, a quality string shorter than its read, a third record
    # of a name, an empty mate far away in the stream
    for j in (40, 41, 900):
        if int(f[j][1]) & 16:
            f[j][9] = f[j][9][:70] + b"R" + f[j][9][71:]
    f[100][10] = f[100][10][:90]
    f.append(list(f[200]))
    f.append([b"lonely"] + f[300][1:])
    order = np.arange(len(f))
    sub = order[1000:1400].copy()
    rng.shuffle(sub)
    order[1000:1400] = sub
    sam = b"".join(b"\t".join(f[i]) + b"\n" for i in order)
    open(f"{d}/in.sam", "wb").write(sam)
    env = dict(os.environ, RFX_INGEST_PIECE="40000")
    cmd = [f"{BIN}/jellyfish", "count", "--disk", "-m", "25", "-L", "2", "-s", "100M", "-t", "5", "-C"]
    r = subprocess.run(cmd + ["--sam", "a.chr", "--spool", "spool.sam", "--keep-packed", "cache.bin", "-o", "a.Jhash", "/dev/stdin"],
                       cwd=d, env=env, input=sam, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr
    assert open(f"{d}/spool.sam", "rb").read() == sam and os.path.getsize(f"{d}/cache.bin") > len(lines) * 60
    r = subprocess.run(cmd + ["--sam", "b.chr", "-o", "b.Jhash", "in.sam"], cwd=d, stderr=subprocess.PIPE)
    assert r.returncode == 0 and _payload(f"{d}/a.Jhash") == _payload(f"{d}/b.Jhash")
    # the cache of a SAM FILE (no pipe, no spool: the file is its own spool)
    r = subprocess.run(cmd + ["--sam", "c.chr", "--keep-packed", "cache2.bin", "-o", "c.Jhash", "in.sam"], cwd=d, env=env, stderr=subprocess.PIPE)
    assert r.returncode == 0 and _payload(f"{d}/c.Jhash") == _payload(f"{d}/b.Jhash")
    fields = [ln.split(b"\t") for ln in sam.split(b"\n") if ln.count(b"\t") >= 10]
    kmers = [fields[j][9][30:55] for j in rng.choice(len(fields), 60, replace=False) if set(fields[j][9][30:55]) <= set(b"ACGT")]
    kmers += [fields[j][9][60:85] for j in (40, 100) if set(fields[j][9][60:85]) <= set(b"ACGT")]
    open(f"{d}/hl", "wb").write(b"".join(km + b" 9\n" for km in kmers))
    runs = {"packed": ["--packed", "cache.bin", "packed.chr", "hl", "spool.sam", "packed", "25", "15", "1", "4"],
            "packed2": ["--packed", "cache2.bin", "packed2.chr", "hl", "in.sam", "packed2", "25", "15", "1", "4"],
            "otherq": ["--packed", "cache.bin", "otherq.chr", "hl", "spool.sam", "otherq", "25", "20", "1", "4"],
            "text": ["--sam", "text.chr", "hl", "spool.sam", "text", "25", "15", "1", "4"],
            "text20": ["--sam", "text20.chr", "hl", "spool.sam", "text20", "25", "20", "1", "4"]}
    out = {}
    for name, a in runs.items():
        r = subprocess.run([f"{BIN}/RUFUS.Filter"] + a, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert r.returncode == 0, r.stderr
        out[name] = r
    r = subprocess.run(f"{BIN}/PassThroughSamCheck.stranded ref.chr ref < in.sam > ref.log && "
                       f"{BIN}/RUFUS.Filter hl ref.mate1.fastq ref.mate2.fastq ref 25 15 1 4", shell=True, cwd=d,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr
    # .. and the REFERENCE's own feeder + filter (oracle/_ref, one thread: input order) on the same stream without the
    # record whose quality string is shorter than its read -- there src/RUFUS.Filter.cpp:205 indexes past the string's end
    refdir = os.path.join(ROOT, "oracle", "_ref")
    if os.path.exists(f"{refdir}/RUFUS.Filter"):
        f_ok = [list(x) for x in f]
        f_ok[100][10] = f_ok[100][10] + f_ok[100][10][:len(f_ok[100][9]) - len(f_ok[100][10])]
        open(f"{d}/ok.sam", "wb").write(b"".join(b"\t".join(f_ok[i]) + b"\n" for i in order))
        r = subprocess.run(f"{refdir}/PassThroughSamCheck.stranded okref.chr okref < ok.sam > okref.log && "
                           f"{refdir}/RUFUS.Filter hl okref.mate1.fastq okref.mate2.fastq okref 25 15 1 1 && "
                           f"{BIN}/RUFUS.Filter --sam oktext.chr hl ok.sam oktext 25 15 1 4", shell=True, cwd=d,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr
        for m in (1, 2):
            assert open(f"{d}/oktext.Mutations.Mate{m}.fastq", "rb").read() == open(f"{d}/okref.Mutations.Mate{m}.fastq", "rb").read()
        assert open(f"{d}/oktext.chr", "rb").read() == open(f"{d}/okref.chr", "rb").read()
    # a cache its producer did not finish (no "done" word in the header), and the cache of ANOTHER stream of the same length
    # (two lines swapped): neither is used -- the text route gives the same files
    blob = bytearray(open(f"{d}/cache.bin", "rb").read())
    blob[16:24] = bytes(8)
    open(f"{d}/undone.bin", "wb").write(blob)
    swapped = list(order)
    same = [i for i in range(len(swapped)) if len(b"\t".join(f[swapped[i]])) == len(b"\t".join(f[swapped[500]]))]
    assert len(same) > 100                      # the lines of one length, rotated by one place: every offset stays a line start
    for x, y in zip(same, same[1:] + same[:1]):
        swapped[x] = int(order[y])
    sam2 = b"".join(b"\t".join(f[i]) + b"\n" for i in swapped)
    assert len(sam2) == len(sam) and sam2 != sam
    open(f"{d}/other.sam", "wb").write(sam2)
    r = subprocess.run(cmd + ["--sam", "o.chr", "--keep-packed", "other.bin", "-o", "o.Jhash", "other.sam"], cwd=d, env=env, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr
    for name, cache in (("undone", "undone.bin"), ("other", "other.bin")):
        r = subprocess.run([f"{BIN}/RUFUS.Filter", "--packed", cache, f"{name}.chr", "hl", "spool.sam", name, "25", "15", "1", "4"], cwd=d,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert r.returncode == 0 and b"scanning the text" in r.stderr, r.stderr
        for m in (1, 2):
            assert open(f"{d}/{name}.Mutations.Mate{m}.fastq", "rb").read() == open(f"{d}/text.Mutations.Mate{m}.fastq", "rb").read()
    for m in (1, 2):
        want = open(f"{d}/text.Mutations.Mate{m}.fastq", "rb").read()
        assert want == open(f"{d}/ref.Mutations.Mate{m}.fastq", "rb").read() and want.count(b"\n") >= 4 * 20
        assert open(f"{d}/packed.Mutations.Mate{m}.fastq", "rb").read() == want
        assert open(f"{d}/packed2.Mutations.Mate{m}.fastq", "rb").read() == want
        assert open(f"{d}/otherq.Mutations.Mate{m}.fastq", "rb").read() == open(f"{d}/text20.Mutations.Mate{m}.fastq", "rb").read()
    for name in ("packed", "packed2", "otherq"):
        assert open(f"{d}/{name}.chr", "rb").read() == open(f"{d}/text.chr", "rb").read() == open(f"{d}/ref.chr", "rb").read()
    # the scan ran on the cache, and only a fraction of the lines was touched as text
    msg = out["packed"].stdout.decode()
    got = [int(x) for x in msg.split("packed cache: ")[1].replace(",", " ").replace(";", " ").split() if x.isdigit()]
    assert got[0] == len(fields) and 0 < got[3] < len(fields) // 2 and got[2] >= 3, msg    # (a 20 kb genome: every k-mer hits ~17 reads)
    assert b"not a usable packed-read cache" in out["otherq"].stderr and b"packed cache:" not in out["otherq"].stdout


def test_round3_tools_on_edge_inputs(testrun, tmp_path):
    """Corners of the round-3 additions: `RUFUS.Filter --sam` on an empty stream, on lines with too few fields, with
    HashCountThreshold 2 (both against the two-process route); `jellyfish count --spool` on a FASTQ pipe; one of
    several `jellyfish query` databases given as a pipe; --spool refused where it cannot work."""
    from tests.test_cli_host import make_sam
    d = str(tmp_path)
    sam = make_sam(1500, seed=33)
    lines = [ln.split(b"\t") for ln in sam.split(b"\n") if ln.count(b"\t") >= 10]
    kmers = []
    for f in lines[::40]:
        sq = f[9]
        kmers += [sq[a:a + 25] for a in (20, 21, 60) if set(sq[a:a + 25]) <= set(b"ACGT") and len(sq) >= a + 25]
    open(f"{d}/hl", "wb").write(b"".join(km + b" 9\n" for km in kmers))
    broken = sam + b"too\tfew\tfields\n" + b"x\t0\tchr1\t1\n"
    for name, data, thr in (("empty", b"", "1"), ("t2", broken, "2")):
        open(f"{d}/{name}.sam", "wb").write(data)
        r = subprocess.run([f"{BIN}/RUFUS.Filter", "--sam", f"{name}.chr", "hl", "stdin", f"{name}", "25", "15", thr, "3"], cwd=d,
                           input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert r.returncode == 0, r.stderr
        r = subprocess.run(f"{BIN}/PassThroughSamCheck.stranded {name}2.chr {name}2 < {name}.sam > /dev/null && "
                           f"{BIN}/RUFUS.Filter hl {name}2.mate1.fastq {name}2.mate2.fastq {name}2 25 15 {thr} 3", shell=True, cwd=d,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert r.returncode == 0, r.stderr
        for m in (1, 2):
            assert open(f"{d}/{name}.Mutations.Mate{m}.fastq", "rb").read() == open(f"{d}/{name}2.Mutations.Mate{m}.fastq", "rb").read()
        assert open(f"{d}/{name}.chr", "rb").read() == open(f"{d}/{name}2.chr", "rb").read()
    assert open(f"{d}/empty.chr").read() == "notachr\n" and os.path.getsize(f"{d}/empty.Mutations.Mate1.fastq") == 0
    assert open(f"{d}/t2.Mutations.Mate1.fastq", "rb").read().count(b"\n") // 4 > 0
    # --spool on a FASTQ pipe
    fq = testrun["Mother"][0] + testrun["Mother"][1]
    cmd = [f"{BIN}/jellyfish", "count", "-m", "25", "-L", "2", "-s", "100M", "-t", "4", "-C"]
    r = subprocess.run(cmd + ["--spool", "m.spool", "-o", "m1.Jhash", "/dev/stdin"], cwd=d, input=fq, stderr=subprocess.PIPE,
                       env=dict(os.environ, RFX_INGEST_PIECE="30000"))
    assert r.returncode == 0, r.stderr
    assert open(f"{d}/m.spool", "rb").read() == fq
    open(f"{d}/m.fq", "wb").write(fq)
    r = subprocess.run(cmd + ["-o", "m2.Jhash", "m.fq"], cwd=d, stderr=subprocess.PIPE)
    assert r.returncode == 0 and _payload(f"{d}/m1.Jhash") == _payload(f"{d}/m2.Jhash")
    r = subprocess.run(cmd + ["--spool", "x.spool", "-o", "m3.Jhash", "m.fq"], cwd=d, stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"--spool" in r.stderr
    # several databases, one of them a pipe
    q = subprocess.run(["bash", "-c", f"{BIN}/jellyfish query m1.Jhash <(cat m2.Jhash) " + fq.split(b"\n")[1][:25].decode()], cwd=d,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert q.returncode == 0, q.stderr
    f = q.stdout.split()
    assert len(f) == 3 and f[1] == f[2] and int(f[1]) >= 2


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


def _payload(path):
    blob = open(path, "rb").read()
    return blob[9 + int(blob[:9]):]


def test_count_cli_parallel_ingest_and_shard_passes(tmp_path):
    """`jellyfish count` with worker threads: a mapped file, a named pipe (one reader cutting the stream), the
    deferred mode (packed reads resident, shard passes inside the table: what a 30x sample takes) and the
    sequential parser all write the same payload -- the oracle's.  A FASTQ without a final newline and blank
    lines between records are tolerated; a multi-line FASTQ falls back to the sequential parser."""
    import numpy as np
    import oracle
    from rufus_amd import capi
    d = str(tmp_path)
    n_pairs = 40_000
    r = sh([f"{BIN}/rfx_synth_fastq", "400000", "0", "12", "4242", "0", str(n_pairs), "reads.fq"], d)
    assert r.returncode == 0, r.stderr
    sy = capi.Synth.sample(400_000, 0, n_snv=12, seed=4242)
    seq, _ = sy.text(0, n_pairs)
    ref = oracle.count(None, 25, 8 << 30, lower=2, reads=[x.tobytes() for x in seq]).payload()
    args = [f"{BIN}/jellyfish", "count", "--disk", "-m", "25", "-L", "2", "-s", "8G", "-C"]
    env = dict(os.environ)

    def run(out, src, extra_env=None, t="8"):
        e = dict(env)
        e.update(extra_env or {})
        p = subprocess.run(args + ["-t", t, "-o", out, src], cwd=d, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           timeout=300)
        assert p.returncode == 0, p.stderr
        return _payload(f"{d}/{out}")

    assert run("mapped.jf", "reads.fq") == ref
    assert run("seq.jf", "reads.fq", t="1") == ref
    # the output's pages allocated ahead of time (inputs > 256 MB only, unless told otherwise): a guess far too
    # large is cut back to the exact size, one far too small is topped up -- same file either way
    whole = open(f"{d}/mapped.jf", "rb").read()
    for frac in ("3.0", "0.001"):
        assert run("pre.jf", "reads.fq", {"RFX_PREALLOC_MIN": "0", "RFX_PREALLOC_FRAC": frac}) == ref
        blob = open(f"{d}/pre.jf", "rb").read()
        assert len(blob) == 9 + int(blob[:9]) + len(ref)
        assert os.stat(f"{d}/pre.jf").st_blocks * 512 < len(whole) + (1 << 20)     # nothing left allocated past the end
    assert run("nopre.jf", "reads.fq", {"RFX_NO_PREALLOC": "1", "RFX_CLEAN_EXIT": "1"}) == ref
    # (round 4: the preallocating thread also populates the mapping the payload is copied through; without that)
    assert run("nomap.jf", "reads.fq", {"RFX_PREALLOC_MIN": "0", "RFX_PREALLOC_FRAC": "3.0", "RFX_NO_PREMAP": "1"}) == ref
    # k = 31 (the tumor/normal config): eager and deferred (shard passes inside the table) agree with the oracle
    ref31 = oracle.count(None, 31, 8 << 30, lower=2, reads=[x.tobytes() for x in seq]).payload()
    for extra in ({}, {"RFX_COUNT_DEFER": "1", "RFX_COUNT_PASSES": "3"}):
        p = subprocess.run([f"{BIN}/jellyfish", "count", "--disk", "-m", "31", "-L", "2", "-s", "8G", "-C", "-t", "8", "-o",
                            "k31.jf", "reads.fq"], cwd=d, env=dict(env, **extra), stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=300)
        assert p.returncode == 0, p.stderr
        assert _payload(f"{d}/k31.jf") == ref31
    assert run("defer.jf", "reads.fq", {"RFX_COUNT_DEFER": "1", "RFX_COUNT_PASSES": "3"}) == ref
    os.mkfifo(f"{d}/pipe.fq")
    feeder = threading.Thread(target=lambda: open(f"{d}/pipe.fq", "wb").write(open(f"{d}/reads.fq", "rb").read()))
    feeder.start()
    got = run("pipe.jf", "pipe.fq", {"RFX_COUNT_PASSES": "2"})
    feeder.join()
    assert got == ref
    # no final newline + blank lines between records
    text = open(f"{d}/reads.fq", "rb").read()
    recs = text.rstrip(b"\n").split(b"\n@")
    open(f"{d}/odd.fq", "wb").write(b"\n\n@".join(recs))
    assert run("odd.jf", "odd.fq") == ref
    # multi-line FASTQ: sequence and qualities wrapped at 60 columns
    lines = text.split(b"\n")
    wrapped = []
    for i in range(0, len(lines) - 1, 4):
        h, s, p, q = lines[i:i + 4]
        wrapped += [h] + [s[j:j + 60] for j in range(0, len(s), 60)] + [p] + [q[j:j + 60] for j in range(0, len(q), 60)]
    open(f"{d}/wrapped.fq", "wb").write(b"\n".join(wrapped) + b"\n")
    assert run("wrapped.jf", "wrapped.fq") == ref


def test_count_cli_reproduces_jellyfish_own_md5_kats(tmp_path):
    """jellyfish's own functional known-answer tests (tests/parallel_hashing.sh in jellyfish-2.2.5.tar.gz) against the
    DROP-IN executable: `count -m 15 -C -s 2M` (+ `-L2 -U3 --disk`) on the seeded 10 Mb sequence, md5 of `histo`."""
    from tests.test_oracle import mt_sequence
    d = str(tmp_path)
    seq10m, = mt_sequence(3141592653, [10_000_000])
    with open(f"{d}/seq10m.fa", "wb") as f:
        f.write(b">read0\n")
        for i in range(0, len(seq10m), 70):
            f.write(seq10m[i:i + 70] + b"\n")
    for extra, md5 in (([], "864c0b0826854bdc72a85d170549b64b"),
                       (["-L2", "-U3", "--disk"], "94625cd2d59e278f08421a673eb0926a")):
        r = sh([f"{BIN}/jellyfish", "count", "-t", "4", "-o", "m15.jf", "-s", "2M", "-C", "-m", "15"] + extra + ["seq10m.fa"], d,
               timeout=600)
        assert r.returncode == 0, r.stderr
        r = sh([f"{BIN}/jellyfish", "histo", "m15.jf"], d)
        assert r.returncode == 0, r.stderr
        assert hashlib.md5(r.stdout).hexdigest() == md5


def test_count_text_route_on_the_device_matches_the_host_route_and_the_golden(testrun, tmp_path):
    """Round 6, SURVEY section 2 K1: `jellyfish count` of a regular FASTQ file with the text parsed ON THE DEVICE
    (RFX_DEVICE_PARSE=1: host/rfx_ingest.hpp TextIngest -> rfx_text_*) writes the payload of the host route
    (RFX_HOST_PARSE=1) = the golden sha256 -- also from pieces of 3 KB (hundreds of appends per arena, every buffer reused
    many times), from a file without its final newline, and from a file with blank lines between records, which the
    device refuses and hands back to the host parser."""
    exp = testrun["expected"]
    d = str(tmp_path)
    fq = testrun["Child"][0] + testrun["Child"][1]
    want = exp["samples"]["Child"]["s100M"]["payload_sha256"]
    recs = fq.split(b"\n@")
    blank = b"\n@".join(recs[:40]) + b"\n\n@" + b"\n@".join(recs[40:])
    base = [f"{BIN}/jellyfish", "count", "--disk", "-m", "25", "-L", "2", "-s", "100M", "-t", "6", "-C"]
    for name, text in (("c.fq", fq), ("nonl.fq", fq.rstrip(b"\n")), ("blank.fq", blank)):
        open(f"{d}/{name}", "wb").write(text)
        for env in ({"RFX_DEVICE_PARSE": "1"}, {"RFX_HOST_PARSE": "1"}, {"RFX_DEVICE_PARSE": "1", "RFX_INGEST_PIECE": "3000"},
                    {"RFX_DEVICE_PARSE": "1", "RFX_TEXT_PREAD": "1"}):
            r = subprocess.run(base + ["-o", "o.Jhash", name], cwd=d, env=dict(os.environ, RFX_CLI_TRACE="1", **env),
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert r.returncode == 0, r.stderr
            assert (b"text route" in r.stderr) == ("RFX_DEVICE_PARSE" in env), r.stderr
            blob = open(f"{d}/o.Jhash", "rb").read()
            assert hashlib.sha256(blob[9 + int(blob[:9]):]).hexdigest() == want, (name, env)
