"""CPU-only: the host-only drop-in executables (feeders P1/P2, RUFUS.Build) against the REAL reference
binaries under oracle/_ref (built from /root/reference/src by `make -C oracle ref`): byte-identical
outputs on the same stdin / files."""
import os
import subprocess

import numpy as np
import pytest

from tests.conftest import ROOT
from tests.synth import make_trio

BIN = os.path.join(ROOT, "rufus_amd", "bin")
REF = os.path.join(ROOT, "oracle", "_ref")

# The host-only tools are part of the drop-in boundary on the GPU box as well (oracle/_ref ships with the snapshot): every
# test of this file runs once in the CPU session (-m "not gpu") and once more in the GPU session (-m gpu).
@pytest.fixture(autouse=True, params=["cpu-session", pytest.param("gpu-session", marks=pytest.mark.gpu)])
def _both_sessions(request):
    return request.param


needs_ref = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "PassThroughSamCheck")),
                               reason="oracle/_ref not built (needs /root/reference)")


def make_sam(n_pairs=400, seed=3) -> bytes:
    """Coordinate-sorted-looking SAM of proper pairs (flags 99/147 and 83/163), a few unpaired
    records, several chromosomes, N bases, and the trailing TAB that scripts/FastqToSam.pl writes."""
    trio = make_trio(genome_len=20_000, n_pairs=n_pairs, n_snv=2, seed=seed)
    s = trio["child"]
    rng = np.random.default_rng(seed)
    recs = []
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    for i in range(n_pairs):
        pos = int(rng.integers(1, 15000))
        chrom = f"chr{1 + pos // 4000}"
        fwd_first = rng.random() < 0.5
        m1, q1 = s.s[0][i].tobytes(), s.q[0][i].tobytes()
        m2, q2 = s.s[1][i].tobytes(), s.q[1][i].tobytes()     # synth stores mate 2 as sequenced (reverse strand)
        # SAM shows the reverse-strand mate reverse-complemented
        rc2, rq2 = m2.translate(comp)[::-1], q2[::-1]
        name = f"r{i}".encode()
        if fwd_first:
            recs.append((chrom, pos, name, 99, m1, q1))
            recs.append((chrom, pos + 200, name, 147, rc2, rq2))
        else:
            recs.append((chrom, pos, name, 163, rc2 if False else m1, q1))
            recs.append((chrom, pos + 200, name, 83, rc2, rq2))
        if i % 50 == 0:                                         # a read whose mate never arrives
            recs.append((chrom, pos + 5, f"orphan{i}".encode(), 73, m1, q1))
    recs.sort(key=lambda r: (r[0], r[1]))
    out = []
    for chrom, pos, name, flag, seq, qual in recs:
        out.append(b"\t".join([name, str(flag).encode(), chrom.encode(), str(pos).encode(), b"60", b"150M", b"=",
                               str(pos + 200).encode(), b"350", seq, qual, b"NM:i:0"]) + b"\n")
    # the FASTQ route (scripts/FastqToSam.pl): 11 fields and a trailing TAB
    out.append(b"fq1\t0\t*\t0\t*\t*\t*\t0\t0\tACGTNACGT\tJJJJ#JJJJ\t\n")
    return b"".join(out)


def run(exe, args, stdin, cwd):
    return subprocess.run([exe] + args, input=stdin, stdout=subprocess.PIPE, cwd=cwd, check=True, timeout=120).stdout


@needs_ref
def test_pass_through_sam_check(tmp_path):
    sam = make_sam() + make_sam(20_000, seed=5)          # 13 MB: several pieces for the helper threads
    b = run(f"{REF}/PassThroughSamCheck", ["ref.chr"], sam, tmp_path)
    for helpers in ("0", "3"):
        a = subprocess.run([f"{BIN}/PassThroughSamCheck", "ours.chr"], input=sam, stdout=subprocess.PIPE, cwd=tmp_path,
                           check=True, timeout=120, env=dict(os.environ, RFX_PTS_THREADS=helpers)).stdout
        assert a == b and a.count(b"\n") % 4 == 0 and len(a) > 100000
        assert (tmp_path / "ours.chr").read_bytes() == (tmp_path / "ref.chr").read_bytes()
        assert (tmp_path / "ours.chr").read_text().startswith("notachr\n")


@needs_ref
def test_pass_through_stranded_pairs(tmp_path):
    sam = make_sam()
    run(f"{BIN}/PassThroughSamCheck.stranded", ["ours.chr", "ours"], sam, tmp_path)
    run(f"{REF}/PassThroughSamCheck.stranded", ["ref.chr", "ref"], sam, tmp_path)
    for m in (1, 2):
        a, b = (tmp_path / f"ours.mate{m}.fastq").read_bytes(), (tmp_path / f"ref.mate{m}.fastq").read_bytes()
        assert a == b and a.count(b"\n") == 4 * 400           # orphans and the unpaired FASTQ-route read are dropped
    assert (tmp_path / "ours.chr").read_bytes() == (tmp_path / "ref.chr").read_bytes()


@needs_ref
def test_pass_through_stranded_many_waiting_reads(tmp_path):
    """Records in random order (tens of thousands of reads wait for their mate: the table grows, fills with
    deletions, is rebuilt), names seen three and four times, lower-case and IUPAC bases in reverse-strand reads
    (the reference drops them), flags with other bits set."""
    rng = np.random.default_rng(9)
    n = 30_000
    alphabet = np.frombuffer(b"ACGT" * 8 + b"Nacgtn" + b"RY", np.uint8)
    recs = []
    for i in range(n):
        times = 2 if i % 97 else (3 if i % 2 else 4)
        for t in range(times):
            L = int(rng.integers(30, 151))
            seq = bytes(rng.choice(alphabet, L))
            qual = bytes(rng.integers(35, 75, L, dtype=np.uint8))
            flag = int(rng.choice([99, 147, 83, 163, 16, 0, 1040, 65]))
            recs.append(b"\t".join([b"q%d" % i, b"%d" % flag, b"chr%d" % (1 + i % 5), b"%d" % (i + t), b"60", b"*", b"=",
                                    b"1", b"0", seq, qual, b"XS:i:%d" % t]) + b"\n")
    order = rng.permutation(len(recs))
    sam = b"".join(recs[j] for j in order)
    run(f"{REF}/PassThroughSamCheck.stranded", ["ref.chr", "ref"], sam, tmp_path)
    for helpers in ("0", "1", "5"):                 # the single-threaded loop, and pieces parsed by helper threads
        subprocess.run([f"{BIN}/PassThroughSamCheck.stranded", "ours.chr", "ours"], input=sam, cwd=tmp_path, check=True,
                       timeout=120, env=dict(os.environ, RFX_PTS_THREADS=helpers))
        for m in (1, 2):
            a, b = (tmp_path / f"ours.mate{m}.fastq").read_bytes(), (tmp_path / f"ref.mate{m}.fastq").read_bytes()
            assert a == b and a.count(b"\n") > 4 * n, helpers
        assert (tmp_path / "ours.chr").read_bytes() == (tmp_path / "ref.chr").read_bytes()


@needs_ref
def test_stranded_feeder_keeps_the_reference_lock_step_reader_moving(tmp_path):
    """runRufus.sh:964-967 with the reference's own RUFUS.Filter as the reader (four lines from one pipe, four from
    the other): the feeder writes both pipes in chunks that hold the same pairs, never one pipe ahead by more than a
    chunk -- the run ends, with the pairs the file route pulls."""
    import threading
    from rufus_amd import capi
    d = str(tmp_path)
    subprocess.run([f"{BIN}/rfx_synth_fastq", "2000000", "0", "20", "7", "0", "40000", "in.sam"], cwd=d, check=True,
                   env=dict(os.environ, RFX_SYNTH_SAM="1"))
    sy = capi.Synth.sample(2_000_000, 0, n_snv=20, seed=7)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    with open(f"{d}/hl", "w") as f:
        for p_, _, alt in sy.snvs():
            c = bytearray(sy.genome(p_ - 24, 49))
            c[24:25] = alt
            for i in range(25):
                km = bytes(c[i:i + 25])
                f.write(min(km, km[::-1].translate(comp)).decode() + " 12\n")
    os.mkfifo(f"{d}/s.mate1.fastq")
    os.mkfifo(f"{d}/s.mate2.fastq")
    feeder = subprocess.Popen([f"{BIN}/PassThroughSamCheck.stranded", "f.chr", "s"], cwd=d, stdin=open(f"{d}/in.sam", "rb"))
    r = subprocess.run([f"{REF}/RUFUS.Filter", "hl", "s.mate1.fastq", "s.mate2.fastq", "piped", "25", "15", "1", "2"], cwd=d,
                       stdout=subprocess.DEVNULL, timeout=120)
    assert r.returncode == 0 and feeder.wait(30) == 0
    subprocess.run([f"{BIN}/PassThroughSamCheck.stranded", "g.chr", "t"], cwd=d, stdin=open(f"{d}/in.sam", "rb"), check=True)
    subprocess.run([f"{REF}/RUFUS.Filter", "hl", "t.mate1.fastq", "t.mate2.fastq", "files", "25", "15", "1", "1"], cwd=d,
                   stdout=subprocess.DEVNULL, check=True, timeout=120)

    def records(path):
        lines = open(path, "rb").read().split(b"\n")
        return sorted(b"\n".join(lines[i:i + 4]) for i in range(0, len(lines) - 1, 4))

    for m in (1, 2):
        a, b = records(f"{d}/piped.Mutations.Mate{m}.fastq"), records(f"{d}/files.Mutations.Mate{m}.fastq")
        assert a == b and len(a) > 20


@needs_ref
def test_pass_through_stranded_single_end(tmp_path):
    sam = make_sam() + make_sam(20_000, seed=6)
    b = run(f"{REF}/PassThroughSamCheck.stranded.se", ["ref.chr"], sam, tmp_path)
    for helpers in ("0", "3"):
        a = subprocess.run([f"{BIN}/PassThroughSamCheck.stranded.se", "ours.chr"], input=sam, stdout=subprocess.PIPE,
                           cwd=tmp_path, check=True, timeout=120, env=dict(os.environ, RFX_PTS_THREADS=helpers)).stdout
        assert a == b
        assert (tmp_path / "ours.chr").read_bytes() == (tmp_path / "ref.chr").read_bytes()


@needs_ref
def test_rufus_build(tmp_path):
    import oracle
    trio = make_trio(genome_len=8000, n_pairs=700, n_snv=3, seed=21)
    tabs = {}
    for name in ("child", "mother", "father"):
        reads = [r.tobytes() for m in (0, 1) for r in trio[name].s[m]]
        rec = oracle.count(None, 21, 1 << 20, lower=2, reads=reads)
        rows = sorted(zip((oracle.jf_decode(int(k), 21) for k in rec.keys), rec.counts.tolist()))
        (tmp_path / f"{name}.tab").write_text("".join(f"{k}\t{c}\n" for k, c in rows))
        tabs[name] = rows
    for mS, mC, mx in (("5", "0", "1200"), ("2", "3", "40")):
        args = ["-c", "mother.tab", "-c", "father.tab", "-s", "child.tab", "-hs", "21", "-mS", mS, "-mC", mC, "-max", mx]
        run(f"{BIN}/RUFUS.Build", args + ["-o", "ours.out"], b"", tmp_path)
        run(f"{REF}/RUFUS.Build", args + ["-o", "ref.out"], b"", tmp_path)
        a, b = (tmp_path / "ours.out").read_bytes(), (tmp_path / "ref.out").read_bytes()
        assert a == b and len(a) > 0
