"""GPU: the assembly chain against the REAL reference binaries under
oracle/_ref run with Threads = 1: OverlapSam -> ReplaceQwithDinFASTQD -> ConvertFASTqD.to.FASTQ ->
AnnotateOverlap, byte-identical files; plus the scoring kernel against a direct restatement."""
import os
import subprocess

import numpy as np
import pytest

import oracle
from rufus_amd import capi
from tests.conftest import ROOT
from tests.synth import make_trio

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "rufus_amd", "bin")
REF = os.path.join(ROOT, "oracle", "_ref")


@pytest.fixture
def _ref_built():
    from tests.conftest import require_ref
    require_ref("OverlapSam")   # (fails, not skips, on a GPU box without the reference binaries)


needs_ref = pytest.mark.usefixtures("_ref_built")


def align3_one(a: bytes, b: bytes, min_pct: float, min_ovl: int, strict3: bool, init: int):
    """Per-candidate body of Align3 (src/OverlapSam.cpp:47-229 / src/Overlap.cpp:176-340), float32 as the
    reference: returns (p1 score, p1 overlap, perfect, full score, full overlap)."""
    f = np.float32
    al, bl = len(a), len(b)
    asm = not (bl > al)
    window, longest = (bl, al) if asm else (al, bl)
    mm = int(f(window) - f(window) * f(min_pct))
    best, ovl, perfect = init, 0, False
    ac = bc = 0
    for i in range(longest - window + 1):
        score = f(0)
        for k in range(window):
            if a[k + ac] == b[k + bc] and b[k + bc] != ord("N"):
                score += f(1)
            if f(k) - score > mm:
                score = f(-1)
                break
        if asm:
            ac += 1
        else:
            bc += 1
        if window and f(score / f(window)) >= f(min_pct):
            if best < score:
                best, ovl = int(score), (-i if asm else i)
            if score == window:
                perfect = True
                break
    p1 = (best, ovl, perfect)
    if not perfect:
        for phase in (2, 3):
            for i in range(window - 1, min_ovl - 1, -1):
                score, k = f(0), 0
                for k in range(i + 1):
                    x, y = (a[al - i + k - 1], b[k]) if phase == 2 else (b[bl - i + k - 1], a[k])
                    if x == y and y != ord("N"):
                        score += f(1)
                    if f(k) - score > mm:
                        score = f(-1)
                        break
                else:
                    k = i + 1
                pct = f(score / f(k)) if k else f(0)
                ok = pct > f(min_pct) if (phase == 3 and strict3) else pct >= f(min_pct)
                if ok and best < score:
                    best, ovl = int(score), (i - al + 1 if phase == 2 else bl - i - 1)
                    if score == i:
                        break
    return p1[0], p1[1], int(p1[2]), best, ovl


def test_overlap_score_kernel_matches_restatement(ctx):
    rng = np.random.default_rng(17)
    base = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 600)].tobytes()

    def mutate(s, n_sub, n_n):
        s = bytearray(s)
        for p in rng.integers(0, len(s), n_sub):
            s[p] = b"ACGT"[int(rng.integers(0, 4))]
        for p in rng.integers(0, len(s), n_n):
            s[p] = ord("N")
        return bytes(s)

    a = base[100:250]
    cands = [base[100:250], base[130:280], base[60:210], mutate(base[120:270], 3, 2), mutate(base[90:240], 12, 0),
             base[150:200], base[50:400], b"moved", b"", mutate(base[245:395], 1, 1), base[300:450], a[:149],
             mutate(base[100:250], 2, 0), b"N" * 150, base[101:251]]
    for variant, strict3, init, pct, movl in ((capi.OVL_SAM, False, 0, 0.95, 20), (capi.OVL_CONTIG, True, -1, 0.98, 50),
                                              (capi.OVL_REGION, False, 0, 0.98, 50), (capi.OVL_CONTIG, True, -1, 0.9, 5)):
        got = capi.overlap_score(ctx, a, cands, pct, movl, variant)
        for j, b in enumerate(cands):
            want = align3_one(a, b, pct, movl, strict3, init)
            assert tuple(int(x) for x in got[j]) == want, (variant, j, got[j], want)


def fabricate_sam(seed=31, genome_len=30_000, n_pairs=4000, n_snv=3):
    """Position-sorted SAM (flags 99/147 + a few unmapped, duplicate-flagged, short and low-quality records)
    of the child read pairs that carry a mutant k-mer, as bwa + samtools sort would hand them over."""
    trio = make_trio(genome_len=genome_len, n_pairs=n_pairs, n_snv=n_snv, seed=seed)
    k = 25
    reads = {n: [r.tobytes() for m in (0, 1) for r in trio[n].s[m]] for n in ("child", "mother", "father")}
    recs = {n: oracle.count(None, k, 1 << 27, lower=2, reads=reads[n]) for n in reads}
    hl = oracle.hash_list(recs["child"], [recs["mother"], recs["father"]], 5, 1200)
    fs = oracle.FilterSet(hl.encode())
    c = trio["child"]
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    rows = []
    rng = np.random.default_rng(seed)
    for i in range(len(c)):
        s1, q1, s2, q2 = c.s[0][i].tobytes(), c.q[0][i].tobytes(), c.s[1][i].tobytes(), c.q[1][i].tobytes()
        if fs.scan(s1, q1, k, 15) < 1 and fs.scan(s2, q2, k, 15) < 1:
            continue
        name = f"c{i}"
        rows.append((int(c.pos[0][i]), name, 99, s1, q1))
        rows.append((int(c.pos[1][i]), name, 147, s2.translate(comp)[::-1], q2[::-1]))
        if rng.random() < 0.1:
            rows.append((int(c.pos[0][i]) + 1, name + "u", 77, s1, q1))                 # unmapped
        if rng.random() < 0.05:
            rows.append((int(c.pos[0][i]) + 2, name + "d", 1024 + 99, s1, q1))          # duplicate: rejected
        if rng.random() < 0.05:
            rows.append((int(c.pos[0][i]) + 3, name + "s", 0, s1[:40], q1[:40]))        # too short: rejected
        if rng.random() < 0.05:
            rows.append((int(c.pos[0][i]) + 4, name + "q", 16, s1, b"#" * 70 + q1[70:]))  # > 33 % low quality
    rows.sort(key=lambda r: r[0])
    sam = b"".join(b"\t".join([n.encode(), str(f).encode(), b"chr1", str(p + 1).encode(), b"60", b"150M", b"=",
                               b"1", b"0", s, q, b"NM:i:0"]) + b"\n" for p, n, f, s, q in rows)
    return sam, hl, len(rows)


@needs_ref
@pytest.mark.parametrize("mincov", ["1", "2"])
def test_overlapsam_and_tail_match_reference(tmp_path, mincov):
    sam, hl, n = fabricate_sam()
    assert n > 60
    d = str(tmp_path)
    open(f"{d}/in.sam", "wb").write(sam)
    open(f"{d}/hl", "w").write(hl)

    def run(exe, args, stdout=None):
        r = subprocess.run([exe] + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert r.returncode in (0,), (exe, r.stderr[-500:])
        if stdout:
            open(f"{d}/{stdout}", "wb").write(r.stdout)
        return r

    for tag, where in (("ours", BIN), ("ref", REF)):
        run(f"{where}/OverlapSam", ["in.sam", ".95", "20", mincov, f"{tag}.sam", "N", "1", "hl", "1"])
        run(f"{where}/ReplaceQwithDinFASTQD", [f"{tag}.sam.fastqd"], f"{tag}.overlap.fastqd")
        run(f"{where}/ConvertFASTqD.to.FASTQ", [f"{tag}.overlap.fastqd"], f"{tag}.overlap.fastq")
        run(f"{where}/AnnotateOverlap", ["hl", f"{tag}.overlap.fastq", f"{tag}.hash.fastq"], f"{tag}.hashcount.fastq")
    for f in ("sam.fastq", "sam.fastqd", "overlap.fastqd", "overlap.fastq", "hashcount.fastq", "hash.fastq"):
        a, b = open(f"{d}/ours.{f}", "rb").read(), open(f"{d}/ref.{f}", "rb").read()
        assert a == b, f
        assert len(a) > 500, f
    # the contigs carry mutant k-mer coverage: some quality characters above '!'
    q = open(f"{d}/ours.hashcount.fastq", "rb").read().split(b"\n")[3::4]
    assert any(max(x) > 33 for x in q if x)


@needs_ref
def test_full_assembly_chain_matches_reference(tmp_path):
    """scripts/Overlap.shorter.sh:127-194 with the reference's own arguments, every stage fed by the
    previous stage of its own side (ours / reference), Threads = 1."""
    sam, hl, n = fabricate_sam(seed=77)
    d = str(tmp_path)
    open(f"{d}/in.sam", "wb").write(sam)
    open(f"{d}/hl", "w").write(hl)

    def run(exe, args, stdout=None):
        r = subprocess.run([exe] + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, (exe, r.stderr[-500:])
        if stdout:
            open(f"{d}/{stdout}", "wb").write(r.stdout)

    sizes = {}
    for t, w in (("ours", BIN), ("ref", REF)):
        run(f"{w}/OverlapSam", ["in.sam", ".95", "20", "1", f"{t}.sam", "NS", "1", "hl", "1"])
        run(f"{w}/Overlap", [f"{t}.sam.fastqd", ".98", "100", "1", "FP", "20", "1", f"{t}.1", "0", "1"])
        run(f"{w}/Overlap", [f"{t}.1.fastqd", ".98", "75", "2", "FP", "20", "1", f"{t}.2", "1", "1"])
        run(f"{w}/Overlap", [f"{t}.2.fastqd", ".98", "50", "2", "NS", "20", "1", f"{t}.3", "1", "1"])
        run(f"{w}/OverlapRegion", [f"{t}.3.fastqd", ".98", "50", "2", f"{t}.4", "NS", "1", "1"])
        run(f"{w}/ReplaceQwithDinFASTQD", [f"{t}.4.fastqd"], f"{t}.overlap.fastqd")
        run(f"{w}/ConvertFASTqD.to.FASTQ", [f"{t}.overlap.fastqd"], f"{t}.overlap.fastq")
        run(f"{w}/AnnotateOverlap", ["hl", f"{t}.overlap.fastq", f"{t}.asm.hash.fastq"], f"{t}.hashcount.fastq")
    stages = ["sam.fastqd", "1.fastqd", "1.fastq", "1.fastqgood.fastq", "1.fastqbad.fastq", "2.fastqd", "3.fastqd",
              "4.fastqd", "4.fastq", "overlap.fastqd", "overlap.fastq", "hashcount.fastq", "asm.hash.fastq"]
    for f in stages:
        a, b = open(f"{d}/ours.{f}", "rb").read(), open(f"{d}/ref.{f}", "rb").read()
        sizes[f] = a.count(b"\n")
        assert a == b, (f, sizes)
    # the chain really assembles: node counts shrink stage by stage and something comes out
    nodes = [sizes[f] // 6 for f in ("sam.fastqd", "1.fastqd", "2.fastqd", "3.fastqd", "4.fastqd")]
    assert nodes[0] > nodes[-1] >= 1, nodes


def _chain(d, tag, where, sam, hl, final_cov="2", timings=None):
    """scripts/Overlap.shorter.sh:127-194 for one side (ours / reference), Threads = 1."""
    import time

    def run(exe, args, stdout=None):
        t0 = time.perf_counter()
        r = subprocess.run([f"{where}/{exe}"] + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1800)
        assert r.returncode == 0, (exe, r.stderr[-500:])
        if stdout:
            open(f"{d}/{stdout}", "wb").write(r.stdout)
        if timings is not None:
            timings[exe + ":" + (args[-3] if exe == "Overlap" else "")] = round(time.perf_counter() - t0, 2)

    t = tag
    run("OverlapSam", [sam, ".95", "20", "1", f"{t}.sam", "NS", "1", hl, "1"])
    run("Overlap", [f"{t}.sam.fastqd", ".98", "100", "1", "FP", "20", "1", f"{t}.1", "0", "1"])
    run("Overlap", [f"{t}.1.fastqd", ".98", "75", "2", "FP", "20", "1", f"{t}.2", "1", "1"])
    run("Overlap", [f"{t}.2.fastqd", ".98", "50", "2", "NS", "20", "1", f"{t}.3", "1", "1"])
    run("OverlapRegion", [f"{t}.3.fastqd", ".98", "50", final_cov, f"{t}.4", "NS", "1", "1"])
    run("ReplaceQwithDinFASTQD", [f"{t}.4.fastqd"], f"{t}.overlap.fastqd")
    run("ConvertFASTqD.to.FASTQ", [f"{t}.overlap.fastqd"], f"{t}.overlap.fastq")
    run("AnnotateOverlap", [hl, f"{t}.overlap.fastq", f"{t}.asm.hash.fastq"], f"{t}.hashcount.fastq")


_STAGES = ["sam.fastqd", "1.fastqd", "2.fastqd", "3.fastqd", "4.fastqd", "4.fastq", "overlap.fastqd", "overlap.fastq",
           "hashcount.fastq", "asm.hash.fastq"]


@needs_ref
def test_testrun_pulled_pairs_assemble_like_the_reference(testrun, tmp_path):
    """SURVEY 8(c) golden 3: the 26 read pairs RUFUS.Filter pulls from the reference's own test trio, as a
    fabricated position-sorted SAM (flags 99/147; no bwa here), through the whole assembly chain with the
    reference's arguments (FinalCoverage 5): byte-identical to the reference binaries at every stage, 7 -> 7 -> 3 ->
    3 -> 2 nodes (the survey's probe counted 8 after OverlapSam: its read order is not recorded) and the 2-record
    hashcount.fastq the VCF step would start from."""
    import gzip
    d = str(tmp_path)
    names = testrun["expected"]["filter_paired_names"]

    def recs(blob):
        t = blob.split(b"\n")
        return {t[i][1:].split()[0].decode(): (t[i + 1], t[i + 3]) for i in range(0, len(t) - 1, 4)}

    m1, m2 = recs(testrun["Child"][0]), recs(testrun["Child"][1])
    rows = []
    for i, n in enumerate(names):
        key = n.lstrip("@")
        for flag, (s, q), off in ((b"99", m1[key], 0), (b"147", m2[key], 5)):
            rows.append(b"\t".join([key.encode(), flag, b"5", str(1000 + 10 * i + off).encode(), b"60", b"151M", b"=", b"1",
                                    b"0", s, q, b"NM:i:0"]))
    open(f"{d}/in.sam", "wb").write(b"\n".join(rows) + b"\n")
    open(f"{d}/hl", "w").write(testrun["hashlist"])
    for tag, where in (("ours", BIN), ("ref", REF)):
        _chain(d, tag, where, "in.sam", "hl", final_cov="5")
    nodes = []
    for f in _STAGES:
        a, b = open(f"{d}/ours.{f}", "rb").read(), open(f"{d}/ref.{f}", "rb").read()
        assert a == b, f
        if f.endswith(".fastqd") and f[0] in "s1234":
            nodes.append(a.count(b"\n") // 6)
    assert nodes == [7, 7, 3, 3, 2], nodes
    assert open(f"{d}/ours.hashcount.fastq", "rb").read().count(b"\n") == 8


@needs_ref
def test_assembly_chain_on_ten_thousand_reads(tmp_path):
    """The chain on > 10^4 pulled reads (500 SNVs): the device-resident read pool (uploaded once, one patched entry
    per merge, both strands per launch) gives the reference's files byte for byte at Threads = 1; wall times of
    both sides are printed (pytest -s) -- the reference's own O(N L^2) scans against one launch per greedy step."""
    sam, hl, n = fabricate_sam(seed=5, genome_len=600_000, n_pairs=60_000, n_snv=500)
    assert n > 10_000
    d = str(tmp_path)
    open(f"{d}/in.sam", "wb").write(sam)
    open(f"{d}/hl", "w").write(hl)
    times = {}
    for tag, where in (("ours", BIN), ("ref", REF)):
        times[tag] = {}
        _chain(d, tag, where, "in.sam", "hl", timings=times[tag])
    print("assembly chain wall times (s):", n, "SAM records;", times)
    for f in _STAGES:
        a, b = open(f"{d}/ours.{f}", "rb").read(), open(f"{d}/ref.{f}", "rb").read()
        assert a == b, f
    assert open(f"{d}/ours.4.fastqd", "rb").read().count(b"\n") // 6 >= 50
