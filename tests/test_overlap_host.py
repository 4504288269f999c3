"""CPU-only: the greedy assemblers of the drop-in tool set through tests/host/overlap_harness.cpp --
the tools' own main() (SAM intake, collapse passes, greedy merge loops, output) over a host stand-in for the scoring
kernel -- against the reference binaries under oracle/_ref, stage by stage, byte for byte: scripts/Overlap.shorter.sh:127-194
with the reference's arguments, Threads = 1.  Plain and under ASan/UBSan.  tests/test_overlap_gpu.py runs the same chain
through the real executables and holds the kernel to the same Align3 restatement."""
import os
import subprocess

import pytest

from tests.conftest import ROOT
from tests.test_cli_host import BIN, REF

needs_ref = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "OverlapSam")), reason="oracle/_ref not built")
SRC = [os.path.join(ROOT, "tests", "host", "overlap_harness.cpp"), os.path.join(ROOT, "rufus_amd", "csrc", "rfx_host.cpp")]
NAMES = {0: "OverlapSam", 1: "Overlap", 2: "OverlapRegion", 3: "AnnotateOverlap"}


def _build(d, flags):
    jobs = [subprocess.Popen(["g++", "-std=c++17", "-pthread", "-ffp-contract=off", f"-DOVL_WHICH={w}", "-o", str(d / name)] + flags + SRC)
            for w, name in NAMES.items()]           # side by side: the sanitizer builds take ~6 s each
    assert all(j.wait() == 0 for j in jobs)
    for name in ("ReplaceQwithDinFASTQD", "ConvertFASTqD.to.FASTQ"):      # device-free: the product's own binaries
        os.symlink(os.path.join(BIN, name), str(d / name))
    return str(d)


@pytest.fixture(scope="module", params=["plain", "address,undefined"])
def tools(request, tmp_path_factory):
    d = tmp_path_factory.mktemp("ovh")
    if request.param == "plain":
        return _build(d, ["-O2"])
    probe = d / "probe.cpp"
    probe.write_text("int main() { return 0; }\n")
    flags = ["-O1", "-g", "-fsanitize=" + request.param, "-fno-sanitize-recover=all"]
    if subprocess.run(["g++"] + flags + ["-o", str(d / "probe"), str(probe)], stderr=subprocess.DEVNULL).returncode != 0:
        pytest.skip(f"no usable -fsanitize={request.param} here")
    return _build(d, flags)


def _chain(d, t, w, mincov="1"):
    def run(exe, args, stdout=None):
        r = subprocess.run([f"{w}/{exe}"] + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200,
                           env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
        assert r.returncode == 0 and b"Sanitizer" not in r.stderr and b"runtime error" not in r.stderr, (exe, r.stderr[-1500:])
        if stdout:
            open(f"{d}/{stdout}", "wb").write(r.stdout)

    run("OverlapSam", ["in.sam", ".95", "20", mincov, f"{t}.sam", "NS", "1", "hl", "1"])
    run("Overlap", [f"{t}.sam.fastqd", ".98", "100", "1", "FP", "20", "1", f"{t}.1", "0", "1"])
    run("Overlap", [f"{t}.1.fastqd", ".98", "75", "2", "FP", "20", "1", f"{t}.2", "1", "1"])
    run("Overlap", [f"{t}.2.fastqd", ".98", "50", "2", "NS", "20", "1", f"{t}.3", "1", "1"])
    run("OverlapRegion", [f"{t}.3.fastqd", ".98", "50", "2", f"{t}.4", "NS", "1", "1"])
    run("ReplaceQwithDinFASTQD", [f"{t}.4.fastqd"], f"{t}.overlap.fastqd")
    run("ConvertFASTqD.to.FASTQ", [f"{t}.overlap.fastqd"], f"{t}.overlap.fastq")
    run("AnnotateOverlap", ["hl", f"{t}.overlap.fastq", f"{t}.asm.hash.fastq"], f"{t}.hashcount.fastq")


@needs_ref
def test_assembly_chain_host_side_matches_reference(tools, tmp_path):
    from tests.test_overlap_gpu import fabricate_sam
    sam, hl, n = fabricate_sam(seed=77)
    d = str(tmp_path)
    open(f"{d}/in.sam", "wb").write(sam)
    open(f"{d}/hl", "w").write(hl)
    _chain(d, "ours", tools)
    _chain(d, "ref", REF)
    sizes = {}
    for f in ["sam.fastq", "sam.fastqd", "1.fastqd", "1.fastq", "1.fastqgood.fastq", "1.fastqbad.fastq", "2.fastqd", "3.fastqd",
              "4.fastqd", "4.fastq", "overlap.fastqd", "overlap.fastq", "hashcount.fastq", "asm.hash.fastq"]:
        a, b = open(f"{d}/ours.{f}", "rb").read(), open(f"{d}/ref.{f}", "rb").read()
        sizes[f] = a.count(b"\n")
        assert a == b, (f, sizes)
    nodes = [sizes[f] // 6 for f in ("sam.fastqd", "1.fastqd", "2.fastqd", "3.fastqd", "4.fastqd")]
    assert nodes[0] > nodes[-1] >= 1, nodes
    # the contigs carry mutant k-mer coverage: some quality characters above '!'
    q = open(f"{d}/ours.hashcount.fastq", "rb").read().split(b"\n")[3::4]
    assert any(max(x) > 33 for x in q if x)
