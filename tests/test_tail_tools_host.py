"""CPU-only: the two tail tools that never touch the device (ReplaceQwithDinFASTQD,
ConvertFASTqD.to.FASTQ; src/ReplaceQwithDinFASTQD.cpp, src/ConvertFASTqD.to.FASTQ.cpp) against the reference binaries on
the .fastqd the REFERENCE OverlapSam writes for a fabricated SAM (tests/test_overlap_gpu.py drives the whole chain with
the drop-in OverlapSam on the GPU); also under ASan/UBSan when the compiler has them."""
import os
import subprocess

import pytest

from tests.conftest import ROOT
from tests.test_cli_host import BIN, REF

needs_ref = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "OverlapSam")), reason="oracle/_ref not built")
SRC = os.path.join(ROOT, "rufus_amd", "csrc", "host", "overlap_tail_main.cpp")


def _run(exe, args, cwd, stdout):
    r = subprocess.run([exe] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0 and b"Sanitizer" not in r.stderr and b"runtime error" not in r.stderr, (exe, r.stderr[-1500:])
    open(os.path.join(cwd, stdout), "wb").write(r.stdout)


@needs_ref
@pytest.mark.parametrize("mincov", ["1", "2"])
def test_tail_tools_match_reference(tmp_path, mincov):
    from tests.test_overlap_gpu import fabricate_sam
    sam, hl, n = fabricate_sam()
    d = str(tmp_path)
    open(f"{d}/in.sam", "wb").write(sam)
    open(f"{d}/hl", "w").write(hl)
    r = subprocess.run([f"{REF}/OverlapSam", "in.sam", ".95", "20", mincov, "ref.sam", "N", "1", "hl", "1"], cwd=d,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr[-500:]
    san = {}
    for mode, name in ((0, "ReplaceQwithDinFASTQD"), (1, "ConvertFASTqD.to.FASTQ")):   # the sanitizer builds, when they can be had
        out = f"{d}/{name}.san"
        if subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=address,undefined", f"-DTAIL_MODE={mode}",
                           "-o", out, SRC], stderr=subprocess.DEVNULL).returncode == 0:
            san[name] = out
    for tag, where in (("ours", BIN), ("ref", REF), ("san", None)):
        exe = (lambda nm: san.get(nm)) if where is None else (lambda nm: f"{where}/{nm}")
        if exe("ReplaceQwithDinFASTQD") is None or exe("ConvertFASTqD.to.FASTQ") is None:
            continue
        _run(exe("ReplaceQwithDinFASTQD"), ["ref.sam.fastqd"], d, f"{tag}.overlap.fastqd")
        _run(exe("ConvertFASTqD.to.FASTQ"), [f"{tag}.overlap.fastqd"], d, f"{tag}.overlap.fastq")
    for f in ("overlap.fastqd", "overlap.fastq"):
        want = open(f"{d}/ref.{f}", "rb").read()
        assert len(want) > 500 and open(f"{d}/ours.{f}", "rb").read() == want, f
        if san:
            assert open(f"{d}/san.{f}", "rb").read() == want, f
