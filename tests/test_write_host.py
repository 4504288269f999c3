"""CPU-only: the .Jhash writer of the drop-in `jellyfish count` (rfx_cli.hpp write_jhash) through
tests/host/write_harness.cpp -- its ring of fetch buffers and writer threads, the per-slice fetch threads of a
several-device run (RUFUS_GPUS), the mapped file and the pipe route -- over stand-ins for the record sets, with 1000-record
fetches so that the rings go round a hundred times.  Built with -fsanitize=thread and with -fsanitize=address,undefined
(plain when the compiler has neither): a report fails the test."""
import os
import re
import subprocess

import numpy as np
import pytest

from tests.conftest import ROOT

SRC = [os.path.join(ROOT, "tests", "host", "write_harness.cpp"), os.path.join(ROOT, "rufus_amd", "csrc", "rfx_host.cpp")]
K = 25


@pytest.fixture(scope="module", params=["thread", "address,undefined"])
def harness(request, tmp_path_factory):
    out = str(tmp_path_factory.mktemp("wh") / "write_harness")
    base = ["g++", "-O1", "-g", "-std=c++17", "-pthread", "-DRFX_WRITE_STEP=1000", "-o", out] + SRC
    if subprocess.run(base + ["-fsanitize=" + request.param], stderr=subprocess.DEVNULL).returncode != 0:
        subprocess.check_call(base)
    return out


def _payload(n, clen):
    rl = (2 * K + 7) // 8 + clen
    g = np.arange(n, dtype=np.uint64)[:, None]
    j = np.arange(rl, dtype=np.uint64)[None, :]
    return (((g * np.uint64(2654435761) + j * np.uint64(40503)) >> np.uint64(7)) & np.uint64(255)).astype(np.uint8).tobytes()


def _no_time(blob):
    """The header carries the time of the run ("time":"Wed Sep 30 03:41:00 2026"): two runs may straddle a second."""
    return re.sub(rb'"time":"[^"]*"', b'"time":""', blob[:9 + int(blob[:9])]) + blob[9 + int(blob[:9]):]


def _run(cmd, **kw):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0", TSAN_OPTIONS="exitcode=66"), **kw)
    assert r.returncode == 0 and b"Sanitizer" not in r.stderr and b"runtime error" not in r.stderr, r.stderr[-3000:]
    return r.stdout


@pytest.mark.parametrize("clen,slices", [(4, [100_000]), (4, [30_000, 0, 50_000, 20_000]), (1, [12_345]), (8, [999, 1000, 1001]),
                                         (4, [0]), (2, [1])])
def test_writer_puts_every_record_where_it_belongs(harness, tmp_path, clen, slices):
    f = str(tmp_path / "out.jf")
    _run([harness, f, str(clen)] + [str(n) for n in slices])
    got = open(f, "rb").read()
    hl = 9 + int(got[:9])
    assert hl % 8 == 0 and got[9:10] == b"{"
    assert got[hl:] == _payload(sum(slices), clen)
    # the same bytes through a pipe (no offsets: one writer, in order) ...
    piped = _run(f"{harness} /dev/stdout {clen} {' '.join(map(str, slices))} | cat", shell=True)
    assert _no_time(piped) == _no_time(got)
    # ... and over a file that was longer before (the tools reuse output names: no stale tail may stay)
    open(f, "wb").write(b"x" * (len(got) + 4096))
    _run([harness, f, str(clen)] + [str(n) for n in slices])
    assert _no_time(open(f, "rb").read()) == _no_time(got)
