"""CPU-only: the .Jhash header of the drop-in tools (jf/include/jellyfish/generic_file_header.hpp:96-121,
file_header.hpp:33-110) -- what rfx_jhash_header writes, rfx_cli.hpp's read_jhash reads back, for the k / table sizes /
counter lengths the tools use; corrupt headers are refused (never a hang, a division by zero or a sanitizer report:
the harness is built with -fsanitize=address,undefined when the compiler has it)."""
import os
import re
import subprocess

import pytest

from tests.conftest import ROOT

SRC = [os.path.join(ROOT, "tests", "host", "jhash_header_harness.cpp"), os.path.join(ROOT, "rufus_amd", "csrc", "rfx_host.cpp")]


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("jh") / "jhash_header_harness")
    base = ["g++", "-O1", "-g", "-std=c++17", "-pthread", "-o", out] + SRC
    if subprocess.run(base + ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"], stderr=subprocess.DEVNULL).returncode != 0:
        subprocess.check_call(base)
    return out


def _run(harness, *args):
    r = subprocess.run([harness, *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=20,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0 and b"Sanitizer" not in r.stderr and b"runtime error" not in r.stderr, r.stderr[-2000:]
    return r.stdout.decode().split()


@pytest.mark.parametrize("k,lsize,canonical,clen,n", [(25, 33, 1, 4, 10), (31, 33, 1, 4, 0), (5, 8, 0, 1, 1000), (32, 27, 1, 2, 3),
                                                     (12, 24, 0, 8, 7)])
def test_header_round_trip(harness, tmp_path, k, lsize, canonical, clen, n):
    f = str(tmp_path / "db.jf")
    hl, c0, c1 = _run(harness, "write", f, str(k), str(lsize), str(canonical), str(clen), str(n))
    got = _run(harness, "read", f)
    rl = (2 * k + 7) // 8 + clen
    # (jellyfish stores "size" = 2^lsize, the reader takes the logarithm)
    assert got == [str(k), str(lsize), str(clen), str(canonical), "binary/sorted", str(2 * k), hl, str(int(hl) + n * rl), c0, c1]
    assert int(hl) % 8 == 0                               # 9 digits + JSON + NUL padding to 8 bytes
    text = open(f, "rb").read(int(hl))
    assert re.fullmatch(rb"\d{9}\{.*\}\x00*", text, re.S) and int(text[:9]) == int(hl) - 9


def test_corrupt_headers_are_refused(harness, tmp_path):
    f = str(tmp_path / "db.jf")
    hl = int(_run(harness, "write", f, "25", "33", "1", "4", "5")[0])
    good = open(f, "rb").read()

    def variant(name, data):
        p = str(tmp_path / name)
        open(p, "wb").write(data)
        return _run(harness, "read", p)

    assert variant("same", good)[0] == "25"
    js = good[9:hl]
    for name, data in [
            ("empty", b""), ("short", good[:5]), ("nodigits", b"abcdefghi" + js), ("cut", good[:hl // 2]),
            ("len0", b"000000000"), ("len1", b"000000001{"), ("notjson", b"%09d" % 16 + b"x" * 16),
            ("hugelen", b"999999999{\"key_len\":50}"),
            ("size64", good[:9] + re.sub(rb'"size":\d+', b'"size":18446744073709551615', js)),
            ("keylen0", good[:9] + re.sub(rb'"key_len":\d+', b'"key_len":0', js).replace(b'"columns":[', b'"columns":[]', 1)),
            ("clen0", good[:9] + re.sub(rb'"counter_len":\d+', b'"counter_len":0', js)),
            ("clen99", good[:9] + re.sub(rb'"counter_len":\d+', b'"counter_len":99', js)),
            ("nocols", good[:9] + js.replace(b'"columns"', b'"colums"')),
            ("fewcols", good[:9] + re.sub(rb'"columns":\[\d+,', b'"columns":[', js))]:
        if len(data) > 9 and data[:9].isdigit() and name in ("size64", "keylen0", "clen0", "clen99", "nocols", "fewcols"):
            body = data[9:]
            data = b"%09d" % len(body) + body           # keep the length field honest: the JSON is what is wrong
        got = variant(name, data)
        if name == "size64":                            # refused or read with a capped table size -- not a hang
            assert got == ["bad"] or got[0] == "25"
        else:
            assert got == ["bad"], (name, got)
