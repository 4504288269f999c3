"""CPU-only: the evidence chain of the bench line.  bench.py quotes the PMC traffic of profiles/rNN_pmc_<workload>.json only when
that file was measured on the kernel sources of THIS build (bench.kernel_source_fingerprint(), written into the JSON by
profiles/summarize_pmc.py); here: the fingerprint is what it says it is, and the committed profile is either current
(then the bench line will carry `traffic`) or visibly stale (skip with the reason -- the bench line will say "refused")."""
import glob
import hashlib
import json
import os

import pytest

import bench
from tests.conftest import ROOT


def test_fingerprint_covers_the_device_sources():
    fp = bench.kernel_source_fingerprint()
    assert len(fp) == 16 and int(fp, 16) >= 0 and fp == bench.kernel_source_fingerprint()
    d = os.path.join(ROOT, "rufus_amd", "csrc")
    names = sorted(n for n in os.listdir(d) if n.endswith((".hip", ".h")) or n == "Makefile")
    assert {"rfx_msp.hip", "rfx_api.hip", "rfx_devutil.h", "Makefile"} <= set(names)
    h = hashlib.sha256()
    for n in names:
        h.update(n.encode() + open(os.path.join(d, n), "rb").read())
    assert h.hexdigest()[:16] == fp


def test_committed_pmc_profile_names_its_build():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_wgs.json")))
    assert files, "no PMC profile of the W workload under profiles/"
    pmc = json.load(open(files[-1]))
    assert "_chain" in pmc and pmc["_chain"]["hbm_bytes_per_sample"] > 1e11
    if len(str(pmc.get("kernel_sources_sha16", ""))) != 16:
        pytest.skip(f"{os.path.basename(files[-1])} predates the fingerprint: bench.py will not quote its traffic")
    if pmc["kernel_sources_sha16"] != bench.kernel_source_fingerprint():
        pytest.skip(f"{os.path.basename(files[-1])} was taken on kernel sources {pmc['kernel_sources_sha16']}, this tree is "
                    f"{bench.kernel_source_fingerprint()}: bench.py will refuse to quote its traffic until it is re-taken")
    assert pmc.get("commit")
