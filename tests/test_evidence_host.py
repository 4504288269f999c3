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


def _kernels_in_sources():
    import re
    d = os.path.join(ROOT, "rufus_amd", "csrc")
    names = set()
    for n in os.listdir(d):
        if n.endswith(".hip"):
            src = open(os.path.join(d, n)).read()
            # (template kernels put attributes and line breaks between __global__ and the name)
            for m in re.finditer(r"__global__[^;{]*?\bvoid\s+(k_[a-z0-9_]+)\s*\(", src, re.S):
                names.add(m.group(1))
    return names


def test_every_kernel_is_classified_for_the_traffic_summary():
    """k_bin_hist_multi (new in round 5) was in no list of profiles/summarize_pmc.py and its 56 GB per sample
    dropped out of `roofline.traffic`.  The chain is now "everything that is not known to be another stage's"; a kernel
    nobody has classified fails here, before a profile is taken with it."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("summarize_pmc", os.path.join(ROOT, "profiles", "summarize_pmc.py"))
    sp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sp)
    kernels = _kernels_in_sources()
    assert {"k_msp_leaf", "k_bin_hist_multi", "k_filter_q", "k_part2"} <= kernels
    both = sp.CHAIN_KNOWN & sp.NOT_CHAIN
    assert not both, f"in both lists: {sorted(both)}"
    unknown = kernels - sp.CHAIN_KNOWN - sp.NOT_CHAIN
    assert not unknown, f"kernels in no list of profiles/summarize_pmc.py: {sorted(unknown)}"


def test_latest_pmc_profile_counts_every_chain_kernel_it_saw():
    import importlib.util
    spec = importlib.util.spec_from_file_location("summarize_pmc", os.path.join(ROOT, "profiles", "summarize_pmc.py"))
    sp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sp)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_*.json")))
    files = [f for f in files if int(os.path.basename(f)[1:3]) >= 6]
    if not files:
        pytest.skip("no round-6 PMC profile yet")
    for f in files:
        pmc = json.load(open(f))
        seen = {k for k, v in pmc.items() if isinstance(v, dict) and "launches" in v and k.startswith("k_")}
        want = seen - sp.NOT_CHAIN
        assert set(pmc["_chain"]["kernels"]) == want, (os.path.basename(f), sorted(want ^ set(pmc["_chain"]["kernels"])))


def test_kernel_trace_summary_is_one_clean_step():
    """the kept kernel statistics must be those of ONE step as bench.py times it: launches per kernel =
    launches per chain (the bench line's own HIP-event brackets) x the chains of a step."""
    import csv
    stats = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_kernel_stats_wgs.csv")))
    stats = [f for f in stats if int(os.path.basename(f)[1:3]) >= 6]
    if not stats:
        pytest.skip("no round-6 kernel statistics yet")
    f = stats[-1]
    rr = os.path.basename(f)[:3]
    meta = json.load(open(f + ".meta.json"))
    assert meta["steps_in_trace"] >= 2 and meta["step_taken"] == meta["steps_in_trace"] - 1
    line = json.load(open(os.path.join(ROOT, "profiles", rr + "_bench.json")))
    per_chain = line["roofline"]["launches_by_kernel_per_chain"]
    chains = 3  # samples per trio step
    calls = {}
    for r in csv.DictReader(open(f)):
        import re
        m = re.search(r"(k_[a-z0-9_]+)", r["Name"])
        if m:
            calls[m.group(1)] = calls.get(m.group(1), 0) + int(r["Calls"])
    for label, kernel in (("k_msp_leaf", "k_msp_leaf"), ("k_msp_replay", "k_msp_replay"), ("k_surv_sort", "k_surv_sort")):
        assert calls[kernel] == round(per_chain[label] * chains), (kernel, calls[kernel], per_chain[label])
