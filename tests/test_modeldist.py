"""Row N4 -- the coverage model fit (reference src/ModelDist.cpp, runRufus.sh:849).

Golden vectors: `tests/golden/modeldist/*` are outputs of the REFERENCE binary (`oracle/_ref/ModelDist`, made by
`tests/golden/make_golden_modeldist.py` in the container that holds the reference).  CPU tests pin the numpy
restatement (`oracle/modeldist.py`) to them; GPU tests put the `ModelDist` executable and the C-ABI underneath it
(`rfx_model_residuals`, `rfx_model_tables`) against the golden files and the restatement.

Tolerance (binary64 throughout): every number of the text outputs within 2e-5 relative (they are printed with 6
significant digits) of the reference's; the four header lines and every integer column exact.  Residuals through
the C-ABI: 1e-9 relative against the restatement (libm vs device exp/log, tree vs sequential column sums).
"""
import gzip
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "modeldist")
BIN = os.path.join(ROOT, "rufus_amd", "bin")
EXTS = (".7.7.model", ".7.7.dist", ".7.7.prob")


def golden(name, ext):
    path = os.path.join(GOLD, name + ext)
    if os.path.exists(path + ".gz"):
        return gzip.open(path + ".gz", "rt").read()
    return open(path).read()


def same_text(got, want, what, rel=2e-5):
    """Token by token: equal strings, or two numbers within `rel` (denormal cells: absolute 1e-300)."""
    gl, wl = got.split("\n"), want.split("\n")
    assert len(gl) == len(wl), f"{what}: {len(gl)} lines, expected {len(wl)}"
    for ln, (g, w) in enumerate(zip(gl, wl)):
        if g == w:
            continue
        gt, wt = g.split("\t"), w.split("\t")
        assert len(gt) == len(wt), f"{what}:{ln + 1}: {len(gt)} fields, expected {len(wt)}"
        for a, b in zip(gt, wt):
            if a == b:
                continue
            if " " in a or " " in b:       # log lines: "name = value name = value"
                ta, tb = a.split(" "), b.split(" ")
                assert len(ta) == len(tb), f"{what}:{ln + 1}: {a!r} != {b!r}"
                pairs = zip(ta, tb)
            else:
                pairs = [(a, b)]
            for x, y in pairs:
                if x == y:
                    continue
                try:
                    fx, fy = float(x), float(y)
                except ValueError:
                    raise AssertionError(f"{what}:{ln + 1}: {x!r} != {y!r}")
                assert abs(fx - fy) <= rel * max(abs(fx), abs(fy)) + 1e-300, f"{what}:{ln + 1}: {x} != {y}"


def header(text):
    return text.split("\n")[:4]


# ------------------------------------------------------------------------------------------------
# CPU: the restatement against the reference binary's outputs
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["child1200", "wgs1200"])
def test_restatement_reproduces_the_reference_binary(name):
    from oracle import modeldist
    rc, out, files = modeldist.model_dist(open(os.path.join(GOLD, name + ".histo")).read(), 25, 150, 8,
                                          name=name + ".histo")
    assert rc == 0
    same_text(out, open(os.path.join(GOLD, name + ".out")).read(), name + ".out")
    for ext in EXTS:
        assert header(files[ext]) == header(golden(name, ext))
        same_text(files[ext], golden(name, ext), name + ext)


@pytest.mark.skipif(not os.environ.get("RFX_SLOW"), reason="4 minutes of numpy: the 10002-row table (RFX_SLOW=1)")
def test_restatement_reproduces_the_reference_binary_at_full_height():
    from oracle import modeldist
    rc, out, files = modeldist.model_dist(open(os.path.join(GOLD, "child.histo")).read(), 25, 150, 8, name="child.histo")
    same_text(out, open(os.path.join(GOLD, "child.out")).read(), "child.out")
    for ext in EXTS:
        same_text(files[ext], golden("child", ext), "child" + ext)


def test_golden_headers_are_the_probe_values():
    """testRun's Child histogram gives model lines 1-4 = 3 / 5 / 17080 / 28."""
    assert header(golden("child", ".7.7.model")) == ["3", "5", "17080", "28"]
    assert header(golden("child", ".7.7.dist")) == ["3", "5", "17080", "28"]


def test_restatement_stops_on_a_table_without_kmers():
    from oracle import modeldist
    rc, out, files = modeldist.model_dist("".join(f"{i}\t0\n" for i in range(40)), 25, 150)
    assert rc == 1 and out.endswith("ERROR there are no kmers in this file\n") and not files


# ------------------------------------------------------------------------------------------------
# GPU: the executable and the C-ABI
# ------------------------------------------------------------------------------------------------
def run_tool(tmp_path, name, text=None):
    path = tmp_path / (name + ".histo")
    path.write_text(text if text is not None else open(os.path.join(GOLD, name + ".histo")).read())
    r = subprocess.run([f"{BIN}/ModelDist", path.name, "25", "150", "8"], cwd=tmp_path, capture_output=True, text=True,
                       timeout=300)
    return r


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["child", "child1200", "wgs1200"])
def test_executable_writes_the_reference_binarys_files(tmp_path, name):
    r = run_tool(tmp_path, name)
    assert r.returncode == 0, r.stderr
    same_text(r.stdout, open(os.path.join(GOLD, name + ".out")).read(), name + ".out")
    for ext in EXTS:
        got = (tmp_path / (name + ".histo" + ext)).read_text()
        assert header(got) == header(golden(name, ext))       # MutantMinCov and MutantSC (runRufus.sh:862-868)
        same_text(got, golden(name, ext), name + ext)


@pytest.mark.gpu
def test_three_fits_at_once_as_the_pipeline_starts_them(tmp_path):
    """runRufus.sh:849-853 starts the parents' fits in the background while the proband's runs: three processes on the
    one device at the same time, each with its own context."""
    names = ["child", "child1200", "wgs1200"]
    procs = []
    for n in names:
        (tmp_path / (n + ".histo")).write_text(open(os.path.join(GOLD, n + ".histo")).read())
        procs.append(subprocess.Popen([f"{BIN}/ModelDist", n + ".histo", "25", "150", "8"], cwd=tmp_path,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for n, pr in zip(names, procs):
        out, err = pr.communicate(timeout=300)
        assert pr.returncode == 0, err
        same_text(out, open(os.path.join(GOLD, n + ".out")).read(), n + ".out")
        assert header((tmp_path / (n + ".histo.7.7.model")).read_text()) == header(golden(n, ".7.7.model"))


@pytest.mark.gpu
def test_wall_time_beside_the_reference_binary(tmp_path):
    """The executable next to `oracle/_ref/ModelDist` (the reference's source, `g++ -O2 -fopenmp`, its 11 OpenMP
    threads) on this box's host cores, same table; the figures go to gpurun_out/modeldist_wall.txt."""
    import time
    ref = os.path.join(ROOT, "oracle", "_ref", "ModelDist")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/ModelDist not built")
    run_tool(tmp_path, "child")                     # first start of the HIP runtime on this box
    t0 = time.time()
    r = run_tool(tmp_path, "child")
    ours = time.time() - t0
    assert r.returncode == 0
    (tmp_path / "ref").mkdir()
    (tmp_path / "ref" / "child.histo").write_text(open(os.path.join(GOLD, "child.histo")).read())
    t0 = time.time()
    rr = subprocess.run([ref, "child.histo", "25", "150", "8"], cwd=tmp_path / "ref", capture_output=True, text=True)
    theirs = time.time() - t0
    assert rr.returncode == 0
    same_text((tmp_path / "child.histo.7.7.model").read_text(), (tmp_path / "ref" / "child.histo.7.7.model").read_text(),
              "model file vs the reference binary run beside it")
    line = f"ModelDist child.histo (10002 rows): this repo {ours:.3f} s wall (process start to exit), reference binary " \
           f"{theirs:.1f} s on {os.cpu_count()} host cpus -> x{theirs / ours:.0f}\n"
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out) and os.access(out, os.W_OK):
        open(os.path.join(out, "modeldist_wall.txt"), "w").write(line)
    print(line)
    assert theirs > 5 * ours


@pytest.mark.gpu
def test_executable_edge_inputs(tmp_path):
    # no k-mers at all: the reference's message and exit status 1 (src/ModelDist.cpp:443-447)
    r = run_tool(tmp_path, "empty", "".join(f"{i}\t0\n" for i in range(40)))
    assert r.returncode == 1 and r.stdout.endswith("ERROR there are no kmers in this file\n")
    # a missing table: message on stdout, exit 0 (:376-379)
    r = subprocess.run([f"{BIN}/ModelDist", "nowhere.histo", "25", "150", "8"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0 and "Error, HistoFile could not be opened" in r.stdout
    # a table that only falls (no inflection) or is too short: a message instead of the reference's crash
    r = run_tool(tmp_path, "falling", "0\t0\n" + "".join(f"{i}\t{5000 // i}\n" for i in range(1, 300)))
    assert r.returncode == 1 and "inflection" in r.stderr
    # space-separated (runRufus.sh:830 not applied)
    r = run_tool(tmp_path, "spaces", "".join(f"{i} {i}\n" for i in range(40)))
    assert r.returncode == 1 and "tab-separated" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_residuals_match_the_restatement_on_random_models(ctx, seed):
    """rfx_model_residuals against testModel / testModelLog of the restatement: random candidates (factor and skew
    switched on too -- the reference's own search never leaves skew = 0), both residual kinds, 1202- and 3000-row tables."""
    from oracle import modeldist
    from rufus_amd import capi
    rng = np.random.default_rng(seed)
    n = int(rng.choice([1201, 3000]))
    m = np.arange(n, dtype=np.float64)
    sc0 = float(rng.uniform(18, 60))
    shape = 3e6 * np.maximum(m, 1) ** -2.3 + 2e6 * np.exp(-0.5 * ((m - sc0) / (sc0 / 5)) ** 2) + \
        3e5 * np.exp(-0.5 * ((m - sc0 / 2) / (sc0 / 7)) ** 2) + 2e5 * np.exp(-0.5 * ((m - 2 * sc0) / (sc0 / 3.5)) ** 2) + 3
    hist = rng.poisson(shape).astype(np.int64)
    hist[0] = 0
    cands = np.stack([rng.uniform(sc0 * .85, sc0 * 1.15, 11), rng.uniform(sc0 / 7, sc0 / 3, 11), rng.uniform(0, 6, 11),
                      rng.choice([0.0, 0.0, 0.02, 0.1], 11), rng.uniform(1, 2, 11)], axis=1)
    infl = int(rng.integers(3, 9))
    for log in (True, False):
        got = capi.model_residuals(ctx, hist, cands, log, infl, 5)
        want = np.array([modeldist._test_model(log, *c, hist, infl, 5) for c in cands])
        assert np.all(np.isfinite(want))
        np.testing.assert_allclose(got, want, rtol=1e-9)
        assert int(np.argmin(got)) == int(np.argmin(want))
    # zeros inside the compared rows: ln 0 on both sides of the difference, as the reference gets it
    holes = hist.copy()
    holes[infl + 2] = 0
    got = capi.model_residuals(ctx, holes, cands, True, infl, 5)
    want = np.array([modeldist._test_model(True, *c, holes, infl, 5) for c in cands])
    assert np.all(np.isinf(want)) and np.array_equal(np.isinf(got), np.isinf(want))
    # the tables of main(): columns summed from row 0, the last column raw
    dist, rowtot = capi.model_tables(ctx, n, cands[0])
    jn, want_d, want_t = modeldist._tables(hist, *cands[0], 0)
    assert dist.shape == want_d.shape
    np.testing.assert_allclose(dist, want_d, rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(rowtot, want_t, rtol=1e-9, atol=1e-300)


@pytest.mark.gpu
def test_model_entry_points_refuse_what_the_reference_would_overrun(ctx):
    from rufus_amd import capi
    hist = np.ones(500, dtype=np.int64)
    with pytest.raises(capi.RufusError):       # 5 x SC beyond the table
        capi.model_residuals(ctx, hist, [[120.0, 10, 1, 0, 1]], True, 5)
    with pytest.raises(capi.RufusError):       # SC / 2 < 1
        capi.model_residuals(ctx, hist, [[1.5, 1, 1, 0, 1]], True, 5)
    with pytest.raises(capi.RufusError):       # more than 64 candidates
        capi.model_residuals(ctx, hist, np.tile([30.0, 6, 1, 0, 1], (65, 1)), True, 5)
    with pytest.raises(capi.RufusError):
        capi.model_residuals(ctx, hist, [[float("nan"), 6, 1, 0, 1]], True, 5)
