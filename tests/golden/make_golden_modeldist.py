#!/usr/bin/env python3
"""Golden vectors of row N4 (coverage model fit), made by the REFERENCE binary: `oracle/_ref/ModelDist`, compiled by
`make -C oracle ref` from /root/reference/src/ModelDist.cpp + Util.cpp where they lie.  Run here (the container that
holds /root/reference); the fixtures travel, the reference does not.

Inputs (tab-separated `jellyfish histo` tables, as runRufus.sh:830 makes them):
  child.histo        k = 25 histogram (-L 2, full, high 10000) of testRun's Child reads, counted by the oracle
  child1200.histo    the same with `-h 1200` (rows above 1200 folded into the last one): the CPU suite's size
  wgs1200.histo      a synthetic 30x-like shape (seeded): error curve, half-copy shoulder, 1x..3x peaks
Outputs per input X: X.out (stdout), X.7.7.model, X.7.7.dist.gz, X.7.7.prob.gz.
"""
import gzip, os, subprocess, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "modeldist")
sys.path.insert(0, ROOT)


def child_histo(high):
    import oracle
    g = os.path.join(HERE, "testRun")
    fq = [gzip.open(os.path.join(g, f"Child.mate{m}.fastq.gz")).read() for m in (1, 2)]
    r = oracle.count(fq, 25, 100_000_000, lower=2)
    return oracle.histo(r.counts, high=high, full=True)[1].replace(" ", "\t")


def wgs_histo(high, seed=7):
    rng = np.random.default_rng(seed)
    m = np.arange(high + 2, dtype=np.float64)
    with np.errstate(all="ignore"):
        shape = 4e6 * np.where(m >= 2, m, 1) ** -2.2
    for mu, sd, amp in ((15, 4.2, 3e5), (30, 5.6, 2.5e6), (60, 8.0, 2e5), (90, 10.0, 5e4), (120, 12.0, 1e4)):
        shape += amp * np.exp(-0.5 * ((m - mu) / sd) ** 2)
    c = rng.poisson(shape).astype(np.int64)
    c[:2] = 0
    c[high + 1] = 1234
    return "".join(f"{i}\t{int(v)}\n" for i, v in enumerate(c))


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = os.path.join(ROOT, "oracle", "_ref", "ModelDist")
    inputs = {"child": child_histo(10000), "child1200": child_histo(1200), "wgs1200": wgs_histo(1200)}
    for name, text in inputs.items():
        path = os.path.join(OUT, name + ".histo")
        open(path, "w").write(text)
        out = subprocess.run([ref, name + ".histo", "25", "150", "8"], cwd=OUT, capture_output=True, text=True)
        assert out.returncode == 0, out
        open(os.path.join(OUT, name + ".out"), "w").write(out.stdout)
        os.rename(path + ".7.7.model", os.path.join(OUT, name + ".7.7.model"))
        for ext in (".7.7.dist", ".7.7.prob"):
            with open(path + ext, "rb") as f, gzip.GzipFile(os.path.join(OUT, name + ext + ".gz"), "wb", mtime=0) as z:
                z.write(f.read())
            os.remove(path + ext)
        print(name, open(os.path.join(OUT, name + ".7.7.model")).read().split("\n")[:4],
              [l for l in out.stdout.split("\n") if l.startswith("Best Model")])


if __name__ == "__main__":
    main()
