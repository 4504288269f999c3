#!/usr/bin/env python3
"""Regenerates the fixtures under tests/golden/ (run in the build container, where /root/reference exists).

Inputs copied verbatim (data files of the reference's own integration test, testRun/runTest.fastq.sh:9):
  testRun/{Child,Mother,Father}.mate{1,2}.fastq  -> gzip
Expected outputs:
  * filter: read names pulled by the REAL reference binaries oracle/_ref/RUFUS.Filter and
    RUFUS.Filter.single (built from /root/reference/src by `make -C oracle ref`) at 1 thread;
  * count / histo / merge / hash list: produced by the oracle restatement, which is itself pinned by
    jellyfish's md5 known-answer tests and by the probe values of SURVEY.md (18 356 / 18 364 / 17 390
    records, 411 merge lines, 50 hash-list k-mers, first record = poly-A x48) -- asserted below.
"""
import gzip
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

REF = "/root/reference/testRun"
OUT = os.path.join(HERE, "testRun")
K, MINQ = 25, 15


def main():
    exp = {"k": K, "samples": {}}
    recs = {}
    for s in ("Child", "Mother", "Father"):
        texts = []
        for m in (1, 2):
            data = open(f"{REF}/{s}.mate{m}.fastq", "rb").read()
            with gzip.GzipFile(f"{OUT}/{s}.mate{m}.fastq.gz", "wb", mtime=0) as f:
                f.write(data)
            texts.append(data)
        e = {}
        for label, size in (("s100M", 100_000_000), ("s8G", 8 << 30)):
            r = oracle.count(texts, K, size, lower=2)
            e[label] = {"records": len(r.keys), "payload_sha256": hashlib.sha256(r.payload()).hexdigest(),
                        "histo_full_md5": hashlib.md5(oracle.histo(r.counts, full=True)[1].encode()).hexdigest(),
                        "first_record_hex": r.payload()[:11].hex(), "max_count": int(r.counts.max()),
                        "sum_counts": int(r.counts.sum())}
            if label == "s100M":
                recs[s] = r
        exp["samples"][s] = e
    assert [exp["samples"][s]["s100M"]["records"] for s in ("Child", "Mother", "Father")] == [18356, 18364, 17390]
    assert exp["samples"]["Child"]["s100M"]["first_record_hex"] == "0000000000000030000000"
    merge = oracle.merge_unique_text([recs["Child"], recs["Mother"], recs["Father"]])
    assert merge.count("\n") == 411
    hl = oracle.hash_list(recs["Child"], [recs["Mother"], recs["Father"]], 5, 140)
    assert hl.count("\n") == 50
    open(f"{OUT}/merge.Child.Mother.Father.txt", "w").write(merge)
    open(f"{OUT}/Child.k25_c5.HashList", "w").write(hl)
    # exclude-list variant of testRun/runDevTest.sh (-e Mother.Jhash, -m 8): Mother appears twice in the merge
    hl_dev = oracle.hash_list(recs["Child"], [recs["Mother"], recs["Father"], recs["Mother"]], 8, 140)
    open(f"{OUT}/Child.k25_c8.dev.HashList", "w").write(hl_dev)

    # the real reference filter binaries
    d = tempfile.mkdtemp()
    for threads in ("1",):
        subprocess.run([f"{ROOT}/oracle/_ref/RUFUS.Filter", f"{OUT}/Child.k25_c5.HashList", f"{REF}/Child.mate1.fastq",
                        f"{REF}/Child.mate2.fastq", d + "/p", str(K), str(MINQ), "1", threads],
                       stdout=subprocess.DEVNULL, check=True)
    m1 = open(d + "/p.Mutations.Mate1.fastq").read().split("\n")
    exp["filter_paired_names"] = [l for i, l in enumerate(m1) if i % 4 == 0 and l]
    exp["filter_paired_sha256"] = {
        m: hashlib.sha256(open(f"{d}/p.Mutations.Mate{m}.fastq", "rb").read()).hexdigest() for m in (1, 2)}
    assert len(exp["filter_paired_names"]) == 26
    subprocess.run([f"{ROOT}/oracle/_ref/RUFUS.Filter.single", f"{OUT}/Child.k25_c5.HashList",
                    f"{REF}/Child.mate1.fastq", d + "/s", str(K), str(MINQ), "1", "1"], stdout=subprocess.DEVNULL,
                   check=True)
    exp["filter_single_sha256"] = hashlib.sha256(open(d + "/s.Mutations.fastq", "rb").read()).hexdigest()
    exp["filter_single_names"] = [l for i, l in enumerate(open(d + "/s.Mutations.fastq").read().split("\n"))
                                  if i % 4 == 0 and l]
    json.dump(exp, open(f"{OUT}/expected.json", "w"), indent=1, sort_keys=True)
    print(json.dumps({k: v for k, v in exp.items() if k != "filter_paired_names"}, indent=1)[:1500])


if __name__ == "__main__":
    main()
