#!/usr/bin/env python3
"""Headline benchmark: reads/s through the k-mer count + set-difference + read-filter hot path at k=25
(BASELINE.json metric), with the achieved HBM GB/s of the dominant stage against the roofline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload wgs|s1] [--genome BASES] [--passes S]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Default workload = BASELINE.json configs[2] (the config the metric is quoted on): synthetic 30x WGS trio,
3.1 Gb random genome, 3.1e8 pairs of 150 bp per sample (6.2e8 reads), 0.5 % substitution errors, 2 % low-
quality bases, 0.1 % N, 1000 heterozygous SNVs planted in the child; k = 25, jellyfish -s 8G -L 2, MinCov 5,
MaxHashDepth 1200, MinQ 15, HashCountThreshold 1.  The reads are generated on the device (counter-based
generator, rufus_amd/csrc/rfx_synth.h) as packed read blocks and stay resident in HBM: 138 GB for the trio.

One step = the whole trio through the path: S minimizer-shard passes (a sample's 187 GB of super-k-mer
records do not fit beside the reads; S is planned from the free HBM), each counting shard s of the three
samples (K2) -> sorted records + histogram (K3) -> subject-minus-controls on the shard (K4); then the filter of
the subject's reads against the mutant k-mers (K5).  value = reads through the count stage (1.86e9 per step)
/ wall time.  --genome scales the workload down (same coverage); --workload s1 is configs[1] (1 M reads per
sample, the round-1 bench).

N > 1: the same trio, read blocks dealt to the ranks (strong scaling): every rank partitions its blocks into
super-k-mer records, the records travel to the owner of their minimizer bin (RCCL all-to-all over xGMI),
owners count complete bins; histograms all-reduce, mutant k-mers all-gather, every rank filters its blocks.
"""
import argparse
import filecmp
import gc
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K, JF_SIZE, LOWER = 25, 8 << 30, 2
MIN_COV, MAX_DEPTH, MIN_Q, THRESH = 5, 1200, 15, 1
READ_LEN = int(os.environ.get("RFX_BENCH_READ_LEN", "150"))   # (sweeps away from the headline geometry: the GPU part of --workload wgs / tn only)
SEED = 12345

# every launch of the count -> sorted-records stage, whichever path rfx_count_add/finish took
K2_CHAIN = ("k_msp_part1", "k_msp_map", "k_msp_replay", "k_msp_count", "k_bin_offsets", "k_rec_hist", "k_part2", "k_bin_hist", "k_part3", "k_part4", "k_msp_leaf", "k_surv_place",
            "k_surv_hist", "k_surv_part2", "k_surv_part3", "k_surv_sort", "k_histo",
            "k_bin_count", "k_bin_scatter", "k_part1", "k_leaf", "k_leaf_compact", "k_count_reads")


def kernel_source_fingerprint():
    """sha256 (16 hex digits) over the sources the device code is built from: what a PMC profile must have been taken on
    to be quoted by this build (profiles/summarize_pmc.py writes it into the JSON)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "rufus_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")) or name == "Makefile":
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def algorithmic_bytes_per_read(L=READ_LEN, k=K):
    """ceil(L/4) code bytes + ceil(L/8) mask bytes + (L-k+1) windows x
    (8 B key read + 4 B count read + 4 B count write)."""
    return (L + 3) // 4 + (L + 7) // 8 + (L - k + 1) * 16


def cpu_baseline():
    """The same path on the host cores, on a bounded sample of the SAME workload (first pairs of each sample
    of the synthetic trio, regenerated as text by the generator's host twin): the oracle's C++ port for
    count (lock-free CAS hash table, mirrors jf/include/jellyfish/large_hash_array.hpp:708-744) and set
    difference, the REAL reference binary oracle/_ref/RUFUS.Filter (built from /root/reference/src in the
    build container) for the filter; at T = 1, 8 and the CPUs the cgroup lets the process use (rfx_host_cpus: the GPU
    box gives its container 16 CPUs' worth of time on 256 hardware threads).
    The filter leg runs on the first 60 k pairs of the subject and is scaled to the leg's read count: the
    reference forks its OpenMP team once per 60 pairs (src/RUFUS.Filter.cpp:196)."""
    import oracle
    from rufus_amd import capi
    from tests.synth import synth_fastq
    ncpu = os.cpu_count() or 1
    usable = int(capi.lib().rfx_host_cpus())         # affinity and cgroup CPU quota: 16 on the 256-thread GPU box
    out = {"unit": "reads/s", "kind": "port", "by_threads": {}, "usable_cpus": usable}
    exe = os.path.join(ROOT, "oracle", "_ref", "RUFUS.Filter")
    n_filter = 60_000
    # (until round 3 a fourth leg ran at nproc - 2 = 254 threads, runRufus.sh:796,:967's choice: on the GPU box that
    # measures its container's 16-CPU quota -- 3.9 k reads/s -- and cost 90 s of every run; dropped)
    for T in sorted({1, min(8, ncpu), usable}):
        n_pairs = int(min(1_000_000, 40_000 * T ** 0.8))     # ~ equal wall time per leg
        G = n_pairs * 10
        sys_ = [capi.Synth.sample(G, w, n_snv=max(4, G // 1_000_000), seed=SEED) for w in range(3)]
        texts = [sy.text(0, n_pairs) for sy in sys_]
        t0 = time.perf_counter()
        recs = [oracle.count_reads_matrix(seq, K, JF_SIZE, lower=LOWER, threads=T) for seq, _ in texts]
        t_count = time.perf_counter() - t0
        t0 = time.perf_counter()
        hl = oracle.hash_list(recs[0], recs[1:], MIN_COV, MAX_DEPTH)
        t_merge = time.perf_counter() - t0
        seq, qual = texts[0]
        nf = min(n_filter, n_pairs)
        m1, m2 = synth_fastq(seq[0:2 * nf:2], qual[0:2 * nf:2]), synth_fastq(seq[1:2 * nf:2], qual[1:2 * nf:2])
        note = ""
        if os.path.exists(exe):
            d = tempfile.mkdtemp(prefix="rfx_cpu_")
            for m, data in ((1, m1), (2, m2)):
                open(f"{d}/m{m}.fq", "wb").write(data)
            open(f"{d}/hl", "w").write(hl)
            t0 = time.perf_counter()
            try:
                subprocess.run([exe, f"{d}/hl", f"{d}/m1.fq", f"{d}/m2.fq", f"{d}/o", str(K), str(MIN_Q), str(THRESH), str(T)],
                               stdout=subprocess.DEVNULL, check=True, timeout=90)
            except subprocess.TimeoutExpired:
                note = " (stopped after 90 s: a lower bound of its time)"
            t_filter = (time.perf_counter() - t0) * n_pairs / nf
            filt = "reference binary oracle/_ref/RUFUS.Filter (-O2)"
            out["kind"] = "port (count, set difference) + reference (filter)"
        else:
            fs = oracle.FilterSet(hl.encode())
            t0 = time.perf_counter()
            fs.pairs(m1, m2, K, MIN_Q, THRESH)
            t_filter = (time.perf_counter() - t0) * n_pairs / nf
            filt = "oracle port of RUFUS.Filter (1 thread)"
        reads = 3 * 2 * n_pairs
        total = t_count + t_merge + t_filter
        out["by_threads"][str(T)] = {"reads_per_s": reads / total, "reads": reads, "count_s": round(t_count, 2),
                                     "set_difference_s": round(t_merge, 2),
                                     "filter_s_scaled_from_%d_pairs" % nf: round(t_filter, 2), "note": note.strip()}
        out["filter_tool"] = filt
    best = max(out["by_threads"], key=lambda t_: out["by_threads"][t_]["reads_per_s"])
    out["value"] = out["by_threads"][best]["reads_per_s"]
    out["cores"] = int(best)
    out["host_cores"] = ncpu
    out["sample"] = ("first pairs of each sample of the same synthetic trio at 30x on a proportionally smaller genome "
                     "(reads per leg in by_threads), k=25: count = CAS hash-table port (oracle), set difference = oracle, "
                     f"filter = {out['filter_tool']} on the subject's first {n_filter} pairs, scaled")
    return out


def end_to_end(n_pairs=32_000_000):
    """the path through the DROP-IN EXECUTABLES, text in -> files out, process start and HIP init
    included: rfx_synth_fastq writes a bounded 30x sample of the same synthetic trio as FASTQ (tmpfs when there is
    one); then, as runRufus.sh does, `jellyfish count` x 3 -> modified `jellyfish merge` -> `jellyfish query` + the
    [MinCov, MaxDepth] filter (CheckJellyHashList.sh:12) -> `RUFUS.Filter` -> (no bwa in the image: the pulled pairs
    get their true coordinates from the generator) OverlapSam -> Overlap x 3 -> OverlapRegion -> tail tools."""
    from rufus_amd import capi
    B = os.path.join(ROOT, "rufus_amd", "bin")
    ncpu = os.cpu_count() or 1
    T = str(max(1, min(64, ncpu - 2)))                # (the tools cap it to rfx_host_cpus())
    G = n_pairs * 10
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    # tmpfs pages count against the container's memory limit (300 GiB on the GPU box: a full-size trio, 700 GB of text,
    # gets the whole box killed -- it did), so the leg refuses sizes that would not leave room
    need = n_pairs * 2 * 3 * 330 + n_pairs * 2 * 3 * 60
    try:
        limit = int(open("/sys/fs/cgroup/memory.max").read())
    except (OSError, ValueError):
        limit = None
    if base and limit and need > 0.6 * limit:
        raise RuntimeError(f"end_to_end: {need >> 30} GiB of tmpfs would not fit the container's memory limit ({limit >> 30} GiB)")
    d = tempfile.mkdtemp(prefix="rfx_e2e_", dir=base)
    out = {"reads_per_sample": 2 * n_pairs, "threads": int(T), "stages_s": {}}

    def run(name, args, stdout=None, stdin=None, env=None):
        t0 = time.perf_counter()
        p = subprocess.run(args, cwd=d, stdout=open(os.path.join(d, stdout), "wb") if stdout else subprocess.DEVNULL,
                           stderr=subprocess.PIPE, stdin=stdin, timeout=900, env=dict(os.environ, **env) if env else None)
        if p.returncode != 0:
            raise RuntimeError(f"{name}: rc {p.returncode}: {p.stderr.decode()[-300:]}")
        if os.environ.get("RFX_CLI_TRACE"):            # the tools' phase marks, for whoever asked for them
            sys.stderr.write(f"--- {name}\n" + p.stderr.decode())
        out["stages_s"][name] = round(out["stages_s"].get(name, 0.0) + time.perf_counter() - t0, 3)

    try:
        n_snv = max(8, G // 1_000_000)
        t_gen = time.perf_counter()
        for w, name in enumerate(("child", "mother", "father")):
            if w == 0:
                run("generate", [f"{B}/rfx_synth_fastq", str(G), "0", str(n_snv), str(SEED), "0", str(n_pairs), "c.m1.fq", "c.m2.fq"])
            else:
                run("generate", [f"{B}/rfx_synth_fastq", str(G), str(w), str(n_snv), str(SEED), "0", str(n_pairs), f"{name}.fq"])
        out["generate_s (not counted)"] = round(time.perf_counter() - t_gen, 2)
        out["stages_s"].pop("generate", None)
        t0 = time.perf_counter()
        for name in ("child", "mother", "father"):
            files = ["c.m1.fq", "c.m2.fq"] if name == "child" else [f"{name}.fq"]   # (the filter wants the mates apart)
            # RFX_COUNT_HISTO=1: the count also leaves NAME.Jhash.histo -- the bytes of `jellyfish histo -f` --,
            # and scripts/RunJellyForRUFUS.sh:36-38 runs histo only `if [ ! -s $GEN.Jhash.histo ]`: so does this leg
            run("jellyfish count", [f"{B}/jellyfish", "count", "--disk", "-m", str(K), "-L", str(LOWER), "-s", "8G", "-t", T,
                                    "-o", f"{name}.Jhash", "-C"] + files, env={"RFX_COUNT_HISTO": "1"})
            hp = os.path.join(d, f"{name}.Jhash.histo")
            if not (os.path.exists(hp) and os.path.getsize(hp) > 0):
                run("jellyfish histo", [f"{B}/jellyfish", "histo", "-f", "-o", f"{name}.Jhash.histo", f"{name}.Jhash"])
            else:
                out["histo"] = "written by jellyfish count (RFX_COUNT_HISTO=1): RunJellyForRUFUS.sh:36 skips its own histo run then"
        run("jellyfish merge", [f"{B}/jellyfish", "merge", "child.Jhash", "mother.Jhash", "father.Jhash"], stdout="merge.txt")
        t1 = time.perf_counter()
        with open(os.path.join(d, "merge.txt")) as f, open(os.path.join(d, "q.fa"), "w") as q:
            for ln in f:
                km = ln.split()[0]
                q.write(f">{km}\n{km}\n")
        out["stages_s"]["awk (python)"] = round(time.perf_counter() - t1, 3)
        run("jellyfish query", [f"{B}/jellyfish", "query", "-s", "q.fa", "child.Jhash"], stdout="query.txt")
        t1 = time.perf_counter()
        with open(os.path.join(d, "query.txt")) as f, open(os.path.join(d, "child.HashList"), "w") as h:
            n_hl = 0
            for ln in f:
                c_ = int(ln.split()[1])
                if MIN_COV <= c_ <= MAX_DEPTH:
                    h.write(ln)
                    n_hl += 1
        out["stages_s"]["awk (python)"] += round(time.perf_counter() - t1, 3)
        run("RUFUS.Filter", [f"{B}/RUFUS.Filter", "child.HashList", "c.m1.fq", "c.m2.fq", "child", str(K), str(MIN_Q),
                             str(THRESH), T])
        t_path = time.perf_counter() - t0
        out["mutant_kmers"] = n_hl
        out["count_to_filter_s"] = round(t_path, 2)
        out["value"] = 3 * 2 * n_pairs / t_path
        # ---- runRufus.sh -pj (--Parallelize_Jelly, runRufus.sh:305,766-782): the three RunJellyForRUFUS.sh at once, each
        # with its share of the threads; here the three drop-in counts share the one GPU (samples of this size fit side by side;
        # a full-size trio wants a device each: RUFUS_GPUS).  Reported beside `value`, never instead of it.
        # (runRufus.sh:769 gives each Threads / 3; Threads = the CPUs this container may use, cgroup quota included)
        tj = os.environ.get("RFX_E2E_PJ_T") or str(max(1, int(capi.lib().rfx_host_cpus()) // 3))
        t1 = time.perf_counter()
        procs = []
        for name in ("child", "mother", "father"):
            files = ["c.m1.fq", "c.m2.fq"] if name == "child" else [f"{name}.fq"]
            procs.append((name, subprocess.Popen(
                [f"{B}/jellyfish", "count", "--disk", "-m", str(K), "-L", str(LOWER), "-s", "8G", "-t", tj, "-o", f"pj.{name}.Jhash",
                 "-C"] + files, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, RFX_COUNT_HISTO="1"))))
        outs = [(name, p, p.communicate(timeout=900)[1]) for name, p in procs]
        errs = [(name, p.returncode, e) for name, p, e in outs if p.returncode != 0]
        t_pj = time.perf_counter() - t1
        if os.environ.get("RFX_CLI_TRACE"):
            for name, p, e in outs:
                sys.stderr.write(f"--- jellyfish count (-pj) {name}\n" + e.decode())
        if errs:
            raise RuntimeError(f"jellyfish count (-pj) {errs[0][0]}: rc {errs[0][1]}: {errs[0][2].decode()[-300:]}")
        for name in ("child", "mother", "father"):
            if not filecmp.cmp(os.path.join(d, name + ".Jhash.histo"), os.path.join(d, "pj." + name + ".Jhash.histo"), shallow=False):
                raise RuntimeError(f"-pj: {name}.Jhash.histo differs from the one-at-a-time count's")
            with open(os.path.join(d, name + ".Jhash"), "rb") as fa, open(os.path.join(d, "pj." + name + ".Jhash"), "rb") as fb:
                for f in (fa, fb):                     # (the header holds argv and the time of day: payloads are compared)
                    f.seek(9 + int(f.read(9)))
                while True:
                    a, b = fa.read(1 << 26), fb.read(1 << 26)
                    if a != b:
                        raise RuntimeError(f"-pj: the records of {name}.Jhash differ from the one-at-a-time count's")
                    if not a:
                        break
            os.unlink(os.path.join(d, f"pj.{name}.Jhash"))
        t_seq = out["stages_s"]["jellyfish count"]
        out["parallel_jelly"] = {"flag": "runRufus.sh -pj: the three counts at once, -t " + tj + " each, one GPU shared",
                                 "jellyfish count x 3_s": round(t_pj, 3),
                                 "count_to_filter_s": round(t_path - t_seq + t_pj, 2),
                                 "value": 3 * 2 * n_pairs / (t_path - t_seq + t_pj),
                                 "outputs": "records and .histo byte-identical to the one-at-a-time files"}
        out["unit"] = "reads/s (3 samples counted, subject filtered; FASTQ text in tmpfs -> .Jhash, .histo, HashList, Mutations.Mate*.fastq)"
        # ---- overlap chain on the pulled pairs: a position-sorted SAM from the generator's own coordinates ----
        from tests.synth import _PHI, _U, _mix64, _scale32
        sy = capi.Synth.sample(G, 0, n_snv=n_snv, seed=SEED)
        comp = bytes.maketrans(b"ACGTN", b"TGCAN")
        m1 = open(os.path.join(d, "child.Mutations.Mate1.fastq"), "rb").read().split(b"\n")
        m2 = open(os.path.join(d, "child.Mutations.Mate2.fastq"), "rb").read().split(b"\n")
        rows = []
        for i in range(0, len(m1) - 1, 4):
            pair = int(m1[i][2:].split(b"/")[0])
            with np.errstate(over="ignore"):
                key = _mix64(_U(sy.read_seed) ^ (_U(pair) * _PHI + _U(1)))
                k1 = _mix64(key + _U(1))
                start = int(_scale32(key, G - (sy.insert_lo + sy.insert_span)))
                end = start + sy.insert_lo + int(_scale32(k1, sy.insert_span))
            name = m1[i][1:].split(b"/")[0]
            rows.append((start, name, b"99", m1[i + 1], m1[i + 3]))
            rows.append((end - READ_LEN, name, b"147", m2[i + 1].translate(comp)[::-1], m2[i + 3][::-1]))
        rows.sort(key=lambda r: r[0])
        with open(os.path.join(d, "pulled.sam"), "wb") as f:
            for pos, name, flag, s_, q_ in rows:
                f.write(b"\t".join([name, flag, b"chr1", str(pos + 1).encode(), b"60", b"150M", b"=", b"1", b"0", s_, q_,
                                    b"NM:i:0"]) + b"\n")
        t0 = time.perf_counter()
        ov = {}
        out["stages_s"], keep = ov, out["stages_s"]
        run("OverlapSam", [f"{B}/OverlapSam", "pulled.sam", ".95", "20", "1", "ov.sam", "N", "1", "child.HashList", T])
        run("Overlap", [f"{B}/Overlap", "ov.sam.fastqd", ".98", "100", "1", "FP", "20", "1", "ov.1", "0", "1"])
        run("Overlap", [f"{B}/Overlap", "ov.1.fastqd", ".98", "75", "2", "FP", "20", "1", "ov.2", "1", "1"])
        run("Overlap", [f"{B}/Overlap", "ov.2.fastqd", ".98", "50", "2", "N", "20", "1", "ov.3", "1", "1"])
        run("OverlapRegion", [f"{B}/OverlapRegion", "ov.3.fastqd", ".98", "50", "5", "ov.4", "N", "1", "1"])
        run("tail", [f"{B}/ReplaceQwithDinFASTQD", "ov.4.fastqd"], stdout="ov.overlap.fastqd")
        run("tail", [f"{B}/ConvertFASTqD.to.FASTQ", "ov.overlap.fastqd"], stdout="ov.overlap.fastq")
        run("tail", [f"{B}/AnnotateOverlap", "child.HashList", "ov.overlap.fastq", "ov.hash.fastq"], stdout="ov.hashcount.fastq")
        out["stages_s"] = keep
        out["overlap_wall_s"] = round(time.perf_counter() - t0, 2)
        out["overlap"] = {"sam_records": len(rows), "stages_s": ov,
                          "contigs": open(os.path.join(d, "ov.hashcount.fastq"), "rb").read().count(b"\n") // 4}
    finally:
        subprocess.run(["rm", "-rf", d])
    return out


def run_s1(args, ctx, rank, world, dist, torch):
    """BASELINE.json configs[1]: 1 M x 150 bp reads per sample (per GPU), the round-1 workload."""
    from rufus_amd import capi
    from rufus_amd.dist import TrioShard
    from tests.synth import make_trio, flat_reads
    trio = make_trio(genome_len=5_000_000, n_pairs=500_000, n_snv=20, seed=SEED, read_seed=1000 + rank)
    blocks = {}
    for name in ("child", "mother", "father"):
        seq, qual, off = flat_reads(trio[name])
        blocks[name] = ctx.upload(capi.PackedReads(seq, off, qual, MIN_Q, capi.PACK_COUNT | capi.PACK_FILTER))
    n_reads = blocks["child"].n
    shard = TrioShard(ctx, K, JF_SIZE, LOWER, MIN_COV, MAX_DEPTH, THRESH, capacity=1 << 26,
                      group=dist.group.WORLD if world > 1 else None)
    step = lambda: shard.run(blocks["child"], [blocks["mother"], blocks["father"]])  # noqa: E731
    desc = (f"synthetic trio, {n_reads} x {READ_LEN} bp reads per sample per GPU (genome 5000000 bp, 20 SNVs, seed "
            f"{SEED}), k={K}, -s 8G -L {LOWER}, MinCov {MIN_COV}, MaxHashDepth {MAX_DEPTH}, MinQ {MIN_Q}, thresh {THRESH}")
    return step, 3 * n_reads * world, n_reads, n_reads * world, desc, "weak", {"passes": 1}


def run_wgs(args, ctx, rank, world, dist, torch):
    from rufus_amd import capi, wgs
    G = args.genome
    tn = args.workload == "tn"      # BASELINE configs[4]: tumor 60x / normal 30x, one control, k = 31
    k = args.k or (31 if tn else K)
    covs = [2 * args.coverage, args.coverage] if tn else [args.coverage] * 3
    pairs = [G * c_ // (2 * READ_LEN) for c_ in covs]
    n_pairs = pairs[0]
    n_snv = max(20, min(1000, G // 3_000_000))
    sys_ = [capi.Synth.sample(G, w, n_snv=n_snv, seed=SEED, read_len=READ_LEN) for w in range(len(covs))]
    free0, total = torch.cuda.mem_get_info()
    # this rank's share of every sample: pairs [p0, p1) (strong scaling: the trio is the same for every N)
    # bytes per pair, resident: codes + ACGT mask + offsets -- or the compact block form (rufus_hip.h RFX_SYNTH_COMPACT:
    # reads of one length need no offsets, and only the ~14 % of reads with an N keep a mask): 43 instead of 68 B/read
    compact = not args.dense_reads
    bpp = 2 * (40 + 0.2 + 0.15 * 20) if compact else 2 * (40 + 20 + 8)
    resident = int(sum(n * (rank + 1) // world - n * rank // world for n in pairs) * bpp) + (n_pairs // world) * 2 * 20
    passes = args.passes or wgs.plan_passes(2 * n_pairs, READ_LEN, k, resident + (total - free0), total, world=world,
                                            n_samples=len(covs), coverage_hint=covs[0], wide=k > 30)
    if world > 1:                                             # every rank must run the same number of passes
        t = torch.tensor([passes], dtype=torch.int64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        passes = int(t.item())
    if world > 1:
        # what this rank plans to hold, before anything big is allocated (wgs.plan_passes' own terms): a first run on real
        # hardware that dies of memory says on which rank and by how much
        windows = 2 * n_pairs * max(READ_LEN - k + 1, 0)
        share = passes * world
        rec = (1.6 if k > 30 else 2.2) * windows / share
        plan = {"rank": rank, "device": torch.cuda.current_device(), "hbm_total_GB": round(total / 1e9, 1),
                "hbm_free_GB": round(free0 / 1e9, 1), "passes": passes, "resident_reads_GB": round(resident / 1e9, 2),
                "records_per_pass_GB": round(rec / 1e9, 2), "receive_buffers_per_pass_GB": round(rec * (world - 1) / world / 1e9, 2),
                "planned_peak_GB": round((resident + 2 * rec * 1.125) / 1e9, 1)}
        print("bench.py plan: " + json.dumps(plan), file=sys.stderr, flush=True)
        if not args.one_device and plan["planned_peak_GB"] > 0.95 * free0 / 1e9:
            raise SystemExit(f"rank {rank}: the planned peak ({plan['planned_peak_GB']} GB) exceeds the free HBM of device "
                             f"{plan['device']} ({plan['hbm_free_GB']} GB): something else holds this device")
    t0 = time.perf_counter()
    # pairs per resident block: a block's k-mer windows are indexed with 32 bits (limits) -- 2^24 pairs of
    # 150 bp reads have 2^32 windows at k = 23 exactly (found by the self-check sweep, profiles/r06_selfcheck_sweep.txt)
    blk_pairs = 1 << 24
    while 2 * blk_pairs * max(READ_LEN - k + 1, 1) >= 1 << 32:
        blk_pairs >>= 1
    samples = [wgs.make_sample(ctx, sy, n * (rank + 1) // world - n * rank // world, blk_pairs, MIN_Q, want_good=(i == 0),
                               first_pair=n * rank // world, compact=compact) for i, (sy, n) in enumerate(zip(sys_, pairs))]
    ctx.sync()
    resident = sum(b.device_bytes for s_ in samples for b in s_)
    t_gen = time.perf_counter() - t0
    trio = wgs.WgsTrio(ctx, k, JF_SIZE, LOWER, MIN_COV, MAX_DEPTH, THRESH, passes=passes,
                       group=dist.group.WORLD if world > 1 else None)
    trio.masks_are_views = True     # (a step's hit masks lie in page-locked host memory; the next step overwrites them)
    if os.environ.get("RFX_BENCH_MAP_BUDGET"):     # (profiling runs without a warm-up step: the run-map pool from the start)
        trio.map_budget = int(float(os.environ["RFX_BENCH_MAP_BUDGET"]))
    step = lambda: trio.run(samples)  # noqa: E731
    reads = [2 * n for n in pairs]
    what = (f"tumor {covs[0]}x / normal {covs[1]}x pair (BASELINE configs[4])" if tn
            else f"{args.coverage}x WGS trio (BASELINE configs[2])")
    desc = (f"synthetic {what}: genome {G} bp, {'/'.join(map(str, reads))} x {READ_LEN} bp reads per sample, on each of "
            f"{world} GPU(s) {len(samples[0])} resident blocks of the subject ({resident / 1e9:.0f} GB of packed reads in "
            f"HBM, generated on the device in {t_gen:.1f} s), {n_snv} SNVs, seed {SEED}, k={k}, -s 8G -L {LOWER}, MinCov "
            f"{MIN_COV}, MaxHashDepth {MAX_DEPTH}, MinQ {MIN_Q}, thresh {THRESH}; {passes} minimizer-shard pass(es) per step "
            f"(per-shard (pos,key)-sorted records, struck out shard by shard: no cross-shard payload of a sample is assembled -- "
            f"the drop-in `jellyfish count` does that, end_to_end)")
    args.k = k
    args.n_samples = len(covs)
    return (step, sum(reads), sum(reads) // len(reads) // world, reads[0], desc, "strong",
            {"passes": passes, "hbm_total": total, "hbm_free_at_start": free0, "resident_read_bytes": resident,
             "read_blocks": "compact" if compact else "dense", "_trio": trio, "_samples": samples, "_sys": sys_,
             "_pairs": pairs})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)   # (the second one runs with the blocks cut ahead: the arena grows there)
    ap.add_argument("--workload", choices=("wgs", "tn", "s1"), default="wgs",
                    help="wgs: 30x trio, k=25 (configs[2]); tn: tumor 60x / normal 30x, k=31 (configs[4]); s1: configs[1]")
    ap.add_argument("--k", type=int, default=0, help="k-mer length (default 25; 31 for tn)")
    ap.add_argument("--genome", type=int, default=3_100_000_000, help="wgs: genome length (reads scale with it)")
    ap.add_argument("--coverage", type=int, default=30)
    ap.add_argument("--passes", type=int, default=0, help="wgs: minimizer-shard passes (0 = plan from free HBM)")
    ap.add_argument("--dense-reads", action="store_true",
                    help="wgs: keep the read blocks in the dense form (68 B per 150 bp read instead of 43)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="(internal) print the cpu_baseline object and exit")
    ap.add_argument("--end-to-end-only", action="store_true", help="(internal) print the end_to_end object and exit")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-check", action="store_true",
                    help="skip the self-check after the timed region (rufus_amd.wgs.self_check: three more steps)")
    ap.add_argument("--e2e-pairs", type=int, default=32_000_000,
                    help="read pairs per sample of the end-to-end leg (refused when the text would not fit the container's memory)")
    ap.add_argument("--inner", action="store_true", help="(internal) the GPU part only, run by the launcher below")
    ap.add_argument("--one-device", action="store_true",
                    help="dry run of --gpus N on a one-GPU box: all ranks share device 0 (RFX_BENCH_BACKEND=gloo if RCCL refuses); "
                         "the line is marked, its value is not a scaling measurement")
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline()))
        return
    if args.end_to_end_only:
        print(json.dumps(end_to_end(args.e2e_pairs)))
        return

    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.inner:
        # One GPU: this process only launches.  The timed part runs in a child (--inner) that exits -- and with it
        # its HIP context -- before the CPU baseline and the end-to-end leg start their own processes: an idle
        # process that still holds a context on the GPU slows the drop-in tools' ingest by 0.5 s per 64 M reads
        # (measured: scratch/parent_effect.sh).
        p = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ["--inner"], stdout=subprocess.PIPE)
        out = p.stdout.decode().strip().splitlines()
        if p.returncode != 0 or not out:
            sys.stdout.write(p.stdout.decode())
            raise SystemExit(p.returncode or 1)
        for ln in out[:-1]:
            print(ln)
        line = json.loads(out[-1])
        legs(args, line)
        print(json.dumps(line))
        return

    import torch
    import torch.distributed as dist
    from rufus_amd import capi

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.one_device:     # dry run of the N-rank path on a one-GPU box: every rank on device 0
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        backend = os.environ.get("RFX_BENCH_BACKEND", "nccl")    # (gloo: when RCCL refuses two ranks on one device)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=None if args.one_device else torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        # The first real N-rank run must diagnose itself: the group the collectives will run over has
        # exactly --gpus ranks, every rank sits on a device of its own, and one all-reduce really crosses all of them.
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"rank {rank}: process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")
        probe = torch.tensor([1, local], dtype=torch.int64, device="cuda")
        gathered = [torch.zeros_like(probe) for _ in range(world)]
        dist.all_gather(gathered, probe)
        seen = [int(g[1].item()) for g in gathered]
        if sum(int(g[0].item()) for g in gathered) != args.gpus:
            raise SystemExit(f"rank {rank}: all_gather over {dist.get_backend()} answered for {len(seen)} ranks, --gpus {args.gpus}")
        if not args.one_device and len(set(seen)) != world:
            raise SystemExit(f"rank {rank}: ranks share devices {seen}: one process per GPU expected (LOCAL_RANK)")

    ctx = capi.Context(local)   # raises without a gfx950 GPU: no CPU fallback
    step, reads_per_step, reads_per_launch, reads_filtered, desc, scaling, extra = (
        run_wgs if args.workload != "s1" else run_s1)(args, ctx, rank, world, dist, torch)
    k_used = getattr(args, "k", 0) or K
    n_samples = getattr(args, "n_samples", 3)

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # HIP-event brackets (rfx_prof_*) on the library's stream around every launch of the count chain and the
    # filter.  A bracket costs ~10-30 us of host time: nothing against the multi-second WGS step, so there they
    # stay on for the whole timed region; on the 4 ms S1 step only during the last timed step.
    ctx.prof_filter(K2_CHAIN + ("k_filter",))
    ctx.prof(True)
    for w_ in range(args.warmup):
        res = step()
        # After the first (untimed) step the peak of a step is known: what the device has left beside it (up to 93 % of its
        # memory) may hold RUN MAPS -- 32 bytes per read that the first shard pass over a block leaves so that the later
        # passes rebuild their records from reads + map instead of hashing the block again (rfx_runmaps_*,
        # rufus_amd/wgs.py _count_shard_local).  Same results (the self-check compares the record multisets with a run of
        # S + 1 plain passes).  RFX_BENCH_EARLY=1: round 4's scheme instead (blocks cut ahead for the next pass, 132 bytes
        # per read: rfx_count_set_early); RFX_BENCH_NO_EARLY=1: neither.
        if w_ == 0 and "_trio" in extra and world == 1 and extra.get("passes", 1) > 1 and not os.environ.get("RFX_BENCH_NO_EARLY"):
            # (should a step not fit after all, WgsTrio.run drops the maps and repeats it -- same passes)
            frac = float(os.environ.get("RFX_BENCH_HEADROOM", "0.86" if os.environ.get("RFX_BENCH_EARLY") else "0.93"))
            head = int(frac * min(extra["hbm_total"], extra["hbm_free_at_start"])) - int(ctx.mem_stats()["peak"])
            if head > (2 << 30):
                if os.environ.get("RFX_BENCH_EARLY"):
                    extra["_trio"].early_budget = head
                    extra["early_cut_budget_bytes"] = head
                else:
                    # the pool holds two samples' maps (WgsTrio.run orders the passes so that no more are alive)
                    sizes = sorted((sum(((b.n * 32 + 255) & ~255) + ((b.n // 32 + 4097) * 4 + 255 & ~255) for b in s_)
                                    for s_ in extra["_samples"]), reverse=True)
                    need = sum(sizes[:2]) + (1 << 20)
                    # When the maps do not fit beside S passes, S + 1 passes with every block mapped beat S passes with some
                    # blocks hashed S times (configs[4]: 3 passes 576 M reads/s, 4 passes 608 M): one more pass if ITS
                    # transients (estimated: they shrink like S / (S + 1)) leave the room.
                    S = extra["passes"]
                    if head < 0.75 * need and not args.passes and S < 16:      # (a few blocks without a map: not worth a pass)
                        st = ctx.mem_stats()
                        peak1 = st["used"] + (st["peak"] - st["used"]) * S // (S + 1)
                        head1 = int(frac * min(extra["hbm_total"], extra["hbm_free_at_start"])) - int(peak1)
                        if head1 >= 0.85 * need:
                            extra["_trio"].passes = S + 1
                            extra["passes"] = S + 1
                            extra["passes_note"] = f"{S} passes fit the records; {S + 1} leave room for every block's run map"
                            desc = desc.replace(f"; {S} minimizer-shard pass(es) per step", f"; {S + 1} minimizer-shard pass(es) per step")
                            head = head1
                    pool = min(head, need)
                    extra["_trio"].map_budget = pool
                    extra["run_map_budget_bytes"] = pool
    live_all = args.workload != "s1"
    ctx.prof(live_all)
    ctx.prof_reset()
    # No cyclic-GC pass of the interpreter inside the timed region: the synthetic reads are millions of Python objects, a
    # full collection over them takes 40-80 ms, and where it lands depends on the allocation count so far (on a box's
    # first run, which compiles the .pyc files, it fell outside the 20 timed S1 steps; on every later run inside the
    # last one: 3.6 -> 5.5 ms per step).  Reference counting still frees every step's tables as before.
    gc.collect()
    gc.disable()
    fence()
    t0 = time.perf_counter()
    step_ms = []
    for i in range(args.steps):
        if not live_all and i == args.steps - 1:
            ctx.prof(True)
        t_s = time.perf_counter()
        res = step()
        step_ms.append(round((time.perf_counter() - t_s) * 1e3, 3))
    fence()
    dt = time.perf_counter() - t0
    gc.enable()
    if os.environ.get("RFX_BENCH_STEP_TIMES"):
        print("bench.py step wall ms: " + json.dumps(step_ms), file=sys.stderr, flush=True)
    prof = ctx.prof_dict()
    prof_steps = args.steps if live_all else 1
    ctx.prof(False)
    # After the timed region: the run proves itself (size-independent properties -- at 3.1 Gb nothing else can).
    checks = None
    if args.workload != "s1" and not args.no_check:
        from rufus_amd import wgs
        t_chk = time.perf_counter()
        checks = wgs.self_check(ctx, extra["_trio"], extra["_samples"], extra["_sys"], res, extra["_pairs"][0], MIN_Q)
        checks["seconds"] = round(time.perf_counter() - t_chk, 1)
    trio_ = extra.get("_trio")
    extra = {k_: v for k_, v in extra.items() if not k_.startswith("_")}
    multi = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # what the first real SCALE run is checked by at a glance: every rank's view of the group and what it exchanged
        mine = torch.tensor([rank, local, dist.get_world_size(), int(getattr(trio_, "exchange_sent", 0)),
                             int(getattr(trio_, "exchange_received", 0))], dtype=torch.int64, device="cuda")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        multi = {"backend": dist.get_backend(), "rccl_world_size": dist.get_world_size(),
                 "per_rank": [{"rank": int(x[0]), "device": int(x[1]), "rccl_world_size": int(x[2]),
                               "record_bytes_sent_per_step": int(x[3]), "record_bytes_received_per_step": int(x[4])}
                              for x in (y.tolist() for y in allr)]}

    if rank == 0:
        # K2+K3 (count -> sorted records) is the dominant stage: a chain of launches per sample (super-k-mer
        # partition of every read block, refinement of the partition, LDS count of every minimizer bin,
        # partition + sort of the survivors).  The algorithmic bytes of SURVEY 8(d) K2 cover the stage as a
        # whole, so one "launch" = the chain of one sample, its duration = the SUM of its kernels' durations.
        k2 = [n for n in K2_CHAIN if n in prof and prof[n][1]]
        n_chains = n_samples * prof_steps
        parts = {n: prof[n][0] / n_chains for n in k2}
        chain_ms = sum(parts.values())
        bytes_per_launch = algorithmic_bytes_per_read(k=k_used) * reads_per_launch
        achieved = bytes_per_launch / (chain_ms * 1e-3) / 1e9 if chain_ms else 0.0
        f_ms = prof["k_filter"][0] / prof_steps if "k_filter" in prof and prof["k_filter"][1] else 0.0
        line = {
            "metric": f"reads/sec through k-mer count+filter at k={k_used}",
            "value": reads_per_step * args.steps / dt,
            "unit": "reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": desc, "reads_counted_per_step": reads_per_step,
                       "reads_filtered_per_step": reads_filtered, "parallelism": f"read-block shard x{world}",
                       "mutant_kmers": int(res["n_mutant"]), "pulled_pairs": int(res["n_pulled"]),
                       "records_per_sample": [int(x) for x in res["n_records"]], **extra,
                       **({"blocks_replayed_from_run_maps_per_step": int(trio_.replayed_blocks),
                           "run_maps_hashed_ahead_on_the_second_stream_per_step": int(getattr(trio_, "maps_ahead", 0)),
                           "count_wall_ms_per_sample": round(getattr(trio_, "count_wall_s", 0.0) * 1e3 / max(n_samples, 1), 1)}
                          if trio_ is not None else {}),
                       "checked": checks is not None, "checks": checks,
                       **({"multi_gpu": multi} if multi else {}),
                       **({"one_device_dry_run": f"{world} ranks share device 0 over {dist.get_backend()}: the N-rank path is "
                                                 "exercised, the value is NOT a scaling measurement"} if args.one_device and world > 1 else {}),
                       "hbm_peak_bytes": ctx.mem_stats()["peak"], "hbm_mapped_bytes": ctx.mem_stats()["mapped"]},
            "roofline": {"bound": "hbm", "kernel": "count chain of one sample: " + "+".join(k2), "achieved": achieved,
                         "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": None,
                         "avg_launch_ms": chain_ms, "avg_launch_ms_by_kernel": {n: round(v, 3) for n, v in parts.items()},
                         "launches": n_chains, "reads_per_launch": reads_per_launch,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "launches_by_kernel_per_chain": {n: round(prof[n][1] / n_chains, 1) for n in k2}},
            # K5 (read filter) against the same roofline: SURVEY 8(d) prices it at 61 B/read of streaming
            "roofline_filter": {"bound": "hbm", "kernel": "k_filter", "achieved": 61.0 * reads_filtered / world / (f_ms * 1e-3) / 1e9,
                                "peak": 8000.0, "unit": "GB/s",
                                "frac": 61.0 * reads_filtered / world / (f_ms * 1e-3) / 1e9 / 8000.0,
                                "ms_per_step": f_ms, "algorithmic_bytes_per_step": 61 * reads_filtered // world,
                                "note": "rank 0's share"} if f_ms else None,
        }
        # HBM traffic of the chain: PMC counters cannot be collected inside this run (rocprofv3 wraps the process), so the
        # line quotes the newest profiles/rNN_pmc_<workload>.json -- but only if it was taken on THIS build: the file
        # carries the fingerprint of the kernel sources it was measured with (kernel_source_fingerprint(); the snapshot a
        # GPU box gets has no .git to ask), and a stale one is refused (traffic stays null, the reason is in the line).
        import glob
        pmc_files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_pmc_{args.workload}.json")))
        pmc_path = pmc_files[-1] if pmc_files else ""          # the newest round's counters
        if pmc_path:
            pmc = json.load(open(pmc_path))
            fp = kernel_source_fingerprint()
            if pmc.get("kernel_sources_sha16") != fp:
                line["roofline"]["traffic_source"] = (f"{os.path.basename(pmc_path)} refused: taken on kernel sources "
                                                      f"{pmc.get('kernel_sources_sha16')}, this build is {fp}")
            elif pmc.get("genome") in (None, getattr(args, "genome", None)) and "_chain" in pmc:
                line["roofline"]["traffic"] = pmc["_chain"]["hbm_bytes_per_sample"]
                line["roofline"]["traffic_source"] = (os.path.basename(pmc_path) + f" (commit {pmc.get('commit')}, kernel sources "
                                                      f"{fp}): " + pmc["_chain"].get("source", ""))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    if rank == 0:
        print(json.dumps(line))


def legs(args, line):
    """The two host-side legs of the one-GPU line, each in a child process: a report is never a reason to lose the
    measurement (a crash of the CPU code, an OpenMP runtime clash, a timeout -- the line is printed regardless)."""
    if not args.no_cpu_baseline:
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"], stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE, timeout=420)
            line["cpu_baseline"] = json.loads(p.stdout.decode().strip().splitlines()[-1])
        except Exception as e:
            err = locals().get("p")
            line["cpu_baseline"] = {"value": None, "unit": "reads/s", "cores": 1, "kind": "port",
                                    "sample": f"failed: {e!r} " + (err.stderr.decode()[-300:] if err is not None else "")}
    if not args.no_end_to_end and not args.no_cpu_baseline:
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--end-to-end-only"], stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE, timeout=600)
            line["end_to_end"] = json.loads(p.stdout.decode().strip().splitlines()[-1])
            line["overlap_wall_s"] = line["end_to_end"].get("overlap_wall_s")
        except Exception as e:
            err = locals().get("p")
            line["end_to_end"] = {"value": None, "error": f"{e!r} " + (err.stderr.decode()[-300:] if err is not None else "")}


if __name__ == "__main__":
    main()
