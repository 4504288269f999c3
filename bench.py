#!/usr/bin/env python3
"""Headline benchmark: reads/s through the k-mer count + set-difference + read-filter hot path at k=25
(BASELINE.json metric), with the achieved HBM GB/s of the dominant kernel against the roofline.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (N=1): BASELINE.json configs[1] -- synthetic 1 M x 150 bp trio (0.5 M pairs per sample, 5 Mb
genome = 30x, 20 planted SNVs, seed 12345), k=25, jellyfish -s 8G / -L 2, MinCov 5, MaxHashDepth 1200,
MinQ 15, HashCountThreshold 1.  One step = the whole trio through the path, inputs already packed and
resident in HBM:  for each of the 3 samples count (K2) -> sorted records + histogram (K3);  mutant
hash list = subject minus controls (K4);  filter of the subject's 1 M reads (K5).
value = reads that went through the count stage (3 M per step) / wall time.

N>1 (weak scaling): every rank holds its own 1 M-read block of each sample (same genome, different
reads).  Per sample the ranks count locally, exchange (key,count) partials by pos-range owner with
an RCCL all-to-all, reduce at the owner, all-reduce the count-of-counts histogram; the owner slices
of the mutant set are all-gathered and every rank filters its own subject block.
"""
import argparse
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
READ_LEN = 150


def algorithmic_bytes_per_read(L=READ_LEN, k=K):
    """ceil(L/4) code bytes + ceil(L/8) mask bytes + (L-k+1) windows x
    (8 B key read + 4 B count read + 4 B count write)."""
    return (L + 3) // 4 + (L + 7) // 8 + (L - k + 1) * 16


def cpu_baseline(n_pairs, genome_len):
    """Same path on the host CPU, one thread, on a scaled-down trio of the same shape (30x coverage,
    same read length / error model): oracle C++ port for count + set difference, the REAL reference
    binary oracle/_ref/RUFUS.Filter (built from /root/reference/src in the build container) for the filter."""
    import oracle
    from tests.synth import make_trio, fastq_bytes
    trio = make_trio(genome_len=genome_len, n_pairs=n_pairs, n_snv=8, seed=4242)
    fq = {n: [fastq_bytes(trio[n], m) for m in (1, 2)] for n in ("child", "mother", "father")}
    t0 = time.perf_counter()
    recs = {n: oracle.count(fq[n], K, JF_SIZE, lower=LOWER) for n in fq}
    t_count = time.perf_counter() - t0
    t0 = time.perf_counter()
    hl = oracle.hash_list(recs["child"], [recs["mother"], recs["father"]], MIN_COV, MAX_DEPTH)
    t_merge = time.perf_counter() - t0
    kind = "port"
    d = tempfile.mkdtemp(prefix="rfx_cpu_")
    exe = os.path.join(ROOT, "oracle", "_ref", "RUFUS.Filter")
    if os.path.exists(exe):
        for m in (1, 2):
            open(f"{d}/m{m}.fq", "wb").write(fq["child"][m - 1])
        open(f"{d}/hl", "w").write(hl)
        t0 = time.perf_counter()
        subprocess.run([exe, f"{d}/hl", f"{d}/m1.fq", f"{d}/m2.fq", f"{d}/o", str(K), str(MIN_Q), str(THRESH), "1"],
                       stdout=subprocess.DEVNULL, check=True)
        t_filter = time.perf_counter() - t0
        filt = "reference binary oracle/_ref/RUFUS.Filter (-O2)"
    else:
        fs = oracle.FilterSet(hl.encode())
        t0 = time.perf_counter()
        fs.pairs(fq["child"][0], fq["child"][1], K, MIN_Q, THRESH)
        t_filter = time.perf_counter() - t0
        filt = "oracle port of RUFUS.Filter"
    reads = 3 * 2 * n_pairs
    total = t_count + t_merge + t_filter
    return {"value": reads / total, "unit": "reads/s", "cores": 1, "kind": kind,
            "sample": f"trio of 3 x {2 * n_pairs} reads x {READ_LEN} bp on a {genome_len} bp genome (30x), k={K}: "
                      f"count {t_count:.2f}s (oracle C++ port, sort-based) + set difference {t_merge:.2f}s (oracle) + "
                      f"filter of the subject {t_filter:.2f}s ({filt}), 1 thread each"}


def revcomp_keys(keys, k):
    keys = np.asarray(keys, dtype=np.uint64)
    r = np.zeros_like(keys)
    x = keys.copy()
    for _ in range(k):
        r = (r << np.uint64(2)) | (np.uint64(3) - (x & np.uint64(3)))
        x >>= np.uint64(2)
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cpu-pairs", type=int, default=200_000, help="pairs per sample of the CPU baseline sample")
    ap.add_argument("--pairs", type=int, default=500_000, help="read pairs per sample per GPU")
    ap.add_argument("--genome", type=int, default=5_000_000)
    ap.add_argument("--capacity", type=int, default=1 << 26, help="initial count-table slots")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from rufus_amd import capi
    from rufus_amd.dist import TrioShard
    from tests.synth import make_trio, flat_reads

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    ctx = capi.Context(local)   # raises without a gfx950 GPU: no CPU fallback
    # same genome + SNVs on every rank (seed), different reads per rank (read_seed)
    trio = make_trio(genome_len=args.genome, n_pairs=args.pairs, n_snv=20, seed=12345, read_seed=1000 + rank)
    blocks = {}
    for name in ("child", "mother", "father"):
        seq, qual, off = flat_reads(trio[name])
        blocks[name] = ctx.upload(capi.PackedReads(seq, off, qual, MIN_Q, capi.PACK_COUNT | capi.PACK_FILTER))
    n_reads = blocks["child"].n
    del trio

    shard = TrioShard(ctx, K, JF_SIZE, LOWER, MIN_COV, MAX_DEPTH, THRESH, capacity=args.capacity,
                      group=dist.group.WORLD if world > 1 else None)

    def step():
        return shard.run(blocks["child"], [blocks["mother"], blocks["father"]])

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # every launch of the count -> sorted-records stage, whichever path rfx_count_add/finish took
    K2_CHAIN = ("k_msp_part1", "k_msp_count", "k_msp_leaf", "k_surv_hist", "k_surv_part2", "k_surv_sort",
                "k_bin_count", "k_bin_offsets", "k_bin_scatter", "k_part1", "k_part2", "k_leaf", "k_leaf_compact",
                "k_count_reads")
    ctx.prof(True)          # warm the profiling path too (event pool)
    for _ in range(args.warmup):
        res = step()
    # Timed region.  The HIP-event brackets that give roofline.achieved cost ~30 us of pipeline
    # bubble each (measured: 8.5 ms/step with all ~100 launches bracketed, 7.6 ms with none), so they
    # are live on the launches of the dominant stage during the LAST timed step only.
    ctx.prof_filter(K2_CHAIN + ("k_filter",))
    ctx.prof(False)
    ctx.prof_reset()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i == args.steps - 1:
            ctx.prof(True)
        res = step()
    fence()
    dt = time.perf_counter() - t0
    prof = ctx.prof_dict()
    # One extra, untimed step with every launch bracketed, for the per-kernel breakdown.
    ctx.prof(True)
    ctx.prof_filter(())
    ctx.prof_reset()
    step()
    fence()
    prof_all = ctx.prof_dict()
    ctx.prof(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        reads_per_step = 3 * n_reads * world
        # K2+K3 (count -> sorted records) is the dominant stage.  It is a chain of launches per read
        # block (MSP path: super-k-mer partition in two levels, LDS count of every minimizer bin,
        # partition + sort of the survivors); the algorithmic bytes of SURVEY 8(d) K2 cover the stage as
        # a whole, so the time used for roofline.achieved is the SUM of their average durations.
        k2 = [n for n in K2_CHAIN if n in prof]
        n_count = max([prof[n][1] for n in k2] or [0])
        parts = {n: prof[n][0] / max(prof[n][1], 1) for n in k2}
        avg_ms = sum(parts.values())
        bytes_per_launch = algorithmic_bytes_per_read() * n_reads
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms else 0.0
        line = {
            "metric": "reads/sec through k-mer count+filter at k=25",
            "value": reads_per_step * args.steps / dt,
            "unit": "reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"synthetic trio, {n_reads} x {READ_LEN} bp reads per sample per GPU "
                                   f"(genome {args.genome} bp, 20 SNVs, seed 12345), k={K}, -s 8G -L {LOWER}, "
                                   f"MinCov {MIN_COV}, MaxHashDepth {MAX_DEPTH}, MinQ {MIN_Q}, thresh {THRESH}",
                       "reads_counted_per_step": reads_per_step, "reads_filtered_per_step": n_reads * world,
                       "parallelism": f"read-block shard x{world}" + (
                           f", all-to-all of super-k-mer records by {shard.shard_by}-bin owner" if world > 1 else ""),
                       "mutant_kmers": int(res["n_mutant"]), "pulled_pairs": int(res["n_pulled"]),
                       "records_subject": int(res["n_records"][0])},
            "roofline": {"bound": "hbm", "kernel": "+".join(k2), "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": None,
                         "avg_launch_ms": avg_ms, "avg_launch_ms_by_kernel": {n: round(v, 4) for n, v in parts.items()},
                         "launches": int(n_count), "reads_per_launch": n_reads,
                         "algorithmic_bytes_per_launch": bytes_per_launch},
            # K5 (read filter) against the same roofline: SURVEY 8(d) prices it at 61 B/read of streaming
            "roofline_filter": (lambda ms: {"bound": "hbm", "kernel": "k_filter", "achieved": 61.0 * n_reads / (ms * 1e-3) / 1e9,
                                            "peak": 8000.0, "unit": "GB/s", "frac": 61.0 * n_reads / (ms * 1e-3) / 1e9 / 8000.0,
                                            "avg_launch_ms": ms, "algorithmic_bytes_per_launch": 61 * n_reads})(
                prof["k_filter"][0] / max(prof["k_filter"][1], 1)) if "k_filter" in prof and prof["k_filter"][0] > 0 else None,
            "kernels_ms_per_step": {k: round(v[0], 4) for k, v in sorted(prof_all.items())},
            "kernels_ms_per_step_source": "one extra untimed step with every launch bracketed",
        }
        # HBM bytes per launch from the committed rocprofv3 --pmc passes of this same command
        # (profiles/summarize_pmc.py; 2*FETCH_SIZE + WRITE_SIZE, KB -> bytes), default workload only.
        pmc_path = os.path.join(ROOT, "profiles", "r01_pmc.json")
        if os.path.exists(pmc_path) and args.pairs == 500_000 and args.genome == 5_000_000:
            pmc = json.load(open(pmc_path))
            if "_chain" in pmc and "k_msp_part1" in k2 and "k_msp_part1" in pmc:
                line["roofline"]["traffic"] = pmc["_chain"]["hbm_bytes_per_sample"]
                line["roofline"]["traffic_source"] = ("profiles/r01_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                                                      "all launches of the chain, per sample)")
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(args.cpu_pairs, args.cpu_pairs * 10)
            except Exception as e:   # the baseline is a report, never a reason to lose the measurement
                line["cpu_baseline"] = {"value": None, "unit": "reads/s", "cores": 1, "kind": "port",
                                        "sample": f"failed: {e!r}"}
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
