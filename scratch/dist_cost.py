"""Cost of the multi-GPU exchange path's compute on ONE GPU: partial count (lower = 1) and the owner-side
reduce of partials from `world` simulated ranks (the same partials fed world times, pos slice 1/world)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rufus_amd import capi
from rufus_amd.dist import HipBackend, owner_bounds
from tests.synth import make_trio, flat_reads

trio = make_trio(genome_len=5_000_000, n_pairs=500_000, n_snv=20, seed=12345, read_seed=1000)
ctx = capi.Context(0)
seq, qual, off = flat_reads(trio["child"])
blk = ctx.upload(capi.PackedReads(seq, off, qual, bench.MIN_Q, capi.PACK_COUNT | capi.PACK_FILTER))
be = HipBackend(ctx, bench.K, bench.JF_SIZE, 1 << 26)
for world in (2, 8):
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        keys, counts, pos = be.count_partials(blk)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        b = owner_bounds(be.lsize, world)
        cuts = torch.searchsorted(pos, torch.tensor(b, dtype=torch.int64, device=pos.device))
        lo, hi = int(cuts[0]), int(cuts[1])
        rk = keys[lo:hi].repeat(world); rc = counts[lo:hi].repeat(world)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        rec, histo = be.reduce_partials(rk, rc, bench.LOWER, b[0], b[1])
        torch.cuda.synchronize(); t3 = time.perf_counter()
        n = len(rec); rec.free()
    print(f"world {world}: partials {len(keys)} in {1e3*(t1-t0):.2f} ms; reduce of {len(rk)} pairs -> {n} records in {1e3*(t3-t2):.2f} ms")

# ---- minimizer sharding: partition, then the owner's count of `world` runs (the same block's bins 1/world)
from rufus_amd.dist import bin_owner_bounds
for world in (2, 8):
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rec, bs, keep, ext = be.partition(blk)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        bins = bs.numel() - 1
        b = bin_owner_bounds(bins, world)
        bsh = bs.cpu()
        lo, hi = int(bsh[b[0]]), int(bsh[b[1]])
        full = torch.zeros(bins + 1, dtype=torch.int64)
        full[b[0]:b[1] + 1] = bsh[b[0]:b[1] + 1] - bsh[b[0]]
        full[b[1] + 1:] = hi - lo
        runs = [(rec[lo:hi].clone(), full.to(rec.device), ext[lo:hi].clone()) for _ in range(world)]
        keep.free()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        out, histo = be.count_records(runs, bench.LOWER)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        n = len(out); out.free()
    print(f"world {world}: partition {rec.numel()} records in {1e3*(t1-t0):.2f} ms ({rec.numel()*8/1e6:.0f} MB to exchange); "
          f"owner count of {world} x {hi-lo} records -> {n} records in {1e3*(t3-t2):.2f} ms")
