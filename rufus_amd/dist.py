"""Read-block sharding of the trio hot path over the GPUs of one node.

One process per GPU, ``torch.distributed`` for the exchange (backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests).  Per sample:

1. every rank counts ITS read block into a local table and drains all (key,count) partials in
   (pos,key) order (K2 + K3 with lower = 1);
2. the pos range is cut into ``world`` equal owner slices; because the partials are pos-sorted each
   destination's share is one contiguous run -> one ``all_to_all_single`` with split sizes (keys,
   counts);
3. the owner adds the partials it received into a fresh table (``rfx_count_add_pairs_dev``) and
   finishes it with the real lower bound: its records are a contiguous slice of the single-GPU
   ``.Jhash`` payload, so concatenating the slices in rank order reproduces that file;
4. count-of-counts histograms are additive across owners (keys are disjoint) -> ``all_reduce(SUM)``.

The set difference needs no communication (every rank owns the same pos slice of every sample);
the per-slice mutant k-mers are all-gathered (small) and every rank filters its own subject block.

**Minimizer sharding** (default where the MSP count path applies, 23 <= k <= 25).  The scheme above
ships (key,count) partials -- 12 B per distinct k-mer per rank, and the owner has to reduce them.  The
MSP path offers a cheaper cut: a read block becomes 8-byte super-k-mer records grouped by minimizer
bin, and every instance of a canonical k-mer is in the same bin on every rank.  So

1. every rank partitions ITS read block into records (``rfx_count_add``; no counting yet);
2. bins are dealt to owners in equal contiguous ranges (of 256 virtual top-level bins, so ranks with
   different bin counts agree); a destination's share is one contiguous run of the record array ->
   one ``all_to_all_single`` for the records (2.4 B per k-mer instance) and one for the bin offsets;
3. the owner imports the runs (``rfx_count_add_records_dev``) and finishes: complete counts of the
   k-mers of its bins, sorted by (pos,key).  No reduce, no partial counts.
4. The shard is a function of the k-mer, the same for every sample: histograms add up
   (``all_reduce``), the set difference is local to the shard, mutant k-mers are all-gathered and put
   back into (pos,key) order.  (A rank's records are NOT a slice of the ``.Jhash`` payload in this mode;
   ``merge_shards`` interleaves them when the file is wanted.)

The compute is delegated to a *backend* object; the product backend is :class:`HipBackend` (C-ABI ->
HIP kernels).  The tests drive the same exchange logic on CPU with gloo and a checker backend.
"""
from __future__ import annotations

import numpy as np
import os

import torch
import torch.distributed as dist

from . import capi

HISTO_BINS = capi.HISTO_BINS


def revcomp_keys(keys: np.ndarray, k: int) -> np.ndarray:
    """Reverse complement of 2-bit packed k-mers: complement every base, reverse the 32 base slots of
    the word (pairs inside nibbles, nibbles inside bytes, then the bytes), shift the k bases down."""
    x = ~np.asarray(keys, dtype=np.uint64)
    m2, m4 = np.uint64(0x3333333333333333), np.uint64(0x0F0F0F0F0F0F0F0F)
    x = ((x >> np.uint64(2)) & m2) | ((x & m2) << np.uint64(2))
    x = ((x >> np.uint64(4)) & m4) | ((x & m4) << np.uint64(4))
    return x.byteswap() >> np.uint64(64 - 2 * k)


class _ShardResult(dict):
    """Result of TrioShard.run; the per-pair boolean vector is unpacked from the hit mask on first use."""

    def __missing__(self, key):
        if key != "pulled":
            raise KeyError(key)
        v = self["_pulled_fn"]()
        self[key] = v
        return v


class HipBackend:
    """Everything a shard computes, through the C-ABI.  Tensors handed to / taken from
    torch.distributed live in torch-owned HBM; data moves between them and library-owned buffers
    with device-to-device copies on the library stream."""

    def __init__(self, ctx: capi.Context, k: int, size: int, capacity: int = 0, device: torch.device | None = None):
        self.ctx, self.k, self.size, self.capacity = ctx, k, size, capacity
        self.lsize = capi.ceil_log2(size)
        self.cols = capi.jf_matrix(self.lsize, k)
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())

    # -- single-rank path -------------------------------------------------------------------------
    def local_count(self, block, lower: int, shard=None):
        """shard = (s, S): only the k-mers of minimizer shard s of S (a pass over a sample too big for HBM)."""
        t = capi.CountTable(self.ctx, self.k, self.size, True, self.capacity)
        try:
            if shard is not None:
                t.set_shard(*shard)
            t.add(block)
            return t.finish(lower, want_histo=True)
        finally:
            t.free()

    def count_begin(self, block, lower: int):
        """Queue count + finish of a block without waiting (MSP path); pair with count_end()."""
        t = capi.CountTable(self.ctx, self.k, self.size, True, self.capacity)
        try:
            t.add(block)
            return t, t.finish_begin(lower, want_histo=True)
        except Exception:
            t.free()
            raise

    def count_end(self, pending):
        t, handle = pending
        try:
            return t.finish_end(handle)
        finally:
            t.free()

    # -- exchange path ----------------------------------------------------------------------------
    def count_partials(self, block):
        """(keys int64, counts int32, pos int64) device tensors of every distinct k-mer of the block."""
        rec, _ = self.local_count(block, 1)
        n = len(rec)
        keys = torch.empty(n, dtype=torch.int64, device=self.device)
        counts = torch.empty(n, dtype=torch.int32, device=self.device)
        pos = torch.empty(n, dtype=torch.int64, device=self.device)
        dk, dc, dp = rec.dev_ptrs()
        torch.cuda.synchronize(self.device)
        self.ctx.memcpy_dev(keys.data_ptr(), dk, n * 8)
        self.ctx.memcpy_dev(counts.data_ptr(), dc, n * 4)
        self.ctx.memcpy_dev(pos.data_ptr(), dp, n * 8)
        rec.free()
        return keys, counts, pos

    def reduce_partials(self, keys: torch.Tensor, counts: torch.Tensor, lower: int, pos_lo: int, pos_hi: int):
        """Owner-side reduce of received partials -> (records slice, histogram)."""
        t = capi.CountTable(self.ctx, self.k, self.size, True, self.capacity, pos_lo, pos_hi)
        try:
            torch.cuda.synchronize(self.device)
            t.add_pairs_dev(keys.data_ptr(), counts.data_ptr(), keys.numel())
            self.ctx.sync()
            return t.finish(lower, want_histo=True)
        finally:
            t.free()

    # -- minimizer-shard path ----------------------------------------------------------------------
    def msp_capable(self) -> bool:
        return 23 <= self.k <= 31     # record export / import between ranks: a 64-bit word + a 32-bit plane per record

    def partition(self, block):
        """(records int64[n], bin_start int64[bins+1], keep, planes int32[n]): the block's super-k-mer records (64-bit
        word + 32-bit plane each) grouped by minimizer bin, as zero-copy views of the library's device memory; `keep`
        owns that memory -- call keep.free() once the tensors are no longer needed (after the exchange)."""
        t = capi.CountTable(self.ctx, self.k, self.size, True, self.capacity, mode=capi.COUNT_MSP)
        try:
            t.add(block)
            segs = t.segments()
        except Exception:
            t.free()
            raise
        if not segs:                         # no k-mer at all
            t.free()
            return (torch.empty(0, dtype=torch.int64, device=self.device),
                    torch.zeros(257, dtype=torch.int64, device=self.device), None,
                    torch.empty(0, dtype=torch.int32, device=self.device))
        d_rec, d_bs, bins, n = segs[0]       # segments() has synchronised: the arrays are complete
        d_ext = t.segment_ext(0)
        rec = _device_view(d_rec, n, self.device) if n else torch.empty(0, dtype=torch.int64, device=self.device)
        ext = (_device_view32(d_ext, n, self.device) if n and d_ext
               else torch.empty(0, dtype=torch.int32, device=self.device))
        return rec, _device_view(d_bs, bins + 1, self.device), t, ext

    def count_records(self, runs, lower: int):
        """runs: [(records, bin_start, planes)] received from every rank for this owner's bins -> (records of
        the shard in (pos,key) order, histogram)."""
        t = capi.CountTable(self.ctx, self.k, self.size, True, self.capacity, mode=capi.COUNT_MSP)
        try:
            torch.cuda.synchronize(self.device)
            for rec, bs, ext in runs:
                t.add_records_dev(rec.data_ptr(), rec.numel(), bs.data_ptr(), bs.numel() - 1, ext.data_ptr() if rec.numel() else 0)
            self.ctx.sync()                      # the copies are done: the tensors may go
            return t.finish(lower, want_histo=True)
        finally:
            t.free()

    def pos_of(self, keys: np.ndarray) -> np.ndarray:
        return np.array([capi.jf_pos(self.cols, self.k, self.lsize, int(x)) for x in keys], dtype=np.uint64)

    def unique(self, subject, others, min_cov: int, max_cov: int):
        return capi.unique_to_subject(self.ctx, subject, others, min_cov, max_cov)

    def filter_pairs(self, canon_keys: np.ndarray, block, thresh: int):
        """Pair i = reads i and i + n/2 of the block (all of mate 1, then all of mate 2)."""
        keys = np.concatenate([canon_keys, revcomp_keys(canon_keys, self.k)]) if len(canon_keys) else canon_keys
        mset = capi.MutantSet(self.ctx, keys, self.k)
        try:
            _, mask, _ = mset.filter(block, thresh, last_base_skipped=True, want_hits=False)
        finally:
            mset.free()
        half = block.n // 2
        m8 = mask.view(np.uint8)

        def unpack():
            bits = np.unpackbits(m8, bitorder="little")[:block.n].astype(bool)
            return bits[:half] | bits[half:2 * half]

        if half % 8 == 0:   # the two mates' bits line up byte for byte: OR and popcount the packed mask
            n_pulled = int(np.bitwise_count(m8[:half // 8] | m8[half // 8:half // 4]).sum())
        else:
            n_pulled = int(unpack().sum())
        return n_pulled, unpack

    def free(self, rec):
        rec.free()

    def n_records(self, rec) -> int:
        return len(rec)


def owner_bounds(lsize: int, world: int):
    """pos range [b[g], b[g+1]) owned by rank g."""
    return [(g << lsize) // world for g in range(world + 1)]


class _DevMem:
    """n int64 at a raw device address, for torch.as_tensor (no copy)."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i8", "data": (ptr, False), "version": 3,
                                         "strides": None}


def _device_view(ptr: int, n: int, device) -> torch.Tensor:
    return torch.as_tensor(_DevMem(ptr, n), device=device)


class _DevMem32:
    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i4", "data": (ptr, False), "version": 2}


def _device_view32(ptr: int, n: int, device) -> torch.Tensor:
    return torch.as_tensor(_DevMem32(ptr, n), device=device)


def bin_owner_bounds(bins: int, world: int):
    """bin range [b[g], b[g+1]) owned by rank g.  Cut on 256 virtual top-level bins (the top 8 bits of
    the minimizer hash) so that ranks / samples partitioned into different bin counts agree."""
    assert bins >= 256 and bins % 256 == 0 and world <= 256
    return [-(-g * 256 // world) * (bins // 256) for g in range(world + 1)]


def _wire(t: torch.Tensor, group) -> torch.Tensor:
    """Tensor as the collective backend wants it: gloo moves host memory (used by the tests, which can
    put two ranks on one GPU), nccl/RCCL moves HBM directly over xGMI."""
    return t.cpu() if dist.get_backend(group) == "gloo" and t.is_cuda else t


def exchange_rows(out: torch.Tensor, inp: torch.Tensor, recv_l, send_l, group, max_bytes: int = 0):
    """all_to_all_single(out, inp, recv_l, send_l) for payloads that can be GBs per peer: this rank's own share is a
    device copy, every other (source, destination) share travels in pieces of at most max_bytes (default 256 MiB,
    RFX_WGS_A2A_MAX_BYTES), one grouped isend/irecv round per piece.  RCCL 2.26's send/recv delivers only the first
    half of a message beyond ~1 GiB (measured on the one-rank group: scratch/a2a_big.py -- 1.0 GiB arrives whole,
    1.5 GiB and 10 GiB arrive as their first half), silently; pieces this size are far below that."""
    world, me = dist.get_world_size(group), dist.get_rank(group)
    so = [0] * (world + 1)
    ro = [0] * (world + 1)
    for d in range(world):
        so[d + 1] = so[d] + int(send_l[d])
        ro[d + 1] = ro[d] + int(recv_l[d])
    if int(send_l[me]) != int(recv_l[me]):
        raise ValueError("exchange_rows: a rank's share for itself differs between send and receive lists")
    if send_l[me]:
        out[ro[me]:ro[me + 1]].copy_(inp[so[me]:so[me + 1]])
    if world == 1:
        return
    if not max_bytes:
        max_bytes = int(os.environ.get("RFX_WGS_A2A_MAX_BYTES", 256 << 20))
    m = max(1, max_bytes // inp.element_size())
    most = max([int(send_l[d]) for d in range(world) if d != me] + [int(recv_l[d]) for d in range(world) if d != me] + [0])
    rounds = -(-most // m)
    for r in range(rounds):
        ops = []
        for d in range(world):
            if d == me:
                continue
            peer = dist.get_global_rank(group, d) if group is not None else d
            s0, s1 = min(r * m, int(send_l[d])), min((r + 1) * m, int(send_l[d]))
            if s1 > s0:
                ops.append(dist.P2POp(dist.isend, inp[so[d] + s0:so[d] + s1], peer, group))
            r0, r1 = min(r * m, int(recv_l[d])), min((r + 1) * m, int(recv_l[d]))
            if r1 > r0:
                ops.append(dist.P2POp(dist.irecv, out[ro[d] + r0:ro[d] + r1], peer, group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()


def exchange_partials(keys: torch.Tensor, counts: torch.Tensor, pos: torch.Tensor, lsize: int, group):
    """Send each (pos-sorted) partial to the owner of its pos; returns what this rank received."""
    world = dist.get_world_size(group)
    dev = keys.device
    bounds = torch.tensor(owner_bounds(lsize, world), dtype=torch.int64, device=pos.device)
    cuts = torch.searchsorted(pos, bounds)                 # partials are pos-sorted: contiguous runs
    send = _wire((cuts[1:] - cuts[:-1]).to(torch.int64), group)
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)
    send_l, recv_l = send.tolist(), recv.tolist()
    wk, wc = _wire(keys, group), _wire(counts, group)
    rk = torch.empty(sum(recv_l), dtype=keys.dtype, device=wk.device)
    rc = torch.empty(sum(recv_l), dtype=counts.dtype, device=wc.device)
    exchange_rows(rk, wk, recv_l, send_l, group)
    exchange_rows(rc, wc, recv_l, send_l, group)
    return rk.to(dev), rc.to(dev)


def exchange_records_begin(records: torch.Tensor, bin_start: torch.Tensor, group, planes: torch.Tensor | None = None):
    """Deal the bins to their owners: the (small) size and offset exchanges happen here, then the records travel
    (and, in a second exchange of the same shape, their 32-bit planes if given).  Finish with exchange_records_end()."""
    world, me = dist.get_world_size(group), dist.get_rank(group)
    dev = records.device
    bins = bin_start.numel() - 1
    b = bin_owner_bounds(bins, world)
    bs_host = bin_start.cpu()
    cuts = bs_host[torch.tensor(b)]
    send_l = (cuts[1:] - cuts[:-1]).tolist()
    send = _wire(torch.tensor(send_l, dtype=torch.int64, device=dev), group)
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)
    recv_l = recv.tolist()
    # bin offsets of every destination's range, relative to the start of its run.  Bin counts may differ
    # between ranks (they follow the block size), so the lengths travel first.
    parts = [bs_host[b[d]:b[d + 1] + 1] - bs_host[b[d]] for d in range(world)]
    len_s = _wire(torch.tensor([p.numel() for p in parts], dtype=torch.int64, device=dev), group)
    len_r = torch.empty_like(len_s)
    dist.all_to_all_single(len_r, len_s, group=group)
    len_sl, len_rl = len_s.tolist(), len_r.tolist()
    sb = _wire(torch.cat(parts).to(dev), group)
    rb = torch.empty(sum(len_rl), dtype=torch.int64, device=sb.device)
    dist.all_to_all_single(rb, sb, len_rl, len_sl, group=group)
    wr = _wire(records[:int(cuts[-1])], group)
    rr = torch.empty(sum(recv_l), dtype=records.dtype, device=wr.device)
    exchange_rows(rr, wr, recv_l, send_l, group)       # (in pieces, see exchange_rows: no longer asynchronous)
    re_ = None
    if planes is not None:
        we = _wire(planes[:int(cuts[-1])], group)
        re_ = torch.empty(sum(recv_l), dtype=planes.dtype, device=we.device)
        exchange_rows(re_, we, recv_l, send_l, group)
    return {"work": None, "rr": rr, "wr": wr, "rb": rb, "recv_l": recv_l, "len_rl": len_rl, "world": world, "me": me,
            "dev": dev, "re": re_}


def exchange_records_end(st):
    """Wait for the record all-to-all.  Returns [(records_from_rank, bin_start_full[, planes_from_rank])] for this
    rank's bins, one entry per source rank; bin_start_full has the sender's bin count + 1 entries (empty outside the
    owned range) so that the run can be imported as it is."""
    if st["work"] is not None:
        st["work"].wait()
    if st["rr"].is_cuda:       # wait() orders torch's stream only; the library runs on its own stream
        torch.cuda.current_stream(st["rr"].device).synchronize()
    world, me, dev, rr, rb = st["world"], st["me"], st["dev"], st["rr"], st["rb"]
    recv_l, len_rl = st["recv_l"], st["len_rl"]
    v0, v1 = -(-me * 256 // world), -(-(me + 1) * 256 // world)   # my range in virtual bins
    runs, ro, bo = [], 0, 0
    for src in range(world):
        loc = rb[bo:bo + len_rl[src]].cpu()
        nb = len_rl[src] - 1                      # bins of my range at the SENDER's resolution
        sbins = nb * 256 // (v1 - v0)             # hence the sender's bin count
        lo = v0 * (sbins // 256)
        full = torch.zeros(sbins + 1, dtype=torch.int64)
        full[lo:lo + nb + 1] = loc
        full[lo + nb + 1:] = loc[-1]
        run = (rr[ro:ro + recv_l[src]].to(dev), full.to(dev))
        if st.get("re") is not None:
            run += (st["re"][ro:ro + recv_l[src]].to(dev),)
        runs.append(run)
        ro += recv_l[src]
        bo += len_rl[src]
    return runs


def exchange_records(records: torch.Tensor, bin_start: torch.Tensor, group, planes: torch.Tensor | None = None):
    return exchange_records_end(exchange_records_begin(records, bin_start, group, planes))


def merge_shards(shards):
    """[(keys, counts, pos)] of the ranks' minimizer shards -> one (pos,key)-ordered triple (the .Jhash
    payload order); host-side, for callers that want the file."""
    keys = np.concatenate([s[0] for s in shards])
    counts = np.concatenate([s[1] for s in shards])
    pos = np.concatenate([s[2] for s in shards])
    o = np.lexsort((keys, pos))
    return keys[o], counts[o], pos[o]


def all_gather_keys(keys: np.ndarray, device, group) -> np.ndarray:
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "gloo":
        device = torch.device("cpu")
    n = torch.tensor([len(keys)], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes + [1])
    buf = torch.zeros(m, dtype=torch.int64, device=device)
    buf[:len(keys)] = torch.from_numpy(keys.astype(np.uint64).view(np.int64)).to(device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    return np.concatenate([o[:s].cpu().numpy().view(np.uint64) for o, s in zip(out, sizes)])


class TrioShard:
    """One rank's share of: count x (1 subject + controls) -> histogram -> hash list -> filter."""

    def __init__(self, ctx_or_backend, k: int, size: int, lower: int, min_cov: int, max_cov: int, thresh: int,
                 capacity: int = 0, group=None, shard_by: str | None = None, passes: int = 1):
        self.be = ctx_or_backend if hasattr(ctx_or_backend, "local_count") else HipBackend(ctx_or_backend, k, size,
                                                                                           capacity)
        self.k, self.lsize = k, capi.ceil_log2(size)
        self.lower, self.min_cov, self.max_cov, self.thresh = lower, min_cov, max_cov, thresh
        self.group = group
        self.world = dist.get_world_size(group) if group is not None else 1
        self.rank = dist.get_rank(group) if group is not None else 0
        self.passes = passes   # single rank: minimizer-shard passes over the samples (bounded HBM footprint)
        capable = getattr(self.be, "msp_capable", lambda: False)()
        self.shard_by = shard_by or ("minimizer" if capable else "pos")   # what a rank's records are a shard of
        if self.shard_by == "minimizer" and not capable:
            raise ValueError("minimizer sharding needs the MSP count path (23 <= k <= 25)")

    def count_sample(self, block):
        if self.world == 1:
            return self.be.local_count(block, self.lower)
        if self.shard_by == "minimizer":
            part = self.be.partition(block)
            records, bin_start = part[0], part[1]
            runs = exchange_records(records, bin_start, self.group, part[3] if len(part) > 3 else None)
            if len(part) > 2 and part[2] is not None:
                part[2].free()
            rec, histo = self.be.count_records(runs, self.lower)
            h = _wire(torch.from_numpy(histo.astype(np.int64)).to(records.device), self.group)
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            return rec, h.cpu().numpy().astype(np.uint64)
        keys, counts, pos = self.be.count_partials(block)
        rk, rc = exchange_partials(keys, counts, pos, self.lsize, self.group)
        b = owner_bounds(self.lsize, self.world)
        rec, histo = self.be.reduce_partials(rk, rc, self.lower, b[self.rank], b[self.rank + 1])
        h = _wire(torch.from_numpy(histo.astype(np.int64)).to(keys.device), self.group)
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
        return rec, h.cpu().numpy().astype(np.uint64)

    def _run_in_passes(self, blocks):
        """Single rank, S passes: pass s counts minimizer shard s of every sample and takes the set
        difference on it (the shard of a k-mer is the same in every sample); only one shard of each
        sample is ever resident.  Returns (mutant keys in (pos,key) order, histograms, record counts)."""
        keys, histos, n_rec = [], None, np.zeros(len(blocks), dtype=np.int64)
        for sh in range(self.passes):
            recs, hs = zip(*[self.be.local_count(blk, self.lower, shard=(sh, self.passes)) for blk in blocks])
            k_, _ = self.be.unique(recs[0], recs[1:], self.min_cov, self.max_cov)
            keys.append(k_)
            histos = [a + b for a, b in zip(histos, hs)] if histos else list(hs)
            n_rec += [self.be.n_records(r) for r in recs]
            for r in recs:
                self.be.free(r)
        keys = np.concatenate(keys)
        if len(keys):
            keys = keys[np.lexsort((keys, self.be.pos_of(keys)))]
        return keys, histos, n_rec.tolist()

    def run(self, subject_block, control_blocks, keep_records: bool = False):
        recs, histos = [], []
        blocks = [subject_block] + list(control_blocks)
        if self.world == 1 and self.passes > 1:
            keys, histos, n_rec = self._run_in_passes(blocks)
            res = self.be.filter_pairs(keys, subject_block, self.thresh)
            n_pulled, pulled_fn = res if isinstance(res, tuple) else (int(res.sum()), (lambda: res))
            return _ShardResult({"n_mutant": len(keys), "n_pulled": n_pulled, "n_records": n_rec, "histos": histos,
                                 "mutant_keys": keys, "_pulled_fn": pulled_fn})
        if self.world > 1 and self.shard_by == "minimizer":
            # The record all-to-all of a sample travels (RCCL on its own stream) while the next sample is
            # being partitioned and the previous one counted; histograms are reduced once, at the end.
            started = []
            for blk in blocks:
                part = self.be.partition(blk)          # (records, bin_start[, owner of their memory])
                keep = part[2] if len(part) > 2 else None
                started.append((exchange_records_begin(part[0], part[1], self.group, part[3] if len(part) > 3 else None), keep))
            local = []
            for st, keep in started:
                runs = exchange_records_end(st)
                if keep is not None:
                    keep.free()              # the send buffers were views of this table's memory
                rec, histo = self.be.count_records(runs, self.lower)
                recs.append(rec)
                local.append(histo.astype(np.int64))
            dev = started[0][0]["dev"]
            h = _wire(torch.from_numpy(np.stack(local)).to(dev), self.group)
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            histos = [x.astype(np.uint64) for x in h.cpu().numpy()]
        elif self.world == 1 and hasattr(self.be, "count_begin"):
            # queue all samples before waiting for the first: the device never idles between them
            pending = [self.be.count_begin(blk, self.lower) for blk in blocks]
            for p in pending:
                rec, h = self.be.count_end(p)
                recs.append(rec)
                histos.append(h)
        else:
            for blk in blocks:
                rec, h = self.count_sample(blk)
                recs.append(rec)
                histos.append(h)
        keys, counts = self.be.unique(recs[0], recs[1:], self.min_cov, self.max_cov)
        n_rec = [self.be.n_records(r) for r in recs]
        if self.world > 1:
            dev = torch.device("cpu") if dist.get_backend(self.group) == "gloo" else self.be.device
            keys = all_gather_keys(keys, dev, self.group)
            if self.shard_by == "minimizer" and len(keys):   # shard lists interleave in (pos,key) order
                keys = keys[np.lexsort((keys, self.be.pos_of(keys)))]
            t = torch.tensor(n_rec, dtype=torch.int64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            n_rec = t.tolist()
        res = self.be.filter_pairs(keys, subject_block, self.thresh)
        if isinstance(res, tuple):       # (count, unpack-on-demand): the HIP backend keeps the mask packed
            n_pulled, pulled_fn = res
        else:                            # a plain boolean vector (checker backends)
            n_pulled, pulled_fn = int(res.sum()), (lambda: res)
        if self.world > 1:
            t = torch.tensor([n_pulled], dtype=torch.int64,
                             device="cpu" if dist.get_backend(self.group) == "gloo" else self.be.device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            n_pulled = int(t.item())
        out = _ShardResult({"n_mutant": len(keys), "n_pulled": n_pulled, "n_records": n_rec, "histos": histos,
                            "mutant_keys": keys, "_pulled_fn": pulled_fn})
        if keep_records:
            out["records"] = recs
        else:
            for r in recs:
                self.be.free(r)
        return out
