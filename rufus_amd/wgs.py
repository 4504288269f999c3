"""WGS-scale driver of the trio hot path (BASELINE.json configs[2]-[3]: synthetic 30x WGS trio, k = 25).

A sample is a LIST of packed read blocks resident in HBM (a block holds < 2^32 k-mer windows, i.e. about
32 M reads of 150 bp).  A 30x human-size sample has 7.8e10 k-mer instances = 187 GB of super-k-mer
records -- more than fits next to the reads -- so the trio runs in S minimizer-shard passes
(``rfx_count_set_shard``): pass s counts shard s of every sample (all blocks into one table; the partition
is refined chunk by chunk at finish), takes the set difference on the shard (the shard of a k-mer is the
same in every sample), adds up the histograms and frees the shard's records.  The mutant k-mers of all
shards, put back into (pos,key) order, are the hash list; the subject's blocks are then filtered.

Everything that computes is a C-ABI call into the HIP library; this module only sequences them (the same
sequence the drop-in executables run per sample).
"""
from __future__ import annotations

import os
import time

import numpy as np

from . import capi
from .dist import revcomp_keys


class _DevMem32:
    """n int32 at a raw device address, for torch.as_tensor (no copy)."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i4", "data": (ptr, False), "version": 3,
                                         "strides": None}


def _device_view32(ptr: int, n: int, device):
    import torch
    return torch.as_tensor(_DevMem32(ptr, n), device=device)


class _LibraryCall:
    """fn(*args) on a helper thread (ctypes releases the GIL); .err = what it raised."""

    def __init__(self, fn, *args):
        import threading
        self.err = None
        self._t = threading.Thread(target=self._run, args=(fn, args), daemon=True)

    def _run(self, fn, args):
        try:
            fn(*args)
        except Exception as e:      # carried to the next checkpoint by the caller
            self.err = e

    def start(self):
        self._t.start()

    def join(self):
        self._t.join()


class GroupFailure(capi.RufusError):
    """A step failed on SOME rank of the group and every rank knows it: all of them raise this at the same
    checkpoint, so that none is left waiting in a collective.  retry: the failure was a lack of device memory --
    WgsTrio.run() takes one more pass on every rank and starts over."""

    def __init__(self, msg: str, retry: bool):
        super().__init__(msg)
        self.retry = retry


def _is_out_of_memory(e: BaseException) -> bool:
    return "memory" in str(e).lower() or type(e).__name__ == "OutOfMemoryError"


def shard_cut(q: int, n: int) -> int:
    """First of the 256 virtual top-level minimizer bins of shard q of n (rfx_count_set_shard's cut)."""
    return -(-q * 256 // n)

_EVEN = np.uint64(0x5555555555555555)


def pulled_pairs(mask: np.ndarray, n_reads: int) -> int:
    """Number of pairs (reads 2p, 2p+1) with a hit, from the packed per-read hit mask of a block."""
    m = mask[:(n_reads + 63) // 64]
    return int(np.bitwise_count((m | (m >> np.uint64(1))) & _EVEN).sum())


def plan_passes(n_reads_total: int, read_len: int, k: int, resident_bytes: int, hbm_bytes: int, n_samples: int = 3,
                coverage_hint: float = 30.0, world: int = 1, wide: bool = False) -> int:
    """Smallest number of shard passes whose transients fit beside the resident reads of ONE rank.

    n_reads_total: reads per sample over all ranks.  Per pass, sample and rank: super-k-mer records (8 B per
    ~3 k-mer instances; 1 / (S x world) of the sample -- two copies alive at the peak of an exchange: the
    partition and the receive buffers, which the owner counts in place), the refinement scratch (1/8 of the records), the survivor
    arrays (44 B per surviving k-mer: two partition levels + the records), and the subject's candidate records
    of this pass (20 B each; WgsTrio.run counts the subject first and keeps only them)."""
    windows = n_reads_total * max(read_len - k + 1, 0)           # per sample
    distinct = windows / max(coverage_hint * (read_len - k + 1) / read_len, 1.0)
    for s in range(1, 257 // world):
        share = s * world
        # bytes of records per k-mer instance: a record (12 B) is a whole super-k-mer -- 5.7 k-mers for k <= 25
        # (window of 11 m-mers), 8 for k = 26 .. 31 (window of 16); until round 3: 8 B per ~3 k-mers
        records = (1.6 if wide else 2.2) * windows / share * (2.0 if world > 1 else 1.0)
        # leaf phase of the last sample of a pass: records + scratch, survivor store, the other samples' records
        # (the controls are struck off the subject's candidates one at a time: one set of 20-byte records stays)
        transient = records * 1.125 + 12.0 * distinct / share * 1.3 + (20.0 if n_samples > 1 else 0.0) * distinct / share
        # (measured on the 30x WGS trio, 1 GPU, of 288 GiB: 219 GB at 5 passes, 238 at 4, 271 at 3 -- which this
        # picks; WgsTrio.run() takes one more pass and starts over should a pass not fit after all)
        if resident_bytes + transient < 0.90 * hbm_bytes:
            return s
    return max(1, 256 // world)


class WgsTrio:
    """count x (subject + controls) -> histograms -> hash list -> filter, in minimizer-shard passes."""

    def __init__(self, ctx: capi.Context, k: int, size: int, lower: int, min_cov: int, max_cov: int, thresh: int,
                 passes: int = 1, group=None):
        """group: a torch.distributed process group -- every rank holds ITS blocks of every sample (strong
        scaling of one trio); records travel to the owner of their minimizer bin, see count_shard()."""
        self.ctx, self.k, self.size = ctx, k, size
        self.lsize = capi.ceil_log2(size)
        self.cols = capi.jf_matrix(self.lsize, k)
        self.lower, self.min_cov, self.max_cov, self.thresh = lower, min_cov, max_cov, thresh
        self.passes = passes
        self.group = group
        self.early_budget = 0       # bytes of device memory that may hold records cut ahead for the next shard pass
        self._early, self._early_left, self._early_cost = {}, 0, 0
        # bytes of device memory that may hold RUN MAPS (32 B per read; rfx_runmaps_*): the first shard pass over a block
        # leaves one, the later passes rebuild their records from reads + map instead of hashing the block again
        # (one store for the trio, ONE piece of device memory of that size, kept from run to run: at 90 % of the HBM maps
        # allocated between the transients of a pass left the arena in pieces.)  With maps the passes are ordered so that
        # at most two samples' maps are alive: run().
        self.map_budget = 0
        self._store = None
        self._ahead = {}            # (pass, sample) -> the blocks whose maps that step queues ahead (run())
        self.maps_ahead = 0         # hashing launches queued on the second stream in the last run()
        self.count_wall_s = 0.0     # host wall time inside count_shard() in the last run() (every count ends with a wait)
        self.masks_are_views = False  # run()'s hit masks stay views of the ctx's page-locked read-back buffer (valid until the next run())
        self.replayed_blocks = 0    # blocks added by replay in the last run()
        # bytes of super-k-mer records this rank sent to / received from OTHER ranks in the last run() (the record
        # exchange of count_shard: RCCL all-to-all over xGMI; what stays on the rank is not counted)
        self.exchange_sent = self.exchange_received = 0
        if group is not None:
            import torch.distributed as dist
            self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        else:
            self.world, self.rank = 1, 0
        if self.passes * self.world > 256:
            raise ValueError("passes x ranks must not exceed the 256 virtual minimizer bins")

    def _count_shard_local(self, blocks, shard: int, si):
        """One device: the table of this (sample, shard).  Shard passes hash every block once per pass; with
        `early_budget` bytes of headroom (bench.py: what the device has left beside the peak of a step) the blocks are, as
        far as that goes, hashed ONCE for this shard and the next (rfx_count_set_early): the next shard's records wait
        in a table of their own (self._early) and that pass is given only the other blocks."""
        held = self._early.pop((si, shard), None) if si is not None else None
        if held is not None:
            t, done = held
        else:
            t, done = capi.CountTable(self.ctx, self.k, self.size, True, mode=capi.COUNT_MSP), frozenset()
            if self.passes > 1:
                t.set_shard(shard, self.passes)
        nxt = None
        store = self._store if si is not None and self.passes > 1 else None
        try:
            if store is not None:       # (run maps instead of blocks cut ahead: 32 instead of 132 bytes per read)
                t.set_runmaps(store)
                t.prepare_maps(blocks)       # (all of them behind ONE wait: 0.75 ms of idle device per block otherwise)
                # Round 6: the maps of the sample that is counted NEXT are hashed on the ctx's second stream while this
                # sample's records are partitioned, refined and sorted on the first (run(): self._ahead): the hashing launch
                # is bound by the instructions it issues, the partition levels by the memory, and they share a CU.
                nxt_blocks = self._ahead.pop((shard, si), None) if self._ahead else None
                if nxt_blocks:
                    self.maps_ahead += t.prefetch_maps(nxt_blocks)
                last = shard == self.passes - 1
                for b in blocks:
                    t.add(b)
                    if last:
                        store.drop(b)
                self.replayed_blocks += t.replayed()
                if shard == 0:
                    self._inject("maps", shard)      # (tests: the headroom turns out not to be there)
                return t.finish(self.lower, want_histo=True)
            todo = [i for i in range(len(blocks)) if i not in done]
            ahead = set()
            if si is not None and self.passes > 1 and shard + 1 < self.passes and self._early_left > 0 and not self.map_budget:
                t.set_early(True)
                cost = self._early_cost
                while todo and self._early_left > cost:     # (cost: what the block before took)
                    used0, n0 = self.ctx.mem_stats()["used"], t.early_segments()
                    i = todo.pop(0)
                    t.add(blocks[i])
                    if t.early_segments() == n0:            # not a block for the early cut (small, or the shards meet
                        break                               # inside a coarse bin): neither will the others be
                    ahead.add(i)
                    # (the block's own segment came with it: the early half is the next shard's share of the growth)
                    cost = self._early_cost = (self.ctx.mem_stats()["used"] - used0) // 2
                    self._early_left -= cost
                t.set_early(False)
                if ahead:
                    self._inject("early", shard)     # (tests: the headroom turns out not to be there)
            for i in todo:
                t.add(blocks[i])
            if ahead:
                nxt = capi.CountTable(self.ctx, self.k, self.size, True, mode=capi.COUNT_MSP)
                nxt.set_shard(shard + 1, self.passes)
                nxt.adopt_early(t)
                self._early[(si, shard + 1)] = (nxt, frozenset(ahead))
                nxt = None
            return t.finish(self.lower, want_histo=True)
        finally:
            if nxt is not None:
                nxt.free()
            t.free()

    def _drop_early(self):
        for t, _ in self._early.values():
            t.free()
        self._early = {}
        if self._store is not None:
            self._store.clear()

    def _drop_store(self):
        if self._store is not None:
            self._store.free()
            self._store = None

    def close(self):
        """Give back what the driver holds between runs (the pool of the run maps)."""
        self._drop_early()
        self._drop_store()

    def count_shard(self, blocks, shard: int, si=None):
        """Records (in (pos,key) order) + histogram of the k-mers of minimizer shard `shard` of `passes` --
        on N ranks: of this rank's 1/N of that shard.  The cut is flat over passes x ranks virtual shards
        (q = shard * N + rank of passes * N), so a pass is a contiguous range of bins split among the ranks:
        every rank partitions ITS blocks restricted to the pass (rfx_count_set_shard(shard, passes)), the
        records of owner g's bins are one contiguous run per segment -> one all_to_all_single per segment
        (RCCL over xGMI), the owner imports the runs and counts complete bins.  No partial counts, no reduce."""
        if self.world == 1 and not (self.group is not None and os.environ.get("RFX_WGS_FORCE_EXCHANGE")):
            return self._count_shard_local(blocks, shard, si)
        t = capi.CountTable(self.ctx, self.k, self.size, True, mode=capi.COUNT_MSP)
        try:
            err = None
            try:        # local work: a failure here is carried to the first checkpoint of the exchange
                if self.passes > 1:
                    t.set_shard(shard, self.passes)
                self._inject("partition", shard)
                # on a group the other blocks are partitioned while the records of the one before travel
                for b in blocks[:1] if self._overlap() else blocks:
                    t.add(b)
            except Exception as e:
                err = e
            return self._exchange_and_count(t, shard, err, blocks)
        finally:
            t.free()

    def _overlap(self) -> bool:
        """Partition block i + 1 while the records of block i travel?  Yes between devices (RCCL over xGMI moves
        ~0.3 TB/s, a fraction of what the partition leaves of the HBM bandwidth); no when the "exchange" is a copy
        inside one device (RFX_WGS_FORCE_EXCHANGE on one rank: the copy runs at HBM speed against the partition's
        scattered stores -- measured: 66-86 ms per round together, 25 + 5 ms one after the other).  RFX_WGS_OVERLAP=0/1
        overrides.  Not measured on several GPUs (a one-GPU box cannot): profiles/r03_exchange_big.txt."""
        ev = os.environ.get("RFX_WGS_OVERLAP")
        return (ev != "0") if ev is not None else self.world > 1

    def checkpoint(self, err=None):
        """All ranks: did the step succeed everywhere?  One 8-byte all-reduce (MIN of 2 = fine, 1 = out of device
        memory, 0 = anything else).  Returns when every rank is fine; otherwise EVERY rank raises GroupFailure here
        (the rank that failed chains its own exception), retry = all failures were a lack of memory."""
        if self.world == 1:
            if err is not None:
                raise err
            return
        import torch
        import torch.distributed as dist
        from .dist import _wire
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        code = 2 if err is None else (1 if _is_out_of_memory(err) else 0)
        t = _wire(torch.tensor([code], dtype=torch.int64, device=dev), self.group)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        worst = int(t.item())
        if worst == 2:
            return
        what = "out of device memory" if worst == 1 else "a step failed"
        where = f"on this rank ({err})" if err is not None else "on another rank of the group"
        raise GroupFailure(f"{what} {where}", retry=worst == 1) from err

    def _inject(self, stage: str, shard: int):
        """Fault injection for the tests of the agreed retry: RFX_WGS_INJECT_OOM="rank:stage:shard" makes that rank
        fail once with an out-of-memory error at that stage (partition | receive | finish; early: after blocks were cut
        ahead on one device) of that shard."""
        spec = os.environ.get("RFX_WGS_INJECT_OOM")
        if not spec or getattr(self, "_injected", False):
            return
        r, st, sh = spec.split(":")
        if int(r) == self.rank and st == stage and int(sh) == shard:
            self._injected = True
            raise capi.RufusError(f"injected: out of device memory at {stage} of shard {shard}")

    def _exchange_and_count(self, part: capi.CountTable, shard: int, err, blocks):
        """One round per block: the records of block i go to the owners of their bins (RCCL, torch's stream) WHILE
        block i + 1 is partitioned (the library's stream; rfx_count_add waits for a big block's sizes, so it is called
        from a helper thread -- the library is used by one thread at a time: the main thread makes no library call
        between the helper's start and its join).  part holds blocks[0] already (all blocks when _overlap() says no);
        err: what that raised, if anything.
        Checkpoints (see checkpoint()): before the first round, in every round after the receive buffers are
        allocated (before any record travels), and after the owner's count -- a failure between two of them is
        carried to the next one, never into a collective."""
        import torch
        import torch.distributed as dist
        from .dist import _device_view, _wire, exchange_rows
        W, me, Q = self.world, self.rank, self.passes * self.world
        dev = torch.device("cuda", torch.cuda.current_device())
        vb = [shard_cut(shard * W + g, Q) for g in range(W + 1)]       # owners' ranges in virtual bins
        self.checkpoint(err)
        # every rank may hold a different number of blocks: agree on the rounds
        n_seg = torch.tensor([len(blocks)], dtype=torch.int64, device=dev)
        n_seg = _wire(n_seg, self.group)
        dist.all_reduce(n_seg, op=dist.ReduceOp.MAX, group=self.group)
        rounds = int(n_seg.item())
        own = capi.CountTable(self.ctx, self.k, self.size, True, mode=capi.COUNT_MSP)
        trace = os.environ.get("RFX_WGS_TRACE")
        tt = {"segment": 0.0, "meta": 0.0, "alloc": 0.0, "move": 0.0, "join": 0.0, "import": 0.0}

        def lap(what, t0):      # RFX_WGS_TRACE=host: host wall time only (no device synchronisation: the overlap stays)
            if trace:
                if trace != "host":
                    torch.cuda.synchronize(dev)
                tt[what] += time.perf_counter() - t0
            return time.perf_counter()
        helper = None
        overlap = self._overlap()
        try:
            own.set_shard(shard * W + me, Q)
            keep = []
            wide = True             # records = 64-bit word + 32-bit plane: the plane travels in a second all-to-all
            seg_done = 0
            for i in range(rounds):
                ext = None
                seg = None
                d_ext = 0
                t_ = time.perf_counter()
                if err is None and i < len(blocks):
                    try:        # block i's segment, if it made one (waits for its partition)
                        if part.n_segments() > seg_done:
                            seg = part.segment(seg_done)
                            d_ext = part.segment_ext(seg_done) if wide else 0
                            seg_done += 1
                    except Exception as e:
                        err = e
                t_ = lap("segment", t_)
                helper = None
                if seg is not None:
                    d_rec, d_bs, bins, n = seg
                    rec = _device_view(d_rec, n, dev) if n else torch.empty(0, dtype=torch.int64, device=dev)
                    bs_host = _device_view(d_bs, bins + 1, dev).cpu()
                    if wide:
                        ext = (_device_view32(d_ext, n, dev) if n and d_ext else
                               torch.empty(0, dtype=torch.int32, device=dev))
                else:                                   # nothing to send in this round
                    bins = 256
                    rec = torch.empty(0, dtype=torch.int64, device=dev)
                    bs_host = torch.zeros(bins + 1, dtype=torch.int64)
                    if wide:
                        ext = torch.empty(0, dtype=torch.int32, device=dev)
                per = bins // 256
                cuts = bs_host[torch.tensor([v * per for v in vb])]
                send_l = (cuts[1:] - cuts[:-1]).tolist()
                # sizes, then the bin offsets of each destination's range (relative to its run), then the records
                meta_s = _wire(torch.tensor([[send_l[d], bins] for d in range(W)], dtype=torch.int64, device=dev).flatten(),
                               self.group)
                meta_r = torch.empty_like(meta_s)
                dist.all_to_all_single(meta_r, meta_s, group=self.group)
                meta_r = meta_r.view(W, 2).tolist()
                recv_l = [m[0] for m in meta_r]
                off_parts = [bs_host[vb[d] * per:vb[d + 1] * per + 1] - bs_host[vb[d] * per] for d in range(W)]
                off_sl = [p.numel() for p in off_parts]
                off_rl = [(vb[me + 1] - vb[me]) * (m[1] // 256) + 1 for m in meta_r]
                sb = _wire(torch.cat(off_parts).to(dev), self.group)
                rb = torch.empty(sum(off_rl), dtype=torch.int64, device=sb.device)
                dist.all_to_all_single(rb, sb, off_rl, off_sl, group=self.group)
                t_ = lap("meta", t_)
                rr = re_ = None
                try:
                    self._inject("receive", shard)
                    wr = _wire(rec[int(cuts[0]):int(cuts[-1])], self.group)
                    rr = torch.empty(sum(recv_l), dtype=torch.int64, device=wr.device)
                    if wide:
                        we = _wire(ext[int(cuts[0]):int(cuts[-1])], self.group)
                        re_ = torch.empty(sum(recv_l), dtype=torch.int32, device=we.device)
                except Exception as e:
                    err = e
                self.checkpoint(err)        # also carries a failed import / partition of the round before
                t_ = lap("alloc", t_)
                # Only now, with the round's small collectives done: a partition kernel fills every CU for its ~10 ms,
                # and anything queued behind it on the device waits that long -- measured on one GPU: with the helper
                # started before the metadata exchange every one of its two all-to-alls took a block's partition time
                # (16 ms per round, +2.5 s per W trio in 171 rounds).  The bulk transfer can wait; latency cannot.
                if overlap and i + 1 < len(blocks):
                    helper = _LibraryCall(part.add, blocks[i + 1])
                    helper.start()
                per_rec = 12 if wide else 8
                self.exchange_sent += per_rec * (sum(send_l) - send_l[me])
                self.exchange_received += per_rec * (sum(recv_l) - recv_l[me])
                exchange_rows(rr, wr, recv_l, send_l, self.group)
                if wide:
                    exchange_rows(re_, we, recv_l, send_l, self.group)
                    re_ = re_.to(dev)
                if rr.is_cuda:
                    torch.cuda.current_stream(rr.device).synchronize()   # the library runs on its own stream
                rr, rb = rr.to(dev), rb.cpu()
                t_ = lap("move", t_)
                if helper is not None:      # the next block is partitioned (or failed to be): the library is ours again
                    helper.join()
                    err = err or helper.err
                t_ = lap("join", t_)
                # the bin offsets of every source's run go up first, ONE synchronisation of torch's stream covers them
                # all (round 2 waited once per source), then the imports are queued back to back on the library's stream
                ro = bo = 0
                todo = []
                for src in range(W):
                    sbins = meta_r[src][1]
                    sper = sbins // 256
                    loc = rb[bo:bo + off_rl[src]]
                    full = torch.zeros(sbins + 1, dtype=torch.int64)
                    lo = vb[me] * sper
                    full[lo:lo + off_rl[src]] = loc
                    full[lo + off_rl[src]:] = loc[-1]
                    full = full.to(dev, non_blocking=True)
                    run = rr[ro:ro + recv_l[src]]
                    run_e = re_[ro:ro + recv_l[src]] if wide else None
                    todo.append((run, full, run_e, sbins, recv_l[src]))
                    ro += recv_l[src]
                    bo += off_rl[src]
                torch.cuda.synchronize(dev)
                # the owner counts the runs where they landed (rfx_count_adopt_records_dev: no third copy of the shard's
                # records); the receive buffers live until the table is finished
                try:
                    for run, full, run_e, sbins, n_run in todo:
                        if n_run:
                            own.adopt_records_dev(run.data_ptr(), run.numel(), full.data_ptr(), sbins,
                                                  run_e.data_ptr() if wide else 0)
                        keep.append((run, full, run_e))
                except Exception as e:
                    err = e
                del rr
                t_ = lap("import", t_)
            if trace:
                print("[wgs] exchange of shard %d: " % shard + ", ".join(f"{k_} {v * 1e3:.0f} ms" for k_, v in tt.items()), flush=True)
            part.free()                     # the send views were this table's memory
            out = None
            if err is None:
                try:
                    self._inject("finish", shard)
                    out = own.finish(self.lower, want_histo=True)
                    self.ctx.sync()
                except Exception as e:
                    err = e
            keep.clear()
            self.checkpoint(err)
            return out
        finally:
            if helper is not None:      # (left by an exception: nobody else may be inside the library when tables go)
                helper.join()
            own.free()

    def _after_count(self, rec, recs, si, sh, cand, ver, verify, probe_keys, keep_shard_records, lap):
        """What run() does with one sample's records of one pass (verify, candidates / strike-out); returns cand."""
        if verify:
            v = rec.verify(self.lower)
            for k_ in ("bad_order", "bad_pos", "bad_count"):
                ver[k_] += v[k_]
            ver["sum_counts"][si] += v["sum_counts"]
            ver["checksum"][si] = [(a + b) % (1 << 64) for a, b in zip(ver["checksum"][si], rec.checksum())]
            if probe_keys is not None and len(probe_keys):
                got = rec.query(np.asarray(probe_keys, dtype=np.uint64))
                ver["probe_found"][si] += int((got > 0).sum())
                if si == 0:
                    ver["probe_count_out_of_range"] += int(((got > 0) & ((got < max(5, self.min_cov)) |
                                                                         (got > self.max_cov))).sum())
            lap(f"pass {sh} sample {si} verify")
        if keep_shard_records:
            return cand
        if si == 0:
            # The subject's records stay as they are until the first control is there: "MinCov <= count <= MaxDepth" and
            # "not in control 1" are then ONE pass over them (round 4; before, the range alone made a 31 GB copy of a W
            # shard that the next step read again: 20 ms per pass).  Without any control the range is applied at the end.
            self._cand_raw = True
            recs.pop()
            lap(f"pass {sh} sample {si} kept ({len(rec)})")
            return rec
        raw = getattr(self, "_cand_raw", False)
        raw_bytes = len(cand) * 20 if raw else 0
        nxt = (capi.records_subtract(self.ctx, cand, [rec], max(5, self.min_cov), self.max_cov) if raw else
               capi.records_subtract(self.ctx, cand, [rec]))
        self._cand_raw = False
        cand.free()
        cand = nxt
        # The step's peak is the count of the FIRST control of a pass (the subject's records wait beside it); from here on
        # the pass runs that much lower, and so may hold that much more of the next pass's records cut ahead -- provided the
        # samples they belong to are counted BEFORE the first control of that pass (run() reverses the controls' order on
        # odd passes).  Two passes only: with more, what is held overlaps from pass to pass.
        if raw and self.passes == 2 and sh == 0 and self.early_budget > 0:
            self._early_left += raw_bytes
        rec.free()
        recs.pop()
        lap(f"pass {sh} sample {si} candidates ({len(cand)})")
        return cand

    def pos_of(self, keys: np.ndarray) -> np.ndarray:
        """pos = (M * key) & (2^lsize - 1): bit b of the key selects column 2k-1-b."""
        keys = np.asarray(keys, dtype=np.uint64)
        pos = np.zeros(len(keys), dtype=np.uint64)
        c = 2 * self.k
        for b in range(c):
            pos ^= np.where((keys >> np.uint64(b)) & np.uint64(1), self.cols[c - 1 - b], np.uint64(0))
        return pos & np.uint64((1 << self.lsize) - 1) if self.lsize < 64 else pos

    def run(self, samples, keep_shard_records: bool = False, verify: bool = False, probe_keys=None):
        """samples: [subject blocks, control blocks, ...] (lists of capi.ReadBlock).

        verify: every shard's records are checked where they lie before they are freed (rfx_records_verify: strict
        (pos,key) order, pos == M * key, lower <= count) -> out["verify"]; probe_keys (canonical keys, e.g. the hash
        list of an earlier run): out["verify"]["probe_found"][sample] = how many of them each sample holds."""
        trace = os.environ.get("RFX_WGS_TRACE")
        t_last = time.perf_counter()

        def lap(what):
            nonlocal t_last
            if trace:
                self.ctx.sync()
                now = time.perf_counter()
                print(f"[wgs] {what}: {(now - t_last) * 1e3:.1f} ms", flush=True)
                t_last = now

        while True:
            histos = [np.zeros(capi.HISTO_BINS, dtype=np.uint64) for _ in samples]
            n_rec = [0] * len(samples)
            keys, kept, recs = [], [], []
            cand, cands, shard_recs = None, {}, {}
            self._drop_early()
            self._early_left = int(self.early_budget) if self.world == 1 and not keep_shard_records else 0
            self.replayed_blocks = 0
            self.exchange_sent = self.exchange_received = 0
            use_maps = self.map_budget > 0 and self.world == 1 and self.passes > 1 and not keep_shard_records
            if not use_maps:
                self._drop_store()
            elif self._store is None:
                try:
                    self._store = capi.RunMaps(self.ctx, int(self.map_budget), pooled=True)
                except capi.RufusError:         # no room for the pool: the passes hash as before
                    self.map_budget, use_maps = 0, False
            ver = {"bad_order": 0, "bad_pos": 0, "bad_count": 0, "sum_counts": [0] * len(samples),
                   "probe_found": [0] * len(samples), "probe_count_out_of_range": 0,
                   "checksum": [[0, 0] for _ in samples]}   # rfx_records_checksum, summed over the shards
            try:
                # The subject (sample 0) is counted first and only its CANDIDATES stay: the records with MinCov <=
                # count <= MaxDepth; every control then strikes out what it holds and is freed at once
                # (rfx_records_subtract) -- one sample's records alive at a time instead of all of them.
                # Order of the (pass, sample) steps.  Plain: pass by pass (odd passes take the controls in reverse: the
                # control counted last in one pass -- whose records of the next pass may have been cut ahead in the room
                # the subject's records left, see _after_count -- is the first control of the next).  With run maps:
                # subject and first control pass by pass, then every further control through all its passes at once --
                # the candidates a shard has left after the first control are few, so they can wait, and never more
                # than two samples' maps are alive (the pool is sized for that: bench.py).
                P, ns = self.passes, len(samples)
                if use_maps and ns > 2:
                    steps = [(sh, si) for sh in range(P) for si in (0, 1)] + [(sh, si) for si in range(2, ns) for sh in range(P)]
                else:
                    steps = [(sh, si) for sh in range(P)
                             for si in (list(range(ns)) if sh % 2 == 0 or keep_shard_records else [0] + list(range(ns - 1, 0, -1)))]
                last_step = {sh: max(i for i, (sh_, _) in enumerate(steps) if sh_ == sh) for sh in range(P)}
                shard_recs.update({sh: [] for sh in range(P)})
                # maps made ahead: the step before a sample's FIRST step queues that sample's maps -- if the store has room
                # for them then (two samples' maps fit: the step must come after the last step of the sample before the
                # previous one, whose maps go with its last pass)
                self._ahead, self.maps_ahead, self.count_wall_s = {}, 0, 0.0
                # (opt-in, RFX_MAP_AHEAD=1: measured at W it buys nothing -- 1943 against 1941 ms per step --: the hashing
                # launch holds 414 of a SIMD's 512 registers, the replay and partition workgroups of the first stream do
                # not fit beside it, and the two streams take turns instead of sharing the CUs)
                if use_maps and os.environ.get("RFX_MAP_AHEAD"):
                    first = {}
                    for i, (_, si_) in enumerate(steps):
                        first.setdefault(si_, i)
                    last_of = {si_: max(i for i, (_, s_) in enumerate(steps) if s_ == si_) for si_ in first}
                    order = sorted(first, key=first.get)
                    for n_, si_ in enumerate(order[1:], 1):
                        at = first[si_] - 1                      # the step that queues them
                        if n_ >= 2 and at <= last_of[order[n_ - 2]]:
                            continue                             # (three samples' maps would be alive)
                        self._ahead[steps[at]] = samples[si_]
                for i_step, (sh, si) in enumerate(steps):
                    recs = shard_recs[sh]
                    cand = cands.pop(sh, None)
                    blocks = samples[si]
                    t_c = time.perf_counter()
                    rec, h = self.count_shard(blocks, sh, si)   # (its failures are agreed inside)
                    self.count_wall_s += time.perf_counter() - t_c
                    recs.append(rec)
                    histos[si] += h
                    n_rec[si] += len(rec)
                    lap(f"pass {sh} sample {si} count ({len(rec)} records)")
                    err = None              # the rest of the iteration is local: on a group its failure is
                    try:                    # agreed at its end, so that no rank goes on into a collective alone
                        cand = self._after_count(rec, recs, si, sh, cand, ver, verify, probe_keys, keep_shard_records, lap)
                    except Exception as e:
                        err = e
                    self.checkpoint(err)    # (one rank: raises err)
                    if i_step != last_step[sh]:
                        if cand is not None:
                            cands[sh] = cand
                        cand = None
                        continue
                    if keep_shard_records:
                        k_, _ = capi.unique_to_subject(self.ctx, recs[0], recs[1:], self.min_cov, self.max_cov)
                        kept.append(recs)
                    else:
                        if getattr(self, "_cand_raw", False):      # a subject without controls: the range alone
                            nxt = capi.records_subtract(self.ctx, cand, [], max(5, self.min_cov), self.max_cov)
                            cand.free()
                            cand = nxt
                            self._cand_raw = False
                        k_ = cand.get()[0]
                        cand.free()
                        cand = None
                    lap(f"pass {sh} set difference ({len(k_)} k-mers)")
                    keys.append(k_)
                    shard_recs[sh] = []
                    recs = []
                break
            except capi.RufusError as e:
                # the pass plan is an estimate: if a pass does not fit after all, take one more pass and start over.
                # On a group only a failure every rank knows of (GroupFailure: raised by all ranks at the same
                # checkpoint of count_shard) can be retried -- all ranks are here then, with the same `passes`.
                agreed = isinstance(e, GroupFailure) and e.retry
                self._drop_early()              # (run maps / early tables of the failed attempt, whatever follows)
                if (self.world > 1 and not agreed) or self.passes >= 64 or (self.passes + 1) * self.world > 256 or \
                        not _is_out_of_memory(e):
                    raise
                held = [r_ for rr in shard_recs.values() for r_ in rr]
                held += [r_ for r_ in recs if r_ not in held]
                held += [r_ for shard in kept for r_ in shard if r_ not in held]
                held += [c_ for c_ in [cand] + list(cands.values()) if c_ is not None and c_ not in held]
                for r in held:
                    r.free()
                self._drop_early()
                if self.world > 1:
                    import torch
                    torch.cuda.empty_cache()    # the receive buffers of the failed pass go back to the driver
                if self.early_budget > 0 or self.map_budget > 0:    # the headroom for run maps / blocks cut ahead was not
                    self._drop_store()
                    self.early_budget = self.map_budget = 0           # there after all: the same passes once more without
                    if trace:
                        print("[wgs] out of device memory: retrying without run maps / blocks cut ahead", flush=True)
                    continue
                self.passes += 1
                if trace:
                    print(f"[wgs] out of device memory: retrying with {self.passes} passes", flush=True)
        keys = np.concatenate(keys) if keys else np.zeros(0, np.uint64)
        if self.world > 1:     # every rank needs the whole hash list; histograms and record counts add up
            import torch
            import torch.distributed as dist
            from .dist import all_gather_keys, _wire
            dev = torch.device("cuda", torch.cuda.current_device())
            keys = all_gather_keys(keys, dev, self.group)
            h = _wire(torch.from_numpy(np.stack(histos).astype(np.int64)).to(dev), self.group)
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            histos = [x.astype(np.uint64) for x in h.cpu().numpy()]
            nr = _wire(torch.tensor(n_rec, dtype=torch.int64, device=dev), self.group)
            dist.all_reduce(nr, op=dist.ReduceOp.SUM, group=self.group)
            n_rec = nr.tolist()
        if len(keys):
            keys = keys[np.lexsort((keys, self.pos_of(keys)))]
        lap("hash list order")
        n_pulled = 0
        masks = []
        if len(keys):
            mset = capi.MutantSet(self.ctx, np.concatenate([keys, revcomp_keys(keys, self.k)]), self.k)
            try:
                # (all the subject's blocks behind one wait: a call per block left the device idle between two blocks)
                for b, (mask, _) in zip(samples[0], mset.filter_many(samples[0], self.thresh, last_base_skipped=True)):
                    n_pulled += pulled_pairs(mask, b.n)
                    # (views of the ctx's page-locked buffer, overwritten by the next run(): copied unless the caller says
                    # it is done with them by then -- bench.py's timed steps)
                    masks.append(mask if self.masks_are_views else mask.copy())
            finally:
                mset.free()
        lap("filter")
        n_pulled_local = n_pulled
        if self.world > 1:
            t_ = _wire(torch.tensor([n_pulled], dtype=torch.int64, device=dev), self.group)
            dist.all_reduce(t_, op=dist.ReduceOp.SUM, group=self.group)
            n_pulled = int(t_.item())
        out = {"n_mutant": len(keys), "n_pulled_local": n_pulled_local, "mutant_keys": keys, "n_pulled": n_pulled, "n_records": n_rec, "histos": histos,
               "hit_masks": masks}
        if keep_shard_records:
            out["shard_records"] = kept
        if verify:
            if self.world > 1:
                v_ = _wire(torch.tensor([ver["bad_order"], ver["bad_pos"], ver["bad_count"], ver["probe_count_out_of_range"]]
                                        + ver["sum_counts"] + ver["probe_found"], dtype=torch.int64, device=dev), self.group)
                dist.all_reduce(v_, op=dist.ReduceOp.SUM, group=self.group)
                v_ = v_.tolist()
                n_ = len(samples)
                ver.update(bad_order=v_[0], bad_pos=v_[1], bad_count=v_[2], probe_count_out_of_range=v_[3],
                           sum_counts=v_[4:4 + n_], probe_found=v_[4 + n_:4 + 2 * n_])
            out["verify"] = ver
        return out


_COMP = bytes.maketrans(b"ACGT", b"TGCA")


def expected_snv_kmers(sy: capi.Synth, k: int) -> set:
    """Canonical k-mers that carry the alt allele of a planted SNV of the subject (k per SNV)."""
    expect = set()
    for p, ref, alt in sy.snvs():
        ctxt = bytearray(sy.genome(p - k + 1, 2 * k - 1))
        if bytes(ctxt[k - 1:k]) != ref:
            raise AssertionError("generator: the reference base of an SNV is not the genome's")
        ctxt[k - 1:k] = alt
        for i in range(k):
            km = bytes(ctxt[i:i + k])
            expect.add(min(km, km[::-1].translate(_COMP)))
    return expect


def valid_windows_of_text(seq: np.ndarray, k: int) -> int:
    """ACGT-only windows of length k of a (reads x length) uint8 matrix of read text (host arithmetic only)."""
    ok = np.isin(seq, np.frombuffer(b"ACGTacgt", np.uint8))
    run = np.zeros(seq.shape[0], dtype=np.int32)
    tot = 0
    for j in range(seq.shape[1]):
        run = np.where(ok[:, j], run + 1, 0)
        tot += int((run >= k).sum())
    return tot


def self_check(ctx: capi.Context, trio: "WgsTrio", samples, sys_, res, n_pairs, min_q: int = 15,
               more_passes: bool = True, sample_pairs: int = 1 << 17) -> dict:
    """What must hold at ANY size, checked on the data of a finished run `res` of `trio` on `samples` (bench.py runs
    this after its timed region; tests/test_scale_gpu.py at the full size of BASELINE configs[2]).  Raises
    AssertionError; returns a summary for the bench line.

    1. a second run with every shard's records verified on the device: strict (pos,key) order
       (jf/include/jellyfish/sorted_dumper.hpp:80-112), pos == M * key, every count >= lower; the sum of the counts
       equals sum(i * histo[i]); the same record counts, histograms, hash list and pulled pairs as `res`;
    2. the mutant k-mers: each is held by the subject with MinCov <= count <= MaxDepth and by NO control (looked up
       in every shard of every sample: merge_files.cc:69-155 + CheckJellyHashList.sh:12 semantics), and they are the
       alt-allele k-mers of the planted SNVs (a handful of recurrent sequencing errors aside);
    3. one more shard pass (S + 1) gives the same record counts, histograms, hash list and pulled pairs -- and the same
       multiset of (key, count) records (rfx_records_checksum summed over the shards);
    4. a sampled block of the subject counted alone with lower = 1: sum(i * histo[i]) == the number of ACGT-only
       windows, computed on the host from the generator's host twin (text), not from the packed block."""
    k = trio.k
    out = {}
    keys0 = np.asarray(res["mutant_keys"], dtype=np.uint64)
    rv = trio.run(samples, verify=True, probe_keys=keys0)
    v = rv["verify"]
    assert v["bad_order"] == 0 and v["bad_pos"] == 0 and v["bad_count"] == 0, f"records fail their invariants: {v}"
    assert rv["n_records"] == res["n_records"] and rv["n_pulled"] == res["n_pulled"]
    assert np.array_equal(rv["mutant_keys"], keys0)
    for si, h in enumerate(rv["histos"]):
        assert np.array_equal(h, res["histos"][si])
        assert int(h[-1]) != 0 or int(sum(int(x) * i for i, x in enumerate(h))) == v["sum_counts"][si], "histogram != records"
        assert int(h.sum()) == rv["n_records"][si]
    assert v["probe_found"][0] == len(keys0) and v["probe_count_out_of_range"] == 0, "a mutant k-mer is not the subject's"
    assert all(x == 0 for x in v["probe_found"][1:]), "a mutant k-mer occurs in a control"
    out.update(records_verified=int(sum(rv["n_records"])), order_pos_count_violations=0,
               mutant_in_subject=int(v["probe_found"][0]), mutant_in_controls=int(sum(v["probe_found"][1:])))
    from .tools import keys_to_text
    if sys_[0].n_snv and len(samples) > 1:
        expect = expected_snv_kmers(sys_[0], k)
        got = set(x.encode() for x in keys_to_text(keys0, k))
        assert len(got) == len(keys0)
        extra = got - expect
        out.update(snv_kmers_expected=len(expect), snv_kmers_found=len(got & expect), not_snv_kmers=len(extra))
        # k-mers that are the subject's alone without being an SNV's: sites where >= 5 of the c reads that cover a base
        # carry the SAME substitution -- G * 3 * C(c, 5) * (e / 3)^5 of them (17 at 30x, 660 at 60x for 3.1 Gb and
        # e = 0.5 %), up to k k-mers each; allow three times that plus 30 sites
        import math
        cov = max(5, int(round(n_pairs * 2 * sys_[0].read_len / sys_[0].genome_len)))
        e3 = sys_[0].err_1024 / 1024.0 / 3.0
        sites = sys_[0].genome_len * 3.0 * math.comb(cov, 5) * e3 ** 5
        out["not_snv_kmers_allowed"] = int(k * (3 * sites + 30))
        assert len(extra) <= out["not_snv_kmers_allowed"], f"{len(extra)} mutant k-mers are no SNV k-mers"
        if n_pairs * 300 >= 20 * sys_[0].genome_len:      # at >= 20x nearly every SNV k-mer reaches MinCov
            assert len(got & expect) >= 0.9 * len(expect), f"only {len(got & expect)} of {len(expect)} SNV k-mers found"
    if more_passes and trio.world == 1:
        t2 = WgsTrio(ctx, k, trio.size, trio.lower, trio.min_cov, trio.max_cov, trio.thresh, passes=trio.passes + 1)
        r2 = t2.run(samples, verify=True)
        assert r2["n_records"] == res["n_records"] and r2["n_pulled"] == res["n_pulled"]
        assert np.array_equal(r2["mutant_keys"], keys0)
        assert all(np.array_equal(a, b) for a, b in zip(r2["histos"], res["histos"]))
        # the same (key, count) pairs, every one of them, from two different cuts of the work (rfx_records_checksum)
        assert r2["verify"]["checksum"] == v["checksum"], "S and S + 1 shard passes do not hold the same records"
        out["passes_compared"] = [trio.passes, t2.passes]
        out["multiset_checksums"] = ["%016x" % c[0] for c in v["checksum"]]
    if sample_pairs:
        n = int(min(sample_pairs, n_pairs))
        first = (n_pairs - n) // 3
        seq, _ = sys_[0].text(first, n)
        want = valid_windows_of_text(seq, k)
        blk = ctx.synth_reads(sys_[0], first, n, min_q, False, True)
        t = capi.CountTable(ctx, k, trio.size)
        try:
            t.add(blk)
            rec, h = t.finish(1, want_histo=True)
            got_w = int(sum(int(x) * i for i, x in enumerate(h)))
            assert int(h[-1]) == 0 and got_w == want, f"sampled block: {got_w} k-mer instances counted, {want} valid windows"
            rec.free()
        finally:
            t.free()
            blk.free()
        out["sampled_block_windows"] = want
    return out



def make_sample(ctx: capi.Context, sy: capi.Synth, n_pairs: int, block_pairs: int = 1 << 24, min_q: int = 15,
                want_good: bool = True, first_pair: int = 0, compact: bool = False):
    """Blocks of a synthetic sample, generated on the device (pairs first_pair .. first_pair + n_pairs)."""
    blocks, p = [], first_pair
    while p < first_pair + n_pairs:
        n = min(block_pairs, first_pair + n_pairs - p)
        blocks.append(ctx.synth_reads(sy, p, n, min_q, want_good, compact))
        p += n
    return blocks
