"""CPU, world_size 2 and 3, gloo: the exchange logic of rufus_amd.dist driven with a checker backend --
both sharding schemes: by pos (all-to-all of (key,count) partials, owner reduce; owner slices concatenate
to the single-process result) and by minimizer bin (all-to-all of records, owners count complete bins;
shards interleave to the single-process result).  Histogram all-reduce, mutant-set all-gather, filter."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from rufus_amd import dist as rdist
from tests.synth import make_trio

K, SIZE, LOWER, MIN_COV, MAX_COV, THRESH, MINQ = 25, 1 << 27, 2, 5, 1200, 1, 15


class Block:
    def __init__(self, sample):
        self.seqs = [r.tobytes() for m in (0, 1) for r in sample.s[m]]
        self.quals = [r.tobytes() for m in (0, 1) for r in sample.q[m]]
        self.n = len(self.seqs)


_CODE = np.full(256, 4, dtype=np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _CODE[_c] = _i
_POW = (np.uint64(1) << (np.uint64(2) * np.arange(K - 1, -1, -1, dtype=np.uint64)))


def kmer_instances(seqs, k=K):
    """Canonical key of every window without a non-ACGT character (what jellyfish -C counts)."""
    out = []
    for s in seqs:
        c = _CODE[np.frombuffer(s, dtype=np.uint8)]
        if len(c) < k:
            continue
        w = np.lib.stride_tricks.sliding_window_view(c, k)
        ok = (w < 4).all(axis=1)
        w = w[ok].astype(np.uint64)
        fwd = (w * _POW).sum(axis=1, dtype=np.uint64)
        rc = ((np.uint64(3) - w[:, ::-1]) * _POW).sum(axis=1, dtype=np.uint64)
        out.append(np.minimum(fwd, rc))
    return np.concatenate(out) if out else np.zeros(0, dtype=np.uint64)


class OracleBackend:
    """Stands in for HipBackend on CPU: same interface, oracle arithmetic."""
    device = torch.device("cpu")

    def __init__(self):
        self.lsize = oracle.ceil_log2(SIZE)
        self.cols = oracle.jf_matrix(self.lsize, K)

    def local_count(self, block, lower):
        r = oracle.count(None, K, SIZE, lower=lower, reads=block.seqs)
        return r, oracle.histo(r.counts, full=True)[0]

    def count_partials(self, block):
        r, _ = self.local_count(block, 1)
        return (torch.from_numpy(r.keys.view(np.int64).copy()), torch.from_numpy(r.counts.astype(np.int32)),
                torch.from_numpy(r.pos.view(np.int64).copy()))

    def reduce_partials(self, keys, counts, lower, pos_lo, pos_hi):
        k = keys.numpy().view(np.uint64)
        uk, inv = np.unique(k, return_inverse=True)
        c = np.zeros(len(uk), dtype=np.uint64)
        np.add.at(c, inv, counts.numpy().astype(np.uint64))
        pos = np.array([oracle.jf_pos(self.cols, int(x), self.lsize) for x in uk], dtype=np.uint64)
        assert np.all((pos >= pos_lo) & (pos < pos_hi)), "partial routed to the wrong owner"
        keep = c >= lower
        uk, c, pos = uk[keep], c[keep], pos[keep]
        o = np.lexsort((uk, pos))
        r = oracle.Records(K, self.lsize, self.cols, uk[o], c[o], pos[o])
        return r, oracle.histo(r.counts, full=True)[0]

    # minimizer-shard interface: here a "record" is one canonical k-mer instance and its bin any fixed
    # function of the k-mer -- enough to exercise the exchange (splits, offsets, import) on CPU
    BINS = 512

    def msp_capable(self):
        return True

    def _bin(self, keys):
        return ((keys * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(64 - 9)).astype(np.int64)

    def partition(self, block):
        keys = kmer_instances(block.seqs)                    # canonical key of every valid window
        b = self._bin(keys)
        o = np.argsort(b, kind="stable")
        bs = np.zeros(self.BINS + 1, dtype=np.int64)
        np.cumsum(np.bincount(b, minlength=self.BINS), out=bs[1:])
        return torch.from_numpy(keys[o].view(np.int64).copy()), torch.from_numpy(bs)

    def count_records(self, runs, lower):
        me, world = dist.get_rank(), dist.get_world_size()
        own = rdist.bin_owner_bounds(self.BINS, world)
        chunks = []
        for rec, bs in runs:
            bs = bs.numpy()
            assert len(bs) == self.BINS + 1 and bs[own[me]] == 0 and bs[own[me + 1]] == len(rec) == bs[-1]
            k = rec.numpy().view(np.uint64)
            assert np.all(np.repeat(np.arange(self.BINS), np.diff(bs)) == self._bin(k)), "record in the wrong bin"
            chunks.append(k)
        k = np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.uint64)
        uk, c = np.unique(k, return_counts=True)
        assert np.all((self._bin(uk) >= own[me]) & (self._bin(uk) < own[me + 1]))
        c = c.astype(np.uint64)
        keep = c >= lower
        uk, c = uk[keep], c[keep]
        pos = self.pos_of(uk)
        o = np.lexsort((uk, pos))
        r = oracle.Records(K, self.lsize, self.cols, uk[o], c[o], pos[o])
        return r, oracle.histo(r.counts, full=True)[0]

    def pos_of(self, keys):
        return np.array([oracle.jf_pos(self.cols, int(x), self.lsize) for x in keys], dtype=np.uint64)

    def unique(self, subject, others, min_cov, max_cov):
        keys, vals, which = oracle.merge_unique([subject] + list(others), with_file=True)
        sel = [(k, v) for k, v, f in zip(keys, vals, which) if f == 0 and min_cov <= v <= max_cov]
        return (np.array([k for k, _ in sel], dtype=np.uint64), np.array([v for _, v in sel], dtype=np.uint32))

    def filter_pairs(self, canon_keys, block, thresh):
        text = "".join(f"{oracle.jf_decode(int(k), K)} 1\n" for k in canon_keys).encode()
        fs = oracle.FilterSet(text)
        hit = np.array([fs.scan(s, q, K, MINQ) >= thresh for s, q in zip(block.seqs, block.quals)])
        half = block.n // 2
        return hit[:half] | hit[half:2 * half]

    def free(self, rec):
        pass

    def n_records(self, rec):
        return len(rec.keys)


def _worker(rank, world, port, q, shard_by):
    # pieces of 4 KB: every (source, destination) share of the record exchange takes many isend / irecv rounds
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RFX_WGS_A2A_MAX_BYTES="4096")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        trio = make_trio(genome_len=30_000, n_pairs=1500, n_snv=4, seed=5, read_seed=100 + rank)
        blocks = {n: Block(trio[n]) for n in ("child", "mother", "father")}
        shard = rdist.TrioShard(OracleBackend(), K, SIZE, LOWER, MIN_COV, MAX_COV, THRESH, group=dist.group.WORLD,
                                shard_by=shard_by)
        res = shard.run(blocks["child"], [blocks["mother"], blocks["father"]], keep_records=True)
        q.put((rank, [(r.keys.tolist(), r.counts.tolist(), r.pos.tolist()) for r in res["records"]],
               [h.tolist() for h in res["histos"]],
               res["mutant_keys"].tolist(), res["pulled"].tolist(), res["n_records"], res["n_pulled"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shard_by", [(2, "pos"), (2, "minimizer"), (3, "minimizer")])
def test_exchange_matches_single_process(world, shard_by):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, shard_by)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0

    # single-process truth on the union of both ranks' reads
    be = OracleBackend()
    allb = {}
    for n in ("child", "mother", "father"):
        parts = [Block(make_trio(genome_len=30_000, n_pairs=1500, n_snv=4, seed=5, read_seed=100 + r)[n])
                 for r in range(world)]
        allb[n] = parts
    recs = [oracle.count(None, K, SIZE, lower=LOWER, reads=sum((b.seqs for b in allb[n]), []))
            for n in ("child", "mother", "father")]
    for i in range(3):
        shards = [tuple(np.array(x, dtype=np.uint64) for x in g[1][i]) for g in got]
        if shard_by == "pos":      # owner slices concatenate to the file
            keys_, counts_, pos_ = (np.concatenate([s_[j] for s_ in shards]) for j in range(3))
        else:                      # minimizer shards are disjoint and interleave to the file
            assert all(len(s_[0]) for s_ in shards)
            keys_, counts_, pos_ = rdist.merge_shards(shards)
        assert keys_.tolist() == recs[i].keys.tolist() and counts_.tolist() == recs[i].counts.tolist()
        assert pos_.tolist() == recs[i].pos.tolist()
        assert all(g[2][i] == oracle.histo(recs[i].counts, full=True)[0].tolist() for g in got)
    keys, _ = be.unique(recs[0], recs[1:], MIN_COV, MAX_COV)
    assert all(g[3] == keys.tolist() for g in got) and len(keys) > 0
    assert all(g[5] == [len(r.keys) for r in recs] for g in got)
    n_pulled = 0
    for r in range(world):
        want = be.filter_pairs(keys, allb["child"][r], THRESH)
        assert got[r][4] == want.tolist()
        n_pulled += int(want.sum())
    assert all(g[6] == n_pulled for g in got) and n_pulled > 0


def test_bin_owner_bounds_agree_across_bin_counts():
    """Owner ranges are cut on the top 8 bits of the bin index, so a bin of a 256-bin partition and its
    refinements in 1024- or 8192-bin partitions have the same owner, for any world size."""
    for world in (1, 2, 3, 5, 8, 64, 256):
        owners = {}
        for bins in (256, 1024, 8192):
            b = rdist.bin_owner_bounds(bins, world)
            assert b[0] == 0 and b[-1] == bins and all(x <= y for x, y in zip(b, b[1:]))
            own = np.searchsorted(np.array(b[1:]), np.arange(bins), side="right")
            owners[bins] = own[::bins // 256]                     # owner of each virtual (top-8-bit) bin
            assert np.array_equal(np.repeat(owners[bins], bins // 256), own)   # refinements stay together
        assert np.array_equal(owners[256], owners[1024]) and np.array_equal(owners[256], owners[8192])
        if world <= 256:
            assert len(set(owners[256].tolist())) == world        # every rank owns something


def test_merge_shards_interleaves_by_pos_then_key():
    a = (np.array([5, 9], np.uint64), np.array([1, 2], np.uint64), np.array([0, 7], np.uint64))
    b = (np.array([3, 4, 1], np.uint64), np.array([3, 4, 5], np.uint64), np.array([0, 7, 9], np.uint64))
    k, c, p = rdist.merge_shards([a, b])
    assert k.tolist() == [3, 5, 4, 9, 1] and c.tolist() == [3, 1, 4, 2, 5] and p.tolist() == [0, 0, 7, 7, 9]


def test_owner_bounds_cover_the_range():
    for lsize in (10, 27, 33):
        for world in (1, 2, 3, 8):
            b = rdist.owner_bounds(lsize, world)
            assert b[0] == 0 and b[-1] == 1 << lsize and all(x < y for x, y in zip(b, b[1:]))
    k = np.array([0, 1, 0x3FFFFFFFFFFFF, 12345678901234], dtype=np.uint64)
    assert rdist.revcomp_keys(rdist.revcomp_keys(k, 25), 25).tolist() == k.tolist()
    assert rdist.revcomp_keys(np.array([0], dtype=np.uint64), 3).tolist() == [63]


# ------------------------------------------------------------------------------------------------
# the checkpoints of the WGS driver: a failure on one rank is raised on every rank, at the same place
# ------------------------------------------------------------------------------------------------
def _checkpoint_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rufus_amd import capi, wgs
        trio = object.__new__(wgs.WgsTrio)          # the protocol needs no device: only the group
        trio.world, trio.rank, trio.group = world, rank, dist.group.WORLD
        seen = []
        trio.checkpoint(None)                       # everyone fine: returns
        seen.append("fine")
        for failing, exc in ((1, capi.RufusError("rfx_count_add: out of device memory")),
                             (0, torch.OutOfMemoryError("HIP out of memory")),
                             (world - 1, ValueError("something else")),
                             (None, None)):
            try:
                trio.checkpoint(exc if rank == failing else None)
                seen.append("fine")
            except wgs.GroupFailure as e:
                seen.append(("retry" if e.retry else "fatal", e.__cause__ is not None, "memory" in str(e)))
        # two ranks fail differently in the same step: the worse one decides
        try:
            trio.checkpoint(capi.RufusError("out of memory") if rank == 0 else ValueError("x") if rank == 1 else None)
        except wgs.GroupFailure as e:
            seen.append("retry" if e.retry else "fatal")
        q.put((rank, seen))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_a_failure_on_one_rank_is_raised_on_all_of_them(world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_checkpoint_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, seen in got.items():
        assert seen[0] == "fine"
        assert seen[1] == ("retry", rank == 1, True)             # only the failing rank chains its own exception
        assert seen[2] == ("retry", rank == 0, True)
        assert seen[3] == ("fatal", rank == world - 1, False)
        assert seen[4] == "fine" and seen[5] == "fatal"
