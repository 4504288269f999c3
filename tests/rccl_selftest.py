"""One-rank RCCL group on the one GPU: every collective of the minimizer-shard path is issued for real
(all_to_all_single with split sizes, async variant, all_reduce, all_gather) and the result must be the
single-GPU count.  Multi-rank RCCL cannot be tried on a 1-GPU box; this at least exercises the API use."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import socket
_s = socket.socket()
_s.bind(("127.0.0.1", 0))
_port = _s.getsockname()[1]
_s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_port), RANK="0", WORLD_SIZE="1")
import numpy as np, torch, torch.distributed as dist
import oracle
from rufus_amd import capi, dist as rdist
from tests.synth import make_trio, flat_reads
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
ctx = capi.Context(0)
trio = make_trio(genome_len=40_000, n_pairs=3000, n_snv=5, seed=8)
seq, qual, off = flat_reads(trio["child"])
blk = ctx.upload(capi.PackedReads(seq, off, qual, 15, capi.PACK_COUNT | capi.PACK_FILTER))
be = rdist.HipBackend(ctx, 25, 8 << 30)
rec, bs, keep, ext = be.partition(blk)
runs = rdist.exchange_records(rec, bs, dist.group.WORLD, ext)
keep.free()
out, histo = be.count_records(runs, 2)
reads = [x.tobytes() for m in (0, 1) for x in trio["child"].s[m]]
ref = oracle.count(None, 25, 8 << 30, lower=2, reads=reads)
assert out.payload() == ref.payload()
h = torch.from_numpy(histo.astype(np.int64)).cuda()
dist.all_reduce(h)
assert np.array_equal(h.cpu().numpy().astype(np.uint64), oracle.histo(ref.counts, full=True)[0])
keys = rdist.all_gather_keys(ref.keys[:100], torch.device("cuda", 0), dist.group.WORLD)
assert np.array_equal(keys, ref.keys[:100])
# the pos-sharded exchange too
k_, c_, p_ = be.count_partials(blk)
rk, rc = rdist.exchange_partials(k_, c_, p_, be.lsize, dist.group.WORLD)
r2, _ = be.reduce_partials(rk, rc, 2, 0, 1 << be.lsize)
assert r2.payload() == ref.payload()
# the WGS driver's exchange (multi-block samples, shard passes, flat owner cut) over the real RCCL backend
os.environ["RFX_WGS_FORCE_EXCHANGE"] = "1"
from rufus_amd import wgs
sy = capi.Synth.sample(150_000, 0, n_snv=6, seed=5)
blocks = wgs.make_sample(ctx, sy, 12_000, 5000, 15, True)
tr = wgs.WgsTrio(ctx, 25, 8 << 30, 2, 5, 1200, 1, passes=2, group=dist.group.WORLD)
seqs, _ = sy.text(0, 12_000)
ref2 = oracle.count(None, 25, 8 << 30, lower=2, reads=[r.tobytes() for r in seqs])
shards = [tr.count_shard(blocks, sh)[0] for sh in range(2)]
got = [s_.get() for s_ in shards]
k2, c2, p2 = (np.concatenate([g[i] for g in got]) for i in range(3))
o = np.lexsort((k2, p2))
assert np.array_equal(k2[o], ref2.keys) and np.array_equal(c2[o].astype(np.uint64), ref2.counts)
print("rccl self-test ok:", len(ref.keys), "records;", len(ref2.keys), "through the WGS exchange")
dist.destroy_process_group()
