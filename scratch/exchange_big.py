"""The WGS driver's record exchange over a real (one-rank) RCCL group at a size where the buffers are GBs:
30x trio of a 1 Gb genome, every segment goes through all_to_all_single and the import path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", RANK="0", WORLD_SIZE="1", RFX_WGS_FORCE_EXCHANGE="1")
import numpy as np, torch, torch.distributed as dist
from rufus_amd import capi, wgs
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
ctx = capi.Context(0)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n_pairs = G // 10
K = int(sys.argv[3]) if len(sys.argv) > 3 else 25
sys_ = [capi.Synth.sample(G, w, n_snv=max(8, G // 3_100_000), seed=12345) for w in range(3)]
samples = [wgs.make_sample(ctx, sy, n_pairs, 1 << 24, 15, want_good=(i == 0), compact=True) for i, sy in enumerate(sys_)]
res = {}
modes = {"forced": (True, True), "local": (False, False)}.get(os.environ.get("MODES", ""), (True, True, False, False))
for forced in modes:     # the first run of each kind also grows the arena / torch's cache: warm-up
    if not forced:
        os.environ.pop("RFX_WGS_FORCE_EXCHANGE", None)
    trio = wgs.WgsTrio(ctx, K, 8 << 30, 2, 5, 1200, 1, passes=passes, group=dist.group.WORLD)
    t0 = time.perf_counter()
    r = trio.run(samples)
    ctx.sync()
    res[forced] = r
    print(f"forced exchange={forced}: {time.perf_counter() - t0:.2f} s, {r['n_mutant']} mutant k-mers, {r['n_pulled']} pairs, records {r['n_records']}", flush=True)
if len(set(modes)) < 2:
    dist.destroy_process_group()
    sys.exit(0)
assert np.array_equal(res[True]["mutant_keys"], res[False]["mutant_keys"]) and res[True]["n_records"] == res[False]["n_records"]
assert all(np.array_equal(a, b) for a, b in zip(res[True]["histos"], res[False]["histos"]))
print("exchange path == local path")
dist.destroy_process_group()
