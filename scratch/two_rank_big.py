"""Two ranks on ONE GPU (gloo carries the exchange through host memory) on a 30x trio big enough for the big-block
path (32768 first-level bins, several segments per sample, exchange pieces): record counts, histograms, hash list
and pulled pairs must equal the single-rank run.  usage: two_rank_big.py [genome=200000000] [passes=2] [k=25]"""
import os, sys, socket, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def worker(rank, world, port, q, G, passes, k):
    import torch, torch.distributed as dist
    from rufus_amd import capi, wgs
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    c = capi.Context(0)
    n_pairs = G // 10
    sys_ = [capi.Synth.sample(G, w, n_snv=max(8, G // 3_100_000), seed=12345) for w in range(3)]
    p0, p1 = n_pairs * rank // world, n_pairs * (rank + 1) // world
    samples = [wgs.make_sample(c, sy, p1 - p0, 1 << 24, 15, want_good=(i == 0), first_pair=p0) for i, sy in enumerate(sys_)]
    trio = wgs.WgsTrio(c, k, 8 << 30, 2, 5, 1200, 1, passes=passes, group=dist.group.WORLD if world > 1 else None)
    t0 = time.perf_counter()
    res = trio.run(samples)
    c.sync()
    q.put((rank, time.perf_counter() - t0, res["n_records"], [h.tolist() for h in res["histos"]], res["mutant_keys"].tolist(),
           res["n_pulled"]))
    c.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
    passes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    out = {}
    for world in (1, 2):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=worker, args=(r, world, port, q, G, passes, k)) for r in range(world)]
        for p in procs: p.start()
        got = sorted(q.get(timeout=1500) for _ in range(world))
        for p in procs:
            p.join(120); assert p.exitcode == 0
        out[world] = got
        print(f"world {world}: {[round(g[1], 2) for g in got]} s, records {got[0][2]}, {len(got[0][4])} mutant k-mers, {got[0][5]} pairs", flush=True)
    one, two = out[1][0], out[2]
    for g in two:
        assert g[2] == one[2] and g[3] == one[3] and g[4] == one[4] and g[5] == one[5]
    print("two ranks == one rank")
