"""Host latency of the small operations one exchange round is made of (one-rank RCCL group)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29519", RANK="0", WORLD_SIZE="1")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
g = dist.group.WORLD

def timeit(name, fn, n=50):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms", flush=True)

big = torch.zeros(1 << 27, dtype=torch.int64, device=dev)   # 1 GiB
small = torch.zeros(32769, dtype=torch.int64, device=dev)
timeit("small .cpu()", lambda: small.cpu())
timeit("torch.tensor(list, device)", lambda: torch.tensor([[1, 2]], dtype=torch.int64, device=dev).flatten())
def a2a_small():
    s = torch.tensor([5, 32768], dtype=torch.int64, device=dev)
    r = torch.empty_like(s)
    dist.all_to_all_single(r, s, group=g)
    return r.tolist()
timeit("all_to_all_single(2 x int64) + tolist", a2a_small)
def a2a_off():
    sb = small.clone()
    rb = torch.empty_like(sb)
    dist.all_to_all_single(rb, sb, [32769], [32769], group=g)
    return rb
timeit("all_to_all_single(32769 x int64, split lists)", a2a_off)
timeit("all_reduce MIN (checkpoint)", lambda: dist.all_reduce(torch.tensor([2], dtype=torch.int64, device=dev), op=dist.ReduceOp.MIN, group=g))
timeit("torch.empty(1 GiB) + free", lambda: torch.empty(1 << 27, dtype=torch.int64, device=dev))
out = torch.empty_like(big)
timeit("copy_ 1 GiB", lambda: out.copy_(big))
timeit("current_stream.synchronize", lambda: torch.cuda.current_stream().synchronize())
timeit("zeros(32769).to(dev, non_blocking)", lambda: torch.zeros(32769, dtype=torch.int64).to(dev, non_blocking=True))
dist.destroy_process_group()
