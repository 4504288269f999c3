import os, sys
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29519", RANK="0", WORLD_SIZE="1")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
for n in (1_000_000, 8_000_000, 16_000_000, 33_000_000, 64_000_000, 67_108_864, 70_000_000, 134_217_728, 200_000_000):
    src = torch.arange(n, dtype=torch.int64, device="cuda")
    dst = torch.zeros(n, dtype=torch.int64, device="cuda")
    if os.environ.get("A2A_MODE") == "nosplit":
        dist.all_to_all_single(dst, src)
    elif os.environ.get("A2A_MODE") == "sendrecv":
        ops = [dist.P2POp(dist.isend, src, 0), dist.P2POp(dist.irecv, dst, 0)]
        for r in dist.batch_isend_irecv(ops): r.wait()
    else:
        dist.all_to_all_single(dst, src, [n], [n])
    torch.cuda.synchronize()
    bad = int((dst != src).sum().item())
    first = int(torch.nonzero(dst != src)[0].item()) if bad else -1
    print(f"n={n} ({n*8/2**30:.1f} GiB): mismatches {bad}, first at {first} (byte {first*8 if bad else 0})", flush=True)
    del src, dst
dist.destroy_process_group()
