import cProfile, pstats, sys, time, io
sys.path.insert(0, ".")
import torch
import bench
from rufus_amd import capi
class A: pass
args = A()
ctx = capi.Context(0)
step, *_ = bench.run_s1(args, ctx, 0, 1, None, torch)
for _ in range(5): step()
ctx.sync()
t0 = time.perf_counter()
for _ in range(20): step()
ctx.sync()
print("ms/step", (time.perf_counter() - t0) / 20 * 1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(20): step()
ctx.sync()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:3500])
