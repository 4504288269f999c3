import sys, time
sys.path.insert(0, ".")
import torch
import bench
from rufus_amd import capi
class A: pass
ctx = capi.Context(0)
step, *_ = bench.run_s1(A(), ctx, 0, 1, None, torch)
ctx.prof_filter(bench.K2_CHAIN + ("k_filter",))
out = []
for i in range(12):
    ctx.prof(i in (0, 1, 2, 6, 7, 11))
    if i == 5: ctx.prof_reset()
    t = time.perf_counter(); step(); out.append(round((time.perf_counter() - t) * 1e3, 2))
print("steps (prof on at 0,1,2,6,7,11; reset before 5):", out)
