"""Phase breakdown of k_msp_part1 / k_msp_leaf (library built with -DRFX_TIMING): wave-0 cycles per phase, summed
over workgroups, for one step of the 30x trio on a --genome slice.  usage: timing_probe.py [genome] [lib.so]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rufus_amd import capi, wgs
G = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
lib = capi.lib()
ctx = capi.Context(0)
pairs = G * 30 // 300
sys_ = [capi.Synth.sample(G, w, n_snv=max(20, min(1000, G // 3_000_000)), seed=12345) for w in range(3)]
samples = [wgs.make_sample(ctx, sy, pairs, 1 << 24, 15, want_good=(i == 0)) for i, sy in enumerate(sys_)]
ctx.sync()
passes = int(os.environ.get("PASSES", "1"))
K = int(os.environ.get("K", "25"))
trio = wgs.WgsTrio(ctx, K, 8 << 30, 2, 5, 1200, 1, passes=passes)
trio.run(samples)                       # warm-up
buf = (C.c_ulonglong * 32)()
lib.rfx_debug_timing.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
lib.rfx_debug_timing(buf, 1)
import time
t0 = time.perf_counter(); out = trio.run(samples); ctx.sync(); dt = time.perf_counter() - t0
lib.rfx_debug_timing(buf, 0)
v = list(buf)
print(f"step {dt*1e3:.0f} ms, mutant {out['n_mutant']}, records {out['n_records']}")
p1 = {"main loop (hash+min+close)": v[0], "reserve + barriers": v[1], "stores": v[2], "loop top": v[3]}
lf = {"init": v[8], "phase A (record cache)": v[9], "pack": v[10], "phase B (k-mer table)": v[11], "scan": v[12], "flush": v[13], "between rounds": v[15]}
for name, d in (("k_msp_part1", p1), ("k_msp_leaf", lf)):
    tot = sum(d.values()) or 1
    print(name)
    for k_, x in d.items():
        print(f"   {k_:32s} {x:16d}  {100.0 * x / tot:5.1f} %")
tot = sum(lf.values()) + v[16] + v[17] + v[18] or 1
print("   wave 0: A %.1f %% + wait %.1f %%, B %.1f %% + wait %.1f %%, C %.1f %% + wait %.1f %%, flush %.1f %%" % tuple(
    100.0 * x / tot for x in (v[16], v[9], v[17], v[11], v[18], v[12], v[13])))
print("leaf rounds", v[21], "overflowed", v[20], "records", v[22], "uncached", v[23], "mixed", v[24])
print("flush: n %d; lut %.1f, pass 1 %.1f + wait %.1f, cursors %.1f, pass 2 %.1f + wait %.1f (%% of wave 0's time)" % ((v[31],) + tuple(100.0 * v[i] / tot for i in (25, 26, 27, 28, 29, 30))))
print("   cycles (100 MHz ticks) per flush: " + ", ".join("%.0f" % (v[i] / max(1, v[31])) for i in (25, 26, 27, 28, 29, 30)), "; per bin round:", "%.0f" % (tot / max(1, v[21])))
