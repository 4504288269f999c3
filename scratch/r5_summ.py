"""One line per bench.py JSON line on stdin (round 5 A/B scripts): headline, chain, per-kernel ms, memory."""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else ""
for ln in sys.stdin:
    ln = ln.strip()
    if not ln.startswith("{"):
        continue
    d = json.loads(ln)
    r, c = d["roofline"], d["config"]
    ks = r["avg_launch_ms_by_kernel"]
    ln_ = r["launches_by_kernel_per_chain"]
    print("%-14s %.1f M reads/s  %.1f ms/step  chain %.1f ms (frac %.4f)  filter %.1f ms" % (
        tag, d["value"] / 1e6, d["ms_per_step"], r["avg_launch_ms"], r["frac"],
        (d.get("roofline_filter") or {}).get("ms_per_step", 0.0)))
    print("   " + "  ".join("%s %.1f (%.0f)" % (k.replace("k_", ""), v, ln_.get(k, 0)) for k, v in ks.items()))
    print("   passes %s  map budget %s  early %s  replayed %s  peak %.1f GB  mapped %.1f GB  checked %s  checksums %s" % (
        c.get("passes"), c.get("run_map_budget_bytes"), c.get("early_cut_budget_bytes"),
        c.get("blocks_replayed_from_run_maps_per_step"), c["hbm_peak_bytes"] / 1e9, c["hbm_mapped_bytes"] / 1e9,
        c.get("checked"), (c.get("checks") or {}).get("multiset_checksums")))
