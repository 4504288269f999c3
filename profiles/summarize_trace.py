#!/usr/bin/env python3
"""Per-kernel statistics of ONE step of a `rocprofv3 --kernel-trace` run of bench.py.

    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python bench.py --inner --steps 1 --warmup 1 --no-check
    python profiles/summarize_trace.py DIR/.../t_kernel_trace.csv profiles/rNN_kernel_stats_wgs.csv [reads-filtered blocks per step]

`--stats` sums over the whole process: the warm-up step (which runs without run maps, so with other kernels and launch
counts) and, in round 5, a step that was partially repeated after an out-of-memory retry (VERDICT r5 "What's weak" #5: 255
k_msp_leaf launches where two steps make 204).  Here the launches are cut into steps where the trace says a step ends -- a
step of the W / TN workloads ends with the read filter over the subject's blocks, the last kernel of which is k_hits_mask / k_mask_count
-- and ONLY THE LAST STEP (the one bench.py would time) is summarised, in --stats' own CSV layout.  A sidecar
`<out>.meta.json` says how many steps the trace held, which one was taken and how many launches it has, and
tests/test_evidence_host.py holds the launch counts against `launches_by_kernel_per_chain` of the round's bench line.
"""
import collections
import csv
import json
import statistics
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    rows = []
    for r in csv.DictReader(open(src)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # step boundaries: a step of the W / TN workloads ends with the read filter over the subject's blocks (k_filter_* +
    # k_hits_mask or k_mask_count per block, the runtime's fill / copy kernels between them): the last filter-family
    # launch before a kernel of ours that is NOT of that family (or before the end of the trace)
    def is_filter(n):
        return "k_filter" in n or "k_hits_mask" in n or "k_mask_count" in n

    def ours(n):
        return "k_" in n and "__amd_rocclr" not in n
    ends = []
    for i in range(len(rows)):
        if not is_filter(rows[i][2]):
            continue
        nxt = next((rows[j][2] for j in range(i + 1, len(rows)) if ours(rows[j][2])), None)
        if nxt is None or not is_filter(nxt):
            ends.append(i)
    if not ends:
        raise SystemExit("no read-filter launch in the trace: not a W / TN step")
    # the first launch of the last step: the one after the previous step's end (or after the input was synthesised)
    last_end = ends[-1]
    first = ends[-2] + 1 if len(ends) > 1 else 0
    if len(ends) == 1:
        synth = [i for i in range(last_end) if "k_synth_reads" in rows[i][2]]
        first = synth[-1] + 1 if synth else 0
    step = rows[first:last_end + 1]
    agg = collections.defaultdict(list)
    for s, e, n in step:
        agg[n].append(e - s)
    total = sum(sum(v) for v in agg.values())
    with open(dst, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([n, len(v), sum(v), round(sum(v) / len(v), 6), round(100.0 * sum(v) / total, 2), min(v), max(v),
                        round(statistics.pstdev(v), 6)])
    meta = {"source": "rocprofv3 --kernel-trace (no --stats): launches of the LAST step only, cut where the trace shows "
                      "the step's read filter ending (profiles/summarize_trace.py)",
            "steps_in_trace": len(ends), "step_taken": len(ends) - 1, "launches_in_step": len(step),
            "step_wall_ms": (step[-1][1] - step[0][0]) / 1e6, "kernel_ms": total / 1e6,
            "note": "step_wall_ms is the wall time of the traced step UNDER the tracer, and with --warmup 1 that step is the "
                    "second of its process -- the one in which the arena grows (hipMemCreate / hipMemMap of tens of GB between "
                    "two runtime copy kernels: the large idle gaps); bench.py's timed steps come after its warm-ups and are "
                    "busy to within 1 % (ms_per_step of the bench line vs kernel_ms here)"}
    json.dump(meta, open(dst + ".meta.json", "w"), indent=1)
    print(json.dumps(meta))


if __name__ == "__main__":
    main()
