#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a
pass on gfx950: TCC has 4 slots, FETCH_SIZE takes 3 and WRITE_SIZE 2 -- MI355X_MICROARCH.md).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o f -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o w -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline
    python profiles/summarize_pmc.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv profiles/r01_pmc.json

Units/corrections as the guide prescribes: the counters are in KB (bytes = value * 1024) and on gfx950
FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read, so
hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  (Calibration inside this profile: k_part2 and
k_leaf each stream the 1 008 000 000-byte word array once; FETCH_SIZE reads 483 582 KB / 481 417 KB.)
"""
import collections
import csv
import json
import re
import sys


# kernels of the profiled bench command that are NOT part of a sample's count chain: the synthetic input, the set
# difference (K4), the read filter (K5) and its set, plain copies
NOT_CHAIN = frozenset((
    "k_synth_reads", "k_copy16",
    "k_flag_absent", "k_flag_absent_tiled", "k_fa_bounds", "k_flag_gather", "k_flag_range", "k_flag_rank", "k_flag_present",
    "k_compact_count", "k_compact_scatter", "k_scan_u64", "k_query",
    "k_filter", "k_filter_q", "k_filter_p", "k_filter_fast", "k_filter_big", "k_hits_mask", "k_mask_count", "k_set_bitmap",
    "k_set_bitmap_big", "k_set_bitmap_q", "k_set_bitmap_p",
    "k_set_insert", "k_set_bitmap_packed", "k_records_checksum", "k_records_verify", "k_check_sorted",
    # never launched by the bench's device step: record I/O of the executables, the assembly stage, ModelDist
    "k_parse_records", "k_format_records", "k_txt_count", "k_txt_lines", "k_txt_reads", "k_txt_pack", "k_compute_pos", "k_annotate", "k_overlap_pool", "k_overlap_score",
    "k_model_colsum", "k_model_dist", "k_model_rowtot", "k_model_sum", "k_model_terms", "k_model_weights",
))
# the chain's kernels as rocprofv3 names them (bench.K2_CHAIN holds the LABELS of the library's HIP-event brackets, several
# of which cover more than one kernel: "k_bin_hist" = k_bin_hist / k_bin_hist_stamp / k_bin_hist_multi + its scans,
# "k_part3" = k_part2 instantiated for the refinement)
CHAIN_KNOWN = frozenset((
    "k_msp_part1", "k_msp_replay", "k_msp_count", "k_col_sums", "k_bin_group_sums", "k_bin_offsets", "k_scan_tail", "k_scan_sums",
    "k_scan_apply", "k_part2", "k_flag_if_gt", "k_slice_tag", "k_bin_hist", "k_bin_hist_stamp", "k_bin_hist_multi", "k_msp_leaf",
    "k_surv_place", "k_surv_hist", "k_surv_sort", "k_histo_bins", "k_bin_count", "k_part1", "k_leaf", "k_leaf_compact", "k_bin_scatter",
    "k_tmp_start", "k_split_bins", "k_coarse_counts", "k_histo", "k_count_reads", "k_count_pairs", "k_table_pairs", "k_tile_count",
    "k_tile_scan", "k_tile_emit",
))


def kname(s):
    m = re.search(r"(k_[a-z0-9_]+|__amd_rocclr_[A-Za-z]+)", s)
    return m.group(1) if m else s[:40]


def load(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            a = agg[kname(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return {k: (n, v / n) for k, (n, v) in agg.items()}


def main():
    f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(f) | set(w)):
        fn, fv = f.get(k, (0, 0.0))
        wn, wv = w.get(k, (0, 0.0))
        out[k] = {"launches": max(fn, wn), "FETCH_SIZE_KB_avg": round(fv, 1), "WRITE_SIZE_KB_avg": round(wv, 1),
                  "hbm_bytes_per_launch": int((2 * fv + wv) * 1024)}
    # The count -> sorted-records chain of one sample (what bench.py prices against the roofline) = EVERY kernel of the
    # profiled command that is not known to belong to another stage.  (Until round 5 this was a hand-kept list of chain
    # members, and a kernel added to the chain -- k_bin_hist_multi and its k_scan_sums / k_scan_apply, round 5 -- dropped
    # out of the traffic silently: 56 GB per sample.  An exclusion list errs the other way, and
    # tests/test_evidence_host.py fails on a kernel that is in neither list.)
    chain = [k for k in sorted(out) if k.startswith("k_") and k not in NOT_CHAIN]
    # round 2: a sample is many read blocks (one k_msp_part1 launch each, per shard pass): the number of sample
    # chains in the profiled run is given on the command line (3 per trio step)
    samples = int(sys.argv[4]) if len(sys.argv) > 4 else max(
        out.get("k_msp_part1", {}).get("launches", 0), out.get("k_part1", {}).get("launches", 0),
        out.get("k_bin_scatter", {}).get("launches", 0))
    if samples:
        tot = sum(out[k]["hbm_bytes_per_launch"] * out[k]["launches"] for k in chain if k in out)
        out["_chain"] = {"kernels": [k for k in chain if k in out], "samples": samples,
                         "hbm_bytes_per_sample": int(tot / samples),
                         "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), (2*FETCH+WRITE)*1024, all "
                                   "launches of the count chain of the profiled bench command / sample chains"}
        if len(sys.argv) > 5:
            out["genome"] = int(sys.argv[5])
    # what build this was measured on: bench.py quotes the traffic only when its own kernel sources have this fingerprint
    # (the round-3 file was taken before the last kernel change and quoted regardless)
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out["kernel_sources_sha16"] = bench.kernel_source_fingerprint()
    out["commit"] = os.environ.get("RFX_COMMIT", "unknown (set RFX_COMMIT=$(git rev-parse HEAD) when profiling)")
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    if "_chain" in out:
        print(f"count chain: {out['_chain']['hbm_bytes_per_sample'] / 1e6:.1f} MB per sample over {samples} samples")
    for k, v in sorted(((k, v) for k, v in out.items() if isinstance(v, dict) and "launches" in v), key=lambda kv: -kv[1]["hbm_bytes_per_launch"]):
        print(f"{k:26s} {v['launches']:4d} {v['FETCH_SIZE_KB_avg']:14.1f} {v['WRITE_SIZE_KB_avg']:14.1f} "
              f"{v['hbm_bytes_per_launch'] / 1e6:10.1f} MB")


if __name__ == "__main__":
    main()
