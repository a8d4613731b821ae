#!/usr/bin/env python3
"""Condenses the rocprofv3 CSVs of tools/profile_round.sh into the small tracked files under profiles/:
   <tag>_kernel_stats.csv  (the --kernel-trace --stats table, verbatim)
   <tag>_pmc_per_kernel.csv (per kernel: launches, avg duration, FETCH_SIZE / WRITE_SIZE per launch [raw and corrected], SQ counters)
   <tag>_bench_under_rocprof.json (the bench line printed by the profiled command)
usage: summarize_profile.py <prof_dir> <out_dir> <tag>"""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict


def read_counters(path):
    """-> {kernel: {counter: [sum, dispatches]}} ; rocprofv3 emits one row per (dispatch, counter)"""
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    f = os.path.join(path, "p_counter_collection.csv")
    if not os.path.exists(f):
        return acc
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            a = acc[row["Kernel_Name"]][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
    return acc


def read_clock(path, xcds=8):
    """-> {kernel: MHz}: the shader clock DURING the launches of a kernel, from the pass that collected GRBM_GUI_ACTIVE (busy cycles, summed over
    the device's `xcds` XCDs) against the launch durations of the SAME pass (its own kernel trace): sum of cycles / xcds / sum of durations.
    The nominal 2.4 GHz is not what the device runs at under this path's arithmetic (profiles/r6_power_clock.md)."""
    f, k = os.path.join(path, "p_counter_collection.csv"), os.path.join(path, "p_kernel_trace.csv")
    if not (os.path.exists(f) and os.path.exists(k)):
        return {}
    dur = {}
    with open(k, newline="") as fh:
        for row in csv.DictReader(fh):
            dur[row["Dispatch_Id"]] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    cyc, ns = defaultdict(float), defaultdict(float)
    seen = set()
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] != "GRBM_GUI_ACTIVE" or row["Dispatch_Id"] not in dur:
                continue
            cyc[row["Kernel_Name"]] += float(row["Counter_Value"])
            if row["Dispatch_Id"] not in seen:
                seen.add(row["Dispatch_Id"]); ns[row["Kernel_Name"]] += dur[row["Dispatch_Id"]]
    return {k2: cyc[k2] / xcds / ns[k2] * 1e3 for k2 in cyc if ns[k2] > 0}


def main():
    prof, out, tag = sys.argv[1:4]
    os.makedirs(out, exist_ok=True)
    shutil.copy(os.path.join(prof, "stats", "p_kernel_stats.csv"), os.path.join(out, tag + "_kernel_stats.csv"))
    if os.path.exists(os.path.join(prof, "bench_under_rocprof.json")):
        shutil.copy(os.path.join(prof, "bench_under_rocprof.json"), os.path.join(out, tag + "_bench_under_rocprof.json"))
    stats = {}
    with open(os.path.join(prof, "stats", "p_kernel_stats.csv"), newline="") as fh:
        for row in csv.DictReader(fh):
            stats[row["Name"]] = row
    merged = defaultdict(dict)
    for sub in ("fetch", "write", "sq1", "sq2", "sq3"):
        for k, cs in read_counters(os.path.join(prof, sub)).items():
            for cname, (total, cnt) in cs.items():
                merged[k][cname] = total / max(cnt, 1)
    clock = read_clock(os.path.join(prof, "grbm"))
    counters = sorted({c for v in merged.values() for c in v})
    with open(os.path.join(out, tag + "_pmc_per_kernel.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        # FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3; gfx950 correction for wide coalesced reads: x2 on FETCH_SIZE
        # (MI355X_MICROARCH.md, "HBM [CDNA4]"); WRITE_SIZE is left uncorrected (uncalibrated per the guide).
        # (x 2 holds for kernels that read wide; 64-byte segments are tallied in full: profiles/r6_fetch_factor.txt -- bench.py applies the factor per kernel)
        w.writerow(["kernel", "calls", "avg_ns", "pct_time", "fetch_bytes_per_launch_raw", "fetch_bytes_per_launch_x2", "write_bytes_per_launch_raw", "sclk_MHz"] + counters)
        for k, row in sorted(stats.items(), key=lambda kv: -float(kv[1]["TotalDurationNs"])):
            m = merged.get(k, {})
            fetch = m.get("FETCH_SIZE"); write = m.get("WRITE_SIZE")
            w.writerow([k, row["Calls"], row["AverageNs"], row["Percentage"],
                        "" if fetch is None else int(fetch * 1024), "" if fetch is None else int(fetch * 2048),
                        "" if write is None else int(write * 1024), ("%.0f" % clock[k]) if k in clock else ""] + [("%.1f" % m[c]) if c in m else "" for c in counters])
    print("wrote", out)


if __name__ == "__main__":
    main()
