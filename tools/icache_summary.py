#!/usr/bin/env python3
"""Per-kernel instruction-cache counters from the rocprofv3 passes of tools/icache_round.sh: requests, hit rate, misses per wavefront
and the bytes the misses stand for (64-byte lines) per launch."""
import collections
import csv
import glob
import os
import re
import sys


def table(d):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        return {}
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(files[0])):
        name = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "").replace(" ", "")
        name = re.sub(r"(\d+)u\b", r"\1", name.replace("true", "1").replace("false", "0"))
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (name, r.get("Dispatch_Id"))
        if key not in seen:
            seen.add(key); launches[name] += 1
    return {n: dict(v, launches=launches[n]) for n, v in acc.items()}


def show(title, t, only=None):
    print(title)
    print("  %-44s %5s %12s %7s %12s %10s %12s" % ("kernel", "n", "icache req", "hit", "miss/wave", "MB/launch", "wait_inst"))
    for n, v in sorted(t.items(), key=lambda kv: -kv[1].get("SQC_ICACHE_REQ", 0)):
        if only and not re.search(only, n):
            continue
        req, hit, miss = v.get("SQC_ICACHE_REQ", 0), v.get("SQC_ICACHE_HITS", 0), v.get("SQC_ICACHE_MISSES", 0)
        waves = max(v.get("SQ_WAVES", 0), 1)
        busy = max(v.get("SQ_BUSY_CYCLES", 0), 1)
        print("  %-44s %5d %12.3e %7.4f %12.1f %10.1f %12.3f" % (n[:44], v["launches"], req / v["launches"], hit / max(req, 1), miss / waves,
                                                                  miss * 64 / 1e6 / v["launches"], v.get("SQ_WAIT_INST_ANY", 0) / busy))


d = sys.argv[1]
show("icache_probe", table(os.path.join(d, "probe_pmc")))
show("bench (product library)", table(os.path.join(d, "bench_pmc")), only=r"air_kernel|ntt_pass|trace_leaves|merkle_level|cross8|fold8")
