#!/usr/bin/env python3
"""profiles/<tag>_leases.md: one row per GPU lease (gpurun call) of tools/icache_round.sh -- time per instruction of straight-line
code of 16 ... 256 KiB (tools/icache_probe), the constraint kernels of the product build on that box, and the instruction-cache hit rate
of the largest of them.  max / min of every constraint kernel over the leases is the box-independence figure VERDICT round 2 asked for.
    python tools/lease_table.py gpurun_out profiles/r3_leases.md"""
import glob
import json
import os
import re
import sys

src, out = sys.argv[1], sys.argv[2]
rows, kernels = [], {}
for d in sorted(glob.glob(os.path.join(src, "icache_*"))):
    tag = os.path.basename(d)[7:]
    try:
        probe = {(p["code_kib"], p["lanes"], p["barrier_kib"]): p for p in json.load(open(os.path.join(d, "probe.json")))["icache_probe"]}
    except Exception:                                                # noqa: BLE001
        continue
    air = {}
    total = None
    for line in open(os.path.join(d, "air_ab.txt")):
        if not line.startswith("product"):
            continue
        m = re.search(r"([\d.]+) ms\s+constraint_eval ([\d.]+)", line)
        total = (float(m.group(1)), float(m.group(2)))
        for name, ms in re.findall(r"(<[\d,]+>) ([\d.]+)", line):
            air.setdefault(name, []).append(float(ms))
    hit = None
    summ = os.path.join(d, "summary.txt")
    if os.path.exists(summ):
        best = 0
        for line in open(summ):
            m = re.match(r"\s+air_kernel(<[\d,]+>)\s+\d+\s+([\d.e+]+)\s+([\d.]+)", line)
            if m and float(m.group(2)) > best:
                best, hit = float(m.group(2)), (m.group(1), float(m.group(3)))
    smi = open(os.path.join(d, "rocm_smi.txt")).read() if os.path.exists(os.path.join(d, "rocm_smi.txt")) else ""
    part = ",".join(re.findall(r"(?:Compute|Memory) Partition: (\w+)", smi))
    rows.append((tag, probe, air, total, hit, part))

# box-to-box ratios only among the leases that ran the newest lease's set of kernels (an earlier build has other launches under some of the names)
current = set(rows[-1][2]) if rows else set()
for tag, probe, air, total, hit, part in rows:
    if set(air) == current:
        for k, v in air.items():
            kernels.setdefault(k, []).append(min(v))

with open(out, "w") as fh:
    fh.write("# Per-lease table: straight-line code, constraint kernels, instruction cache\n\n")
    fh.write("Every row is one `gpurun` lease (a fresh MI355X box of the pool) running `tools/icache_round.sh <tag>`.  `probe x KiB` = time per 1000\n"
             "instructions of a straight-line kernel of that code size relative to the 16 KiB kernel of the same lease (128 lanes per workgroup, two\n"
             "waves per SIMD; `tools/icache_probe`).  Constraint-kernel times are per launch at 2^20 steps (ms), product build of that lease.\n\n")
    names = sorted({k for r in rows for k in r[2]})
    fh.write("| lease | partitions | probe 64 KiB | probe 176 KiB | probe 256 KiB | proof ms | constraint_eval ms | " + " | ".join("`%s`" % n for n in names) + " | I-cache hit (largest) |\n")
    fh.write("|---|---|---|---|---|---|---|" + "---|" * len(names) + "---|\n")
    for tag, probe, air, total, hit, part in rows:
        rel = lambda kib: "%.3f" % probe[(kib, 128, 0)]["rel"] if (kib, 128, 0) in probe else "-"
        fh.write("| %s | %s | %s | %s | %s | %s | %s | " % (tag, part or "-", rel(64), rel(176), rel(256), "%.2f" % total[0] if total else "-", "%.2f" % total[1] if total else "-")
                 + " | ".join(("%.3f" % min(air[n])) if n in air else "-" for n in names) + " | %s |\n" % ("%s %.4f" % hit if hit else "-"))
    fh.write("\nLeases whose kernel set differs from the last row's ran an earlier build (l1: the round-2 kernels, 172 KB stack launch `<2,1,4,8,88,0,1>`).\n")
    fh.write("\nmax / min over the leases of the current build: " + ", ".join("`%s` %.3f" % (k, max(v) / min(v)) for k, v in sorted(kernels.items()) if len(v) > 1) + "\n")
print(open(out).read())
