#!/usr/bin/env python3
"""Where the sharded code path (dst_prove_sharded with ONE rank over RCCL) spends more than the single-context path on the same box:
runs `bench.py` both ways and prints the kernels whose time per proof differs by more than 0.03 ms, and the launch counts.
    python tools/sharded_overhead.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-upload-leg", "--no-verify"] + extra,
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
    return json.loads([l for l in out.splitlines() if l.startswith("{")][-1])


a, b = run([]), run(["--force-sharded"])
print("single %.3f ms, sharded (1 rank) %.3f ms" % (a["ms_per_step"], b["ms_per_step"]))
print("phases single :", a["phase_ms"])
print("phases sharded:", b["phase_ms"])
ka, kb = a["kernels"], b["kernels"]
print("launches: single %d, sharded %d; device time %.3f / %.3f ms" % (sum(v["launches"] for v in ka.values()), sum(v["launches"] for v in kb.values()),
                                                                      sum(v["ms_per_step"] for v in ka.values()), sum(v["ms_per_step"] for v in kb.values())))
for k in sorted(set(ka) | set(kb), key=lambda k: -abs(kb.get(k, {}).get("ms_per_step", 0) - ka.get(k, {}).get("ms_per_step", 0))):
    x, y = ka.get(k, {"ms_per_step": 0, "launches": 0}), kb.get(k, {"ms_per_step": 0, "launches": 0})
    if abs(x["ms_per_step"] - y["ms_per_step"]) >= 0.03:
        print("  %-40s single %7.3f ms x%-3d   sharded %7.3f ms x%-3d   %+.3f" % (k[:40], x["ms_per_step"], x["launches"], y["ms_per_step"], y["launches"], y["ms_per_step"] - x["ms_per_step"]))
