#!/usr/bin/env python3
"""Where a small proof spends its time: wall-clock and phases of dst_prove at the given sizes, the launches of one proof and the device
time they account for (the rest is launch latency and the host round trips of the Fiat-Shamir challenges).
    python tools/small_proof_time.py [log_n ...]        (default 10 12 16)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import distaff_amd as D

PH = ["lde", "trace_merkle", "constraint_eval", "combine", "constraint_lde_merkle", "deep_composition", "fri", "pow_queries", "openings"]
for log_n in [int(x) for x in sys.argv[1:]] or [10, 12, 16]:
    cols, program_hash, result = D.fibonacci_trace(log_n)
    ctx = D.Context(log_n, 20, 1, 0)
    ctx.upload(cols)
    for _ in range(3):
        proof = ctx.prove([1, 0], [result])
    runs = 20
    phases = [0.0] * 9
    t0 = time.perf_counter()
    for _ in range(runs):
        proof = ctx.prove([1, 0], [result])
        for i, v in enumerate(ctx.phase_ms()):
            phases[i] += v / runs
    ms = (time.perf_counter() - t0) / runs * 1e3
    ctx.set_profiling(1); ctx.kernel_stats(reset=True)
    ctx.prove([1, 0], [result])
    st = ctx.kernel_stats(reset=True)
    ctx.set_profiling(0)
    launches = sum(v["launches"] for v in st.values())
    dev = sum(v["ms"] for v in st.values())
    print("2^%d steps: %.3f ms per proof; %d launches, %.3f ms of device time in them" % (log_n, ms, launches, dev))
    print("   phases: " + "  ".join("%s %.3f" % (k, v) for k, v in zip(PH, phases)))
    top = sorted(st.items(), key=lambda kv: -kv[1]["launches"])[:8]
    print("   launches: " + "  ".join("%s x%d (%.3f ms)" % (k[:28], v["launches"], v["ms"]) for k, v in top))
    ctx.close()
