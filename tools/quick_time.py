#!/usr/bin/env python3
"""Wall-clock of dst_prove on a Fibonacci trace without importing torch (seconds to start on a fresh box; bench.py is the
contract benchmark, this is the quick look).   python tools/quick_time.py [log_n] [runs]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import distaff_amd as D

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cols, program_hash, result = D.fibonacci_trace(log_n)
ctx = D.Context(log_n, 20, 1, 0)
ctx.upload(cols)
for _ in range(2):
    proof = ctx.prove([1, 0], [result])
t0 = time.perf_counter()
for _ in range(runs):
    proof = ctx.prove([1, 0], [result])
ms = (time.perf_counter() - t0) / runs * 1e3
print("2^%d steps: %.2f ms per proof, %.3e trace-cells/s, %d proof bytes (blake3 %s), phases %s" % (
    log_n, ms, (1 << log_n) * 20 / (ms * 1e-3), len(proof), D.blake3(proof).hex()[:16], [round(v, 2) for v in ctx.phase_ms()]))
ctx.close()
