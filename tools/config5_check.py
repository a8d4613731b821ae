#!/usr/bin/env python3
"""BASELINE config 5 on one GPU: Fibonacci trace of 2^24 steps, ProofOptions(16, 100, 20) -- proves, times the phases and lets the
oracle's restatement of the reference verifier judge the proof (size-independent acceptance check).  ~3 minutes, mostly host-side
trace generation.   usage: python tools/config5_check.py [log_n]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import distaff_amd as D
import oracle as O

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
t0 = time.time()
cols, program_hash, result = D.fibonacci_trace(log_n)
print("trace generated in %.1f s" % (time.time() - t0), flush=True)
ctx = D.Context(log_n, 20, 1, 0, log_blowup=4, num_queries=100, grinding=20)
ctx.upload(cols)
proof = ctx.prove([1, 0], [result], cap=1 << 24)
t0 = time.time()
proof = ctx.prove([1, 0], [result], cap=1 << 24)
dt = time.time() - t0
print("2^%d steps, blowup 16, 100 queries: %.1f ms per proof, %.3g trace-cells/s, proof %d bytes" % (log_n, dt * 1e3, (20 << log_n) / dt, len(proof)))
print("phases (ms):", ["%.1f" % x for x in ctx.phase_ms()])
ctx.close()
ok, err = O.verify(proof, program_hash, [1, 0], [result])
print("oracle verifier:", "accepted" if ok else "REJECTED: " + err)
sys.exit(0 if ok else 1)
