#!/usr/bin/env python3
"""How much of a sharded proof is repeated on every rank -- measured on ONE GPU: with G thread-ranks sharing the device the sharded work adds up
to the single-context work, so   T(G ranks on one GPU) - T(single context)  ~  (G - 1) x (work every rank repeats) + the exchanges (device
copies here).  Prints the estimate per rank for G = 2, 4, 8 at the bench size.      python tools/replicated_estimate.py [log_n]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import distaff_amd as D

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cols, program_hash, result = D.fibonacci_trace(log_n)


def timed(fn, runs=4):
    fn(); fn()
    t0 = time.perf_counter()
    for _ in range(runs):
        out = fn()
    return (time.perf_counter() - t0) / runs * 1e3, out


ctx = D.Context(log_n, 20, 1, 0)
ctx.upload(cols)
single, expected = timed(lambda: ctx.prove([1, 0], [result]))
ctx.close()
print("single context: %.2f ms" % single)
for world in (2, 4, 8):
    ctxs = []
    for r in range(world):
        c = D.Context(log_n, 20, 1, 0, rank=r, world=world)
        c.upload_owned(cols)
        ctxs.append(c)
    ms, proof = timed(lambda: D.prove_sharded_local(ctxs, [1, 0], [result]))
    assert proof == expected
    print("%d thread-ranks on one GPU: %.2f ms  ->  repeated per rank ~ %.2f ms (incl. the in-process exchanges)" % (world, ms, (ms - single) / (world - 1)))
    for c in ctxs:
        c.close()
