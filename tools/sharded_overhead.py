#!/usr/bin/env python3
"""Cost of the sharded orchestration itself on ONE GPU without torch: ShardedProver over a one-rank in-process communicator
against dst_prove on the same trace (stage times of the sharded run are printed).   python tools/sharded_overhead.py [log_n]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import distaff_amd as D
from distaff_amd import sharded

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cols, program_hash, result = D.fibonacci_trace(log_n)
ctx = D.Context(log_n, 20, 1, 0)
ctx.upload(cols)
for _ in range(2):
    expected = ctx.prove([1, 0], [result])
t0 = time.perf_counter()
for _ in range(5):
    ctx.prove([1, 0], [result])
single = (time.perf_counter() - t0) / 5 * 1e3
prover = sharded.ShardedProver(ctx, sharded.LocalComm.create(1)[0])
for _ in range(2):
    proof = prover.prove([1, 0], [result])
assert proof == expected
stages = {}
t0 = time.perf_counter()
for _ in range(5):
    prover.prove([1, 0], [result])
    for k, v in prover.stage_ms.items():
        stages[k] = stages.get(k, 0.0) + v / 5
shard = (time.perf_counter() - t0) / 5 * 1e3
print("2^%d: dst_prove %.2f ms, sharded orchestration (1 rank, host-staged exchanges) %.2f ms" % (log_n, single, shard))
print("   " + ", ".join("%s %.2f" % (k, v) for k, v in stages.items()))
ctx.close()
