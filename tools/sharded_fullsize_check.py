#!/usr/bin/env python3
"""One-off validation of the sharded path at the bench size on ONE GPU: 2^20-step trace, `world` thread-ranks (default 8) sharing
the device; every rank must return the single-context proof.
usage: python tools/sharded_fullsize_check.py [world] [log_n] [log_blowup] [num_queries]     (8 24 4 100 = BASELINE config 5's shape)"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import distaff_amd as D
from distaff_amd import sharded

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
log_n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
options = dict(log_blowup=int(sys.argv[3]) if len(sys.argv) > 3 else 5, num_queries=int(sys.argv[4]) if len(sys.argv) > 4 else 50)
cols, program_hash, result = D.fibonacci_trace(log_n)
ctx = D.Context(log_n, 20, 1, 0, **options)
ctx.upload(cols)
expected = ctx.prove([1, 0], [result])
ctx.close()
t0 = time.time()
proofs = sharded.prove_local(cols, log_n, 20, 1, 0, [1, 0], [result], world, **options)
print("world %d, 2^%d, options %s: %s (%.1f s incl. context creation)" % (world, log_n, options, "all ranks equal the single-context proof" if all(p == expected for p in proofs) else "MISMATCH", time.time() - t0))
sys.exit(0 if all(p == expected for p in proofs) else 1)
