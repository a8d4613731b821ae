#!/usr/bin/env python3
"""Time of the constraint evaluation for a deep-stack program (stack depth 16, trace from the oracle VM: test infrastructure used
as a trace generator only) under the any-shape instances of the constraint kernel.   python tools/deep_time.py [iterations]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle as O
import distaff_amd as D

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
src = "begin " + " ".join("push.%d" % (3 + i) for i in range(12)) + " repeat.%d swap dup.2 drop add end end" % iters
t0 = time.time()
t = O.Trace(src, [1, 0])
print("trace: W=%d n=2^%d ctx=%d loop=%d stack=%d (%.1f s on the host)" % (t.width, t.length.bit_length() - 1, t.ctx_depth, t.loop_depth, t.stack_depth, time.time() - t0), flush=True)
outputs = O.to_ints(t.columns[15 + t.ctx_depth + t.loop_depth:, t.length - 1, :])[:1]
digests = {}
for inst in ("deep", "generic"):
    os.environ["DISTAFF_AIR"] = inst
    ctx = D.Context(t.length.bit_length() - 1, t.width, t.ctx_depth, t.loop_depth)
    ctx.upload(t.columns)
    for _ in range(3):
        proof = ctx.prove(t.public_inputs, outputs)
    ph = ctx.phase_ms()
    digests[inst] = D.blake3(proof).hex()[:16]
    print("%-8s constraint evaluation %.3f ms, whole proof %.2f ms, proof blake3 %s" % (inst, ph[2], sum(ph), digests[inst]), flush=True)
    ctx.close()
assert digests["deep"] == digests["generic"], "the two instances disagree"
