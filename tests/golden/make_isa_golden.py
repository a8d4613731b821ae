#!/usr/bin/env python3
"""Regenerates tests/golden/isa_traces.npz + isa_proof_digests.json: execution traces of three programs that use loops, branches and the
hashing / comparison macros (made by the oracle's VM, which tests/test_oracle_isa.py pins against the reference's fixtures) and the BLAKE3
digests of the oracle's complete proofs for them.  The GPU test that uses them needs no oracle run; the CPU test re-derives both from
the oracle, so an accidental change of the oracle's VM or prover shows up.          usage: python tests/golden/make_isa_golden.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as O

CASES = {
    "while_5_iterations": ("begin mul read while.true dup mul read end end", [5, 3], [1, 1, 1, 1, 1, 0], [], 1),                 # processor/mod.rs:284-346
    "if_false": ("begin read if.true add push.3 else push.7 add push.8 end mul end", [5, 3], [0], [], 2),                          # processor/mod.rs:238-280
    "comparison_6": ("begin push.9 read dup.2 lt.128 if.true mul else add end dup isodd.128 end", [], [6], [], 2),                # examples/comparison.rs
}


def build():
    arrays, cases = {}, []
    for name, (src, pub, a, b, nout) in CASES.items():
        t = O.Trace(src, pub, a, b)
        p = O.Prover.from_trace(t, nout, grinding=16)
        proof = p.prove()
        arrays[name] = t.columns
        cases.append({"name": name, "source": src, "public_inputs": [str(v) for v in pub], "outputs": [str(v) for v in p.outputs], "width": t.width,
                      "length": t.length, "ctx_depth": t.ctx_depth, "loop_depth": t.loop_depth, "grinding_factor": 16,
                      "program_hash": t.program_hash.hex(), "proof_bytes": len(proof), "proof_blake3": O.blake3(proof).hex(),
                      "columns_blake3": O.blake3(np.ascontiguousarray(t.columns).tobytes()).hex()})
    return arrays, cases


if __name__ == "__main__":
    arrays, cases = build()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "isa_traces.npz"), **arrays)
    with open(os.path.join(ROOT, "tests", "golden", "isa_proof_digests.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_isa_golden.py (oracle/liboracle.so)", "cases": cases}, f, indent=1)
    print("wrote %d cases" % len(cases))
