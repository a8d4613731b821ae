#!/usr/bin/env python3
"""Regenerates tests/golden/proof_digests.json: BLAKE3 digests (and sizes) of complete serialised StarkProofs produced by the CPU
oracle for small Fibonacci traces.  The oracle is pinned by the reference's own vectors (tests/test_oracle_*.py); these digests pin
the oracle's END-TO-END output against accidental change and give the GPU tests a fixture that needs no oracle run.
usage: python tests/golden/make_proof_golden.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as O

cases = []
for log_n, ext, nq, grinding in ((7, 32, 50, 20), (8, 16, 100, 20), (10, 32, 50, 20), (12, 32, 50, 16), (10, 64, 30, 12)):
    t = O.fibonacci_trace(1 << log_n)
    proof = O.Prover.from_trace(t, 1, ext=ext, num_queries=nq, grinding=grinding).prove()
    cases.append({"program": "fibonacci", "log_n": log_n, "extension_factor": ext, "num_queries": nq, "grinding_factor": grinding,
                  "proof_bytes": len(proof), "proof_blake3": O.blake3(proof).hex(), "program_hash": t.program_hash.hex()})
with open(os.path.join(ROOT, "tests", "golden", "proof_digests.json"), "w") as f:
    json.dump({"generator": "tests/golden/make_proof_golden.py (oracle/liboracle.so)", "cases": cases}, f, indent=1)
print("wrote %d cases" % len(cases))
