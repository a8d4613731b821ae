"""End-to-end checks of the oracle prover/verifier pair, mirroring the reference's integration tests
(/root/reference/src/tests/mod.rs:11-63), its FRI tests (src/stark/fri/mod.rs:39-95) and the TraceTable tests
(src/stark/trace/trace_table.rs:298-363).  BASELINE config 1 (Fibonacci, 2^10 steps, default options) runs here."""
import random

import numpy as np
import pytest

P = 2**128 - 45 * 2**40 + 1
LOW_DEGREE_ERR = "verification of low-degree proof failed: evaluations did not match column value at depth 0"


def test_execute_verify(oracle):
    O = oracle
    t = O.Trace("begin swap dup.2 drop add swap dup.2 drop add swap dup.2 drop add end", [1, 0])    # tests/mod.rs:11-29
    p = O.Prover.from_trace(t, 1)
    proof = p.prove()
    assert p.outputs == [3]
    assert O.verify(proof, t.program_hash, [1, 0], [3]) == (True, "")
    assert O.verify(proof, t.program_hash, [1, 1], [3]) == (False, LOW_DEGREE_ERR)                   # tests/mod.rs:47-62
    assert O.verify(proof, t.program_hash, [1, 0], [5]) == (False, LOW_DEGREE_ERR)
    h2 = bytes([1]) + t.program_hash[1:]
    assert O.verify(proof, h2, [1, 0], [3]) == (False, LOW_DEGREE_ERR)
    bad = bytearray(proof); bad[40] ^= 1
    ok, err = O.verify(bytes(bad), t.program_hash, [1, 0], [3])
    assert not ok


def test_config1_fibonacci_2_10_default_options(oracle):
    O = oracle
    t = O.fibonacci_trace(1 << 10)                              # BASELINE.json configs[0]
    p = O.Prover.from_trace(t, 1)
    proof = p.prove()
    assert p.get_u64("constraints_ok") == [1]
    assert 60_000 < len(proof) < 110_000                        # README.md:152 quotes ~80 KB
    assert O.verify(proof, t.program_hash, t.public_inputs, p.outputs) == (True, "")
    cp = O.to_ints(p.get("constraint_poly"))
    assert max(i for i, v in enumerate(cp) if v) == 7 * 1024    # constraint_poly.rs:58-61
    assert O.infer_degree(p.get("composed_evaluations")) == 7 * 1024 - 1      # prover.rs:110
    positions = p.get_u64("positions")
    assert len(positions) == 50 and all(x % 32 for x in positions)
    assert p.get_u64("fri_layers") == [5]                       # 2^15, 2^13, 2^11, 2^9 + remainder 2^7


def test_other_options_and_programs(oracle):
    O = oracle
    t = O.fibonacci_trace(256)
    p = O.Prover.from_trace(t, 1, ext=16, num_queries=100, grinding=12)      # config 5's options at a small size
    proof = p.prove()
    assert O.verify(proof, t.program_hash, t.public_inputs, p.outputs) == (True, "")
    t = O.Trace("begin add block push.5 mul push.7 end end", [1, 2])
    p = O.Prover.from_trace(t, 2, grinding=8)
    assert p.outputs == [7, 15]
    proof = p.prove()
    assert O.verify(proof, t.program_hash, [1, 2], [7, 15]) == (True, "")


def test_invalid_trace_is_detected(oracle):
    O = oracle
    t = O.fibonacci_trace(128)
    cols = t.columns.copy()
    cols[16, 40, 0] += 1                                        # corrupt one stack cell
    p = O.Prover(cols, t.ctx_depth, t.loop_depth, [1, 0], [1], grinding=4)
    with pytest.raises(RuntimeError, match="transition constraints"):    # evaluator.rs:155
        p.prove()


def test_trace_table_composition_consistency(oracle):
    # trace_table.rs:298-363: eval_polys_at == interpolation, composition poly == slow path; here via field identities
    O = oracle
    t = O.fibonacci_trace(128)
    p = O.Prover.from_trace(t, 1, grinding=4)
    for k in range(1, 7):
        p.step(k)
    polys = p.get("polys"); regs = p.get("registers")
    n, N = 128, 128 * 32
    g_lde = O.root_of_unity(N)
    for c in (0, 3, 16, 19):
        assert O.to_ints(O.fft_eval(polys[c]))[:n] == O.to_ints(t.columns[c])          # extend(): poly interpolates the trace
        x = pow(g_lde, 77, P)
        assert O.poly_eval(polys[c], x) == O.to_ints(regs[c, 77])
    draws = O.to_ints(p.get("deep_draws")); z = draws[0]
    z1 = O.to_ints(p.get("trace_at_z1")); z2 = O.to_ints(p.get("trace_at_z2"))
    g = O.root_of_unity(n)
    for c in range(20):
        assert z1[c] == O.poly_eval(polys[c], z) and z2[c] == O.poly_eval(polys[c], z * g % P)
    # composition evaluated at an LDE point equals the verifier's formula (verifier.rs:98-162)
    comp = p.get("composed_evaluations"); cons = p.get("constraint_evaluations"); cpoly = p.get("constraint_poly")
    cc1, cc2, k1, k2, k3 = draws[1:257], draws[257:513], draws[513], draws[514], draws[515]
    pos = 1234
    x = pow(g_lde, pos, P)
    inv = lambda v: pow(v, P - 2, P)
    acc = 0
    for c in range(20):
        v = O.to_ints(regs[c, pos])
        acc += (v - z1[c]) * inv(x - z) % P * cc1[c] + (v - z2[c]) * inv(x - z * g) % P * cc2[c]
    acc %= P
    acc = (acc * k1 + acc * pow(x, 6 * n + 1, P) % P * k2) % P
    cz = O.poly_eval(cpoly, z)
    acc = (acc + (O.to_ints(cons[pos]) - cz) * inv(x - z) % P * k3) % P
    assert acc == O.to_ints(comp[pos])


def test_fri_prove_verify(oracle):
    O = oracle
    rnd = random.Random(5)
    def evals(degree, size=512):
        return O.fft_eval(O.to_arr([rnd.randrange(P) for _ in range(degree + 1)] + [0] * (size - degree - 1)))
    assert O.fri_prove_verify(evals(63), 63) == (True, "")                                        # fri/mod.rs:39-58
    assert O.fri_prove_verify(evals(63), 62) == (False, "remainder is not a valid degree 14 polynomial")   # :61-75
    assert O.fri_prove_verify(evals(64), 63) == (False, "remainder is not a valid degree 15 polynomial")   # :77-87
    assert O.fri_prove_verify(evals(63), 63, drop_first_evaluation=True) == (False, "evaluations did not match column value at depth 0")  # :89-93


def test_stepwise_equals_one_shot_and_challenge_override(oracle):
    O = oracle
    t = O.fibonacci_trace(128)
    a = O.Prover.from_trace(t, 1, grinding=6); pa = a.prove()
    b = O.Prover.from_trace(t, 1, grinding=6)
    for k in range(1, 10):
        b.step(k)
    assert b.get_bytes("proof") == pa
    # feeding the same challenges explicitly (the C-ABI's mode of operation) gives the same commitments
    c = O.Prover.from_trace(t, 1, grinding=6)
    c.step(1); c.step(2); c.step(3, a.get("constraint_draws")); c.step(4); c.step(5); c.step(6, a.get("deep_draws"))
    assert c.get_bytes("roots") == a.get_bytes("roots")
    assert (c.get("composed_evaluations") == a.get("composed_evaluations")).all()


def test_oracle_reproduces_golden_proof_digests(oracle):
    """End-to-end regression pin: the oracle's serialised proofs hash to the committed digests (tests/golden/proof_digests.json)."""
    import json
    import os
    O = oracle
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "proof_digests.json")) as f:
        cases = json.load(f)["cases"]
    for c in cases:
        if c["log_n"] > 10:
            continue                                               # keep the CPU suite short; the GPU test covers every case
        t = O.fibonacci_trace(1 << c["log_n"])
        proof = O.Prover.from_trace(t, 1, ext=c["extension_factor"], num_queries=c["num_queries"], grinding=c["grinding_factor"]).prove()
        assert len(proof) == c["proof_bytes"] and O.blake3(proof).hex() == c["proof_blake3"], c
        assert t.program_hash.hex() == c["program_hash"]


def test_point_evaluator_equals_the_whole_domain_loop(oracle):
    """orc_evaluate_at (the evaluator on ONE row pair, what the full-size GPU tests sample with) against the oracle prover's own loop over
    the 8n-point domain (prover.rs:53-64) -- transition and both boundary combinations, incl. trace steps and the excepted last step."""
    O = oracle
    t = O.fibonacci_trace(128)
    p = O.Prover.from_trace(t, 1, grinding=8)
    for k in range(1, 4):
        p.step(k)
    n, B, W = 128, 32, t.width
    regs = p.get("registers")
    te, ie, fe = O.to_ints(p.get("t_evaluations")), O.to_ints(p.get("i_evaluations")), O.to_ints(p.get("f_evaluations"))
    draws = p.get("constraint_draws")
    last = O.to_ints(regs[:, (n - 1) * B, :])
    g8 = O.root_of_unity(8 * n)
    for step in list(range(0, 24)) + [8 * n - 8, 8 * n - 1, 517, 700]:
        pos = step * (B // 8)
        cur, nxt = O.to_ints(regs[:, pos, :]), O.to_ints(regs[:, (pos + B) % (n * B), :])
        tv, iv, fv, ok = O.evaluate_at(n, t.ctx_depth, t.loop_depth, W - 15 - t.ctx_depth - t.loop_depth, draws, last[1:3], last[0], t.public_inputs, p.outputs,
                                       step, O.exp(g8, step), cur, nxt)
        assert ok and (tv, iv, fv) == (te[step], ie[step], fe[step]), step
