"""Pins the oracle's VM / assembler / AIR on the WHOLE instruction set and on if / while blocks against the reference's own fixtures:

* src/programs/assembly/tests.rs:1-402        -- `{:?}` of compiled programs (every block kind, repeat, macros)
* src/programs/tests/mod.rs:11-150            -- program hash == hash of the reference's independent walk (utils.rs), step counts
* src/processor/stack/tests/{mod,comparisons,conditional}.rs -- the user stack operation by operation (states, depth, max_depth, panics)
* src/processor/mod.rs:237-346                -- execute_if_else, execute_loop: final states, op counters, trace shapes
* src/tests/mod.rs:66-315, src/tests/comparisons.rs:8-107 -- end to end: outputs equal the literals, the oracle prover's constraint
  check (evaluator.rs:152-158) passes on the trace, i.e. EVERY transition constraint of oracle/air.hpp vanishes on every row, and the
  oracle verifier (which evaluates the constraints at the out-of-domain point through evaluate_transition_at) accepts the proof.
* src/examples/{conditional,comparison,collatz,range,merkle}.rs -- the example programs with their expected results.

The end-to-end cases run with the reference's default blowup (32) and query count (50); grinding is lowered from 20 to 8 bits (the
proof of work does not touch the AIR) to keep the CPU suite short."""
import random

import pytest

P = 2**128 - 45 * 2**40 + 1


def norm(s):
    return " ".join(s.split())


def inv(x):
    return pow(x, P - 2, P)


# ---------------------------------------------------------------------------------------------------------------------------------
# assembler: src/programs/assembly/tests.rs
# ---------------------------------------------------------------------------------------------------------------------------------
def _assembly_cases():
    """the reference's assembler fixtures (src/programs/assembly/tests.rs:1-402: all 13 tests), kept as data in tests/golden/"""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "assembly_debug_strings.json")
    return json.load(open(path))["cases"]


ASSEMBLY_CASES = _assembly_cases()


@pytest.mark.parametrize("case", range(len(ASSEMBLY_CASES)))
def test_assembler_debug_strings(oracle, case):
    c = ASSEMBLY_CASES[case]
    assert oracle.program_debug(c["source"]) == c["debug"], c["reference_test"]


def test_assembler_macros(oracle):
    """parsers.rs: the macro expansions not covered by the reference's own string tests, checked against parsers.rs line by line."""
    O = oracle
    body = lambda src: O.program_debug("begin " + src + " end")
    first = lambda src, k: body(src).split()[1:1 + k]
    assert first("dup.3", 3) == ["dup4", "roll4", "drop"]                                  # parsers.rs:92
    assert first("pad.7", 4) == ["pad2", "pad2", "dup4", "drop"]                           # :111
    assert first("pick.2", 5) == ["dup4", "roll4", "drop", "drop", "drop"]                 # :125-127
    assert first("drop.7", 3) == ["dup", "drop4", "drop4"]                                 # :146
    assert first("div", 2) == ["inv", "mul"] and first("ne", 3) == ["read::eq", "eq", "not"]    # :210, :266
    gt = body("gt.4").split()                                                              # :272-300
    assert gt[1:5] == ["pad2", "pad2", "pad2", "dup"] and gt[5:8] == ["noop"] * 3 and gt[8] == "push(8)"
    assert gt[9:13] == ["cmp.4", "cmp", "cmp", "cmp"]
    assert gt[13:22] == ["drop4", "pad2", "swap4", "roll4", "asserteq", "asserteq", "roll4", "dup", "drop4"]
    lt = body("lt.4").split()
    assert lt[13:21] == ["drop4", "pad2", "swap4", "roll4", "asserteq", "asserteq", "dup", "drop4"]   # :326-329
    odd = body("isodd.4").split()                                                          # :363-393
    assert odd[8:12] == ["push(1)", "swap", "dup", "binacc.4"] and odd[12:15] == ["swap2", "roll4", "dup"]
    assert odd[15:18] == ["binacc"] * 3 and odd[18:24] == ["drop", "drop", "swap", "roll4", "asserteq", "drop"]
    sm = body("smpath.3").split()                                                          # :444-484
    assert sm[1:6] == ["read2", "swap2", "read2", "cswap2", "pad2"] and sm[6:16] == ["noop"] * 10
    assert sm[16:26] == ["rescr"] * 10 and sm[26:32] == ["drop4", "read2", "swap2", "read2", "cswap2", "pad2"]
    assert sm[32:42] == ["rescr"] * 10 and sm[42] == "drop4"
    pm = body("pmpath.3").split()                                                          # :488-538
    assert pm[1:3] == ["read2.3", "pad2"] and pm[8:15] == ["push(1)", "swap", "dup", "binacc", "swap4", "cswap2", "pad2"]
    assert pm[16:26] == ["rescr"] * 10 and pm[26:35] == ["drop4", "pad2", "swap2", "read2", "swap4", "binacc", "swap4", "cswap2", "pad2"]
    assert pm[35:48] == ["noop"] * 13 and pm[48:58] == ["rescr"] * 10 and pm[58:63] == ["drop4", "swap2", "drop", "roll4", "asserteq"]
    for bad in ("begin end", "begin add", "add end", "begin push.340282366920938463463374557953744961537 end", "begin dup.5 end",
                "begin repeat.1 add end end", "begin if add end end", "begin add else add end end", "begin block end end"):
        with pytest.raises(RuntimeError):
            O.program_debug(bad)


# ---------------------------------------------------------------------------------------------------------------------------------
# program hashes: src/programs/tests/mod.rs (blocks written as assembly that compiles to the same block tree)
# ---------------------------------------------------------------------------------------------------------------------------------
def _walk(O, source, conditions):
    state, program_hash, steps = O.program_traverse(source, conditions)
    return O.to_arr(state[:2]).tobytes() == program_hash, steps


def test_program_hash_matches_independent_walk(oracle):
    O = oracle
    a15, m15, i15 = " add" * 15, " mul" * 15, " inv" * 15
    # the first block of every program there is BEGIN + 14 NOOPs: a forced span before a block (assembly/mod.rs:165)
    assert O.program_debug("begin noop end") == "begin" + " noop" * 14 + " end"
    assert _walk(O, "begin noop end", []) == (True, 31)                                                  # single_block :11-21
    assert _walk(O, "begin block%s end block%s end end" % (a15, m15), []) == (True, 95)                   # linear_blocks :23-50
    assert _walk(O, "begin block%s end block%s end%s end" % (a15, m15, i15), []) == (True, 111)
    assert _walk(O, "begin block%s end block%s block%s end end end" % (a15, m15, i15), []) == (True, 127)  # nested_blocks :52-72
    cond = "begin if.true%s else%s end end" % (" add" * 14, " mul" * 13)                                   # conditional_program :74-108
    assert O.program_debug(cond).split()[15:47] == ["if", "assert"] + ["add"] * 14 + ["else", "not", "assert"] + ["mul"] * 13
    assert _walk(O, cond, [1]) == (True, 63)
    assert _walk(O, cond, [0]) == (True, 63)
    loop = "begin while.true%s end end" % (" add" * 14)                                                    # simple_loop :110-145
    assert _walk(O, loop, [0]) == (True, 63)
    assert _walk(O, loop, [0, 1]) == (True, 79)
    assert _walk(O, loop, [0, 1, 1, 1]) == (True, 111)


# ---------------------------------------------------------------------------------------------------------------------------------
# the user stack, operation by operation: src/processor/stack/tests/*.rs
# ---------------------------------------------------------------------------------------------------------------------------------
def _one(O, inputs, op, a=(), b=()):
    states, depth, max_depth = O.stack_run(inputs, a, b, [op])
    return states[1], depth[1], max_depth[1]


STACK_CASES = [
    # (public inputs, op, state after, depth, max_depth)                                   stack/tests/mod.rs
    ([1, 2, 3, 4], "noop", [1, 2, 3, 4, 0, 0, 0, 0], 4, 4),                               # :16-24
    ([1, 2, 3, 4], "assert", [2, 3, 4, 0, 0, 0, 0, 0], 3, 4),                             # :26-34
    ([1, 1, 3, 4], "asserteq", [3, 4, 0, 0, 0, 0, 0, 0], 2, 4),                           # :43-51
    ([1, 2], "dup", [1, 1, 2, 0, 0, 0, 0, 0], 3, 3),                                      # :106-114
    ([1, 2, 3, 4], "dup2", [1, 2, 1, 2, 3, 4, 0, 0], 6, 6),                               # :116-124
    ([1, 2, 3, 4], "dup4", [1, 2, 3, 4, 1, 2, 3, 4], 8, 8),                               # :126-134
    ([1, 2], "pad2", [0, 0, 1, 2, 0, 0, 0, 0], 4, 4),                                     # :136-144
    ([1, 2], "drop", [2, 0, 0, 0, 0, 0, 0, 0], 1, 2),                                     # :146-154
    ([1, 2, 3, 4, 5], "drop4", [5, 0, 0, 0, 0, 0, 0, 0], 1, 5),                           # :156-164
    ([1, 2, 3, 4], "swap", [2, 1, 3, 4, 0, 0, 0, 0], 4, 4),                               # :166-174
    ([1, 2, 3, 4], "swap2", [3, 4, 1, 2, 0, 0, 0, 0], 4, 4),                              # :176-184
    ([1, 2, 3, 4, 5, 6, 7, 8], "swap4", [5, 6, 7, 8, 1, 2, 3, 4], 8, 8),                  # :186-194
    ([1, 2, 3, 4], "roll4", [4, 1, 2, 3, 0, 0, 0, 0], 4, 4),                              # :196-204
    ([1, 2, 3, 4, 5, 6, 7, 8], "roll8", [8, 1, 2, 3, 4, 5, 6, 7], 8, 8),                  # :206-214
    ([1, 2], "add", [3, 0, 0, 0, 0, 0, 0, 0], 1, 2),                                      # :219-227
    ([2, 3], "mul", [6, 0, 0, 0, 0, 0, 0, 0], 1, 2),                                      # :229-237
    ([2, 3], "inv", [inv(2), 3, 0, 0, 0, 0, 0, 0], 2, 2),                                 # :239-247
    ([2, 3], "neg", [P - 2, 3, 0, 0, 0, 0, 0, 0], 2, 2),                                  # :256-264
    # stack/tests/conditional.rs
    ([2, 3, 0], "choose", [3, 0, 0, 0, 0, 0, 0, 0], 1, 3),                                # :6-13
    ([2, 3, 0, 4], "choose", [3, 4, 0, 0, 0, 0, 0, 0], 2, 4),                             # :15-20
    ([2, 3, 1, 4], "choose", [2, 4, 0, 0, 0, 0, 0, 0], 2, 4),                             # :22-28
    ([2, 3, 4, 5, 0, 6, 7], "choose2", [4, 5, 7, 0, 0, 0, 0, 0], 3, 7),                   # :38-46
    ([2, 3, 4, 5, 1, 6, 7], "choose2", [2, 3, 7, 0, 0, 0, 0, 0], 3, 7),                   # :48-54
    ([2, 3, 4, 5, 0, 6, 7], "cswap2", [2, 3, 4, 5, 7, 0, 0, 0], 5, 7),                    # :67-75
    ([2, 3, 4, 5, 1, 6, 7], "cswap2", [4, 5, 2, 3, 7, 0, 0, 0], 5, 7),                    # :77-83
]


@pytest.mark.parametrize("case", range(len(STACK_CASES)))
def test_stack_single_operations(oracle, case):
    inputs, op, state, depth, max_depth = STACK_CASES[case]
    assert _one(oracle, inputs, op) == (state, depth, max_depth)


STACK_PANICS = [
    ([2, 3, 4], "assert", "ASSERT failed at step 1"),                                          # mod.rs:36-41
    ([2, 3, 4], "asserteq", "ASSERTEQ failed at step 1"),                                      # :53-58
    ([0], "inv", "cannot compute INV of 0 at step 1"),                                         # :249-254
    ([2, 3], "not", "cannot compute NOT of a non-binary value at step 1"),                     # :283-288
    ([1, 3], "and", "cannot compute AND for a non-binary value at step 1"),                    # :307-312
    ([1, 3], "or", "cannot compute OR for a non-binary value at step 1"),                      # :331-336
    ([2, 3, 4], "choose", "CHOOSE on a non-binary condition at step 1"),                       # conditional.rs:30-35
    ([2, 3, 4, 5, 6, 8, 8], "choose2", "CHOOSE2 on a non-binary condition at step 1"),         # :56-61
    ([2, 3, 4, 5, 6, 8, 8], "cswap2", "CSWAP2 on a non-binary condition at step 1"),           # :85-90
]


@pytest.mark.parametrize("case", range(len(STACK_PANICS)))
def test_stack_panics(oracle, case):
    inputs, op, message = STACK_PANICS[case]
    with pytest.raises(RuntimeError, match=message):
        oracle.stack_run(inputs, (), (), [op])


def test_stack_sequences(oracle):
    O = oracle
    # push :63-71
    s, d, m = O.stack_run([], (), (), [("push", "push_value", 3)])
    assert (s[1], d[1], m[1]) == ([3, 0, 0, 0, 0, 0, 0, 0], 1, 1)
    # read :73-89
    s, d, m = O.stack_run([1], [2, 3], (), ["read", "read"])
    assert (s[1], d[1], m[1]) == ([2, 1, 0, 0, 0, 0, 0, 0], 2, 2) and (s[2], d[2], m[2]) == ([3, 2, 1, 0, 0, 0, 0, 0], 3, 3)
    # read2 :91-107
    s, d, m = O.stack_run([1], [2, 4], [3, 5], ["read2", "read2"])
    assert (s[1], d[1], m[1]) == ([3, 2, 1, 0, 0, 0, 0, 0], 3, 3) and (s[2], d[2], m[2]) == ([5, 4, 3, 2, 1, 0, 0, 0], 5, 5)
    # not :266-281
    s, d, m = O.stack_run([1, 2], (), (), ["not", "not"])
    assert s[1] == [0, 2, 0, 0, 0, 0, 0, 0] and s[2] == [1, 2, 0, 0, 0, 0, 0, 0] and d[1:] == [2, 2] and m[1:] == [2, 2]
    # and :290-305, or :314-329
    s, d, m = O.stack_run([1, 1, 0], (), (), ["and", "and"])
    assert s[1] == [1, 0, 0, 0, 0, 0, 0, 0] and s[2] == [0] * 8 and d[1:] == [2, 1] and m[1:] == [3, 3]
    s, d, m = O.stack_run([0, 0, 1], (), (), ["or", "or"])
    assert s[1] == [0, 1, 0, 0, 0, 0, 0, 0] and s[2] == [1, 0, 0, 0, 0, 0, 0, 0] and d[1:] == [2, 1] and m[1:] == [3, 3]
    # rescr :341-357
    s, d, m = O.stack_run([0, 0, 1, 2, 3, 4], (), (), ["rescr", "rescr"])
    e1 = O.hasher_round([0, 0, 1, 2, 3, 4], 0)
    e2 = O.hasher_round(e1, 1)
    assert s[1] == e1 + [0, 0] and s[2] == e2 + [0, 0] and (d[2], m[2]) == (6, 6)
    # eq, eq_with_hint: comparisons.rs:8-45
    inv_diff = inv((1 - 4) % P)
    for ops, tape in ((["read", "eq", "read", "eq"], [0, inv_diff]),
                      ([("read", "eq_start"), "eq", ("read", "eq_start"), "eq"], [])):
        s, d, m = O.stack_run([3, 3, 4, 5], tape, (), ops)
        assert s[2] == [1, 4, 5, 0, 0, 0, 0, 0] and (d[2], m[2]) == (3, 5)
        assert s[4] == [0, 5, 0, 0, 0, 0, 0, 0] and (d[4], m[4]) == (2, 5)


def _cmp_inputs(a, b, size):
    return [(a >> i) & 1 for i in range(size)][::-1], [(b >> i) & 1 for i in range(size)][::-1]     # comparisons.rs:257-269


LT_FINALE = ["drop4", "pad2", "swap4", "roll4", "asserteq", "asserteq", "dup", "drop4"]            # comparisons.rs:271-280
GT_FINALE = ["drop4", "pad2", "swap4", "roll4", "asserteq", "asserteq", "roll4", "dup", "drop4"]   # :282-292


@pytest.mark.parametrize("bits", [128, 64])
def test_stack_cmp_binacc(oracle, bits):
    """comparisons.rs:50-255 with field::rand() replaced by seeded draws (plus the equal and adjacent cases)."""
    O = oracle
    rnd = random.Random(1000 + bits)
    top = P if bits == 128 else 1 << 64
    pairs = [(rnd.randrange(top), rnd.randrange(top)) for _ in range(3)] + [(5, 5), (7, 6), (0, top - 1)]
    for a, b in pairs:
        ia, ib = _cmp_inputs(a, b, bits)
        ops = ["pad2", ("push", "push_value", 1 << (bits - 1))] + ["cmp"] * bits                  # cmp_128 :50-80, cmp_64 :82-112
        s, d, m = O.stack_run([0, 0, 0, 0, 0, a, b], ia, ib, ops, init_len=256)
        for i in range(2, 2 + bits):
            gt, lt = s[i][4], s[i][5]
            assert s[i + 1][3] == (1 - gt) * (1 - lt) % P
        lt, gt = int(a < b), int(a > b)                 # the reference draws a != b; equal values leave both flags clear
        assert s[2 + bits][4:8] == [gt, lt, b, a]
    if bits == 128:
        for a, b in pairs:                                                                         # lt :117-141, gt :143-167
            ia, ib = _cmp_inputs(a, b, 128)
            head = ["pad2", "pad2", ("push", "push_value", 1 << 127)] + ["cmp"] * 128
            s, d, m = O.stack_run([0, 0, 0, a, b, 7, 11], ia, ib, head + LT_FINALE, init_len=256)
            assert s[-1] == [1 if a < b else 0, 7, 11] + [0] * 9
            s, d, m = O.stack_run([0, 0, 0, a, b, 7, 11], ia, ib, head + GT_FINALE, init_len=256)
            assert s[-1] == [1 if a > b else 0, 7, 11] + [0] * 9
    for x in [rnd.randrange(top) for _ in range(3)] + [0, 1, top - 1]:
        tape = [(x >> (bits - 1 - i)) & 1 for i in range(bits)][::-1]                              # binacc_128 :172-196, binacc_64 :198-222
        s, d, m = O.stack_run([0, 0, 1, 0, x, 7, 11], tape, (), ["binacc"] * bits + ["drop"] * 3, init_len=256)
        assert s[bits + 3] == [x, x, 7, 11, 0, 0, 0, 0]
        if bits == 128:                                                                            # isodd_128 :224-255
            ops = ["binacc", "swap2", "roll4", "dup"] + ["binacc"] * 127 + ["drop", "drop", "swap", "roll4", "asserteq", "drop"]
            s, d, m = O.stack_run([0, 0, 1, 0, x, 7, 11], tape, (), ops, init_len=256)
            assert s[137] == [x & 1, 7, 11, 0, 0, 0, 0, 0]


def test_stack_hints(oracle):
    """CmpStart / RcStart / PmpathStart fill the tapes themselves (stack/mod.rs:477-487, 540-549, 212-232)."""
    O = oracle
    a, b = 0xDEADBEEF12345, 0xDEADBEEF12346
    head = ["pad2", "pad2", "pad2", "dup", ("push", "push_value", 1 << 63)]                    # the prologue of gt / lt (parsers.rs:282-284)
    s, d, m = O.stack_run([a, b], (), (), head + [("cmp", "cmp_start", 64)] + ["cmp"] * 63, init_len=128)
    assert s[-1][4:10] == [0, 1, b, a, a, b]
    s, d, m = O.stack_run([0, 0, 1, 0, a], (), (), [("binacc", "rc_start", 52)] + ["binacc"] * 51, init_len=64)
    assert s[-1][3] == a and s[-1][2] == 1 << 52


# ---------------------------------------------------------------------------------------------------------------------------------
# processor: src/processor/mod.rs:237-346
# ---------------------------------------------------------------------------------------------------------------------------------
def _final(t):
    last = t.row(t.length - 1)
    c, l = t.ctx_depth, t.loop_depth
    return {"op_counter": last[0], "hash_ok": t.trace_hash() == t.program_hash, "bits": last[5:15], "ctx": last[15:15 + c],
            "loop": last[15 + c:15 + c + l], "stack": t.user_stack(t.length - 1)}


def test_execute_if_else(oracle):
    O = oracle
    src = "begin read if.true add push.3 else push.7 add push.8 end mul end"                     # processor/mod.rs:237-280
    t = O.Trace(src, [5, 3], [1])
    assert (t.length, t.width) == (128, 19)
    f = _final(t)
    assert f == {"op_counter": 76, "hash_ok": True, "bits": [1] * 10, "ctx": [0], "loop": [], "stack": [24, 0, 0, 0, 0, 0, 0, 0]}
    assert (t.ctx_depth, t.loop_depth) == (1, 0)          # state.loop_stack() == [0] there is the zero-padded view (trace_state.rs:55)
    t = O.Trace(src, [5, 3], [0])
    assert (t.length, t.width) == (128, 19)
    assert _final(t) == {"op_counter": 92, "hash_ok": True, "bits": [1] * 10, "ctx": [0], "loop": [], "stack": [96, 3, 0, 0, 0, 0, 0, 0]}


def test_execute_loop(oracle):
    O = oracle
    src = "begin mul read while.true dup mul read end end"                                         # processor/mod.rs:282-346
    t = O.Trace(src, [5, 3], [0])                                                                  # loop not entered
    assert (t.length, t.width, t.ctx_depth, t.loop_depth) == (64, 18, 1, 0)
    assert _final(t) == {"op_counter": 60, "hash_ok": True, "bits": [1] * 10, "ctx": [0], "loop": [], "stack": [15, 0, 0, 0, 0, 0, 0, 0]}
    t = O.Trace(src, [5, 3], [1, 0])                                                               # one iteration
    assert (t.length, t.width, t.ctx_depth, t.loop_depth) == (128, 19, 1, 1)
    assert _final(t) == {"op_counter": 75, "hash_ok": True, "bits": [1] * 10, "ctx": [0], "loop": [0], "stack": [225, 0, 0, 0, 0, 0, 0, 0]}
    t = O.Trace(src, [5, 3], [1, 1, 1, 1, 1, 0])                                                   # five iterations
    assert (t.length, t.width, t.ctx_depth, t.loop_depth) == (256, 19, 1, 1)
    assert _final(t) == {"op_counter": 135, "hash_ok": True, "bits": [1] * 10, "ctx": [0], "loop": [0],
                         "stack": [43143988327398919500410556793212890625, 0, 0, 0, 0, 0, 0, 0]}


# ---------------------------------------------------------------------------------------------------------------------------------
# end to end: src/tests/mod.rs:66-315, src/tests/comparisons.rs:8-107, src/examples/*.rs
# ---------------------------------------------------------------------------------------------------------------------------------
def prove_and_verify(O, t, num_outputs, expected_outputs):
    """lib.rs:30-64 execute() + lib.rs:71-74 verify(): outputs, the two asserts of execute(), the prover (whose constraint check at
    evaluator.rs:152-158 panics unless every transition constraint vanishes on every trace row), the verifier."""
    outputs = t.outputs(num_outputs)
    assert outputs == expected_outputs
    assert t.row(t.length - 1)[0] >= 16                         # lib.rs:49-52
    assert t.trace_hash() == t.program_hash                     # lib.rs:55-59
    p = O.Prover.from_trace(t, num_outputs, grinding=8)
    assert p.outputs == outputs
    proof = p.prove()
    assert p.get_u64("constraints_ok") == [1]
    assert O.verify(proof, t.program_hash, t.public_inputs, outputs) == (True, "")
    return proof


def pad_ops(ops, n):
    return ops + ["noop"] * (n - len(ops))


ISA_PROGRAMS = {
    # name: (ops, push values, public inputs, tape a, tape b, num_outputs, expected outputs)
    "stack_manipulation": (                                                                       # tests/mod.rs:65-88
        ["begin"] + ["noop"] * 7 + ["swap", "swap2", "swap4", "roll4", "roll8", "dup", "add", "pad2",
                                    "push", "swap4", "drop4", "dup2", "swap4", "add", "add", "dup4",
                                    "push", "add", "add", "add", "add", "noop", "noop"],
        [11, 12], [7, 6, 5, 4, 3, 2, 1, 0], [], [], 8, [46, 19, 4, 11, 0, 11, 0, 6]),
    "choose": (pad_ops(["begin", "choose", "choose"], 15), [], [3, 4, 1, 5, 0, 6, 7, 8], [], [], 8, [5, 6, 7, 8, 0, 0, 0, 0]),   # :90-109
    "choose2": (pad_ops(pad_ops(pad_ops(["begin"], 8) + ["push"], 16) + ["push", "choose2", "choose2"], 31),                      # :111-131
                [3, 4], [5, 6, 1, 0, 7, 8, 0, 0], [], [], 8, [7, 8, 0, 0, 0, 0, 0, 0]),
    "cswap2": (pad_ops(["begin", "cswap2", "pad2", "swap4", "cswap2"], 15), [], [3, 4, 1, 2, 1, 0, 5, 6], [], [], 8,              # :133-150
               [3, 4, 5, 6, 1, 2, 0, 0]),
    "math": (pad_ops(["begin", "add", "mul", "inv", "neg", "swap", "not"], 15), [], [7, 6, 5, 0, 2, 3], [], [], 2,                # :170-190
             [1, (P - inv(65)) % P]),
    "bool": (pad_ops(["begin", "not", "or", "or", "and", "and", "not"], 15), [], [1, 0, 1, 1, 0], [], [], 1, [1]),                # :192-212
    "read": (pad_ops(pad_ops(["begin", "read", "read2"], 8) + ["push"], 15), [5], [1], [2, 3], [4], 5, [5, 4, 3, 2, 1]),         # :277-294
    "assert": (pad_ops(["begin", "assert", "noop", "asserteq"], 15), [], [1, 3, 3], [], [], 2, [0, 0]),                          # :296-315
    "eq": (pad_ops(["begin", "read", "eq", "swap2", "read", "eq"], 15), [], [1, 2, 3, 4, 4], [inv(P - 1), 1], [], 3, [1, 0, 3]),  # comparisons.rs:8-29
}


@pytest.mark.parametrize("name", sorted(ISA_PROGRAMS))
def test_end_to_end_fixtures(oracle, name):
    ops, push, pub, a, b, nout, expected = ISA_PROGRAMS[name]
    t = oracle.Trace.from_ops(ops, push, pub, a, b)
    prove_and_verify(oracle, t, nout, expected)


def test_selection_operations_panic(oracle):                                                      # tests/mod.rs:152-168
    with pytest.raises(RuntimeError, match="CHOOSE on a non-binary condition at step 2"):
        oracle.Trace.from_ops(pad_ops(["begin", "choose", "choose"], 15), [], [3, 4, 2, 5, 0, 6, 7, 8])


def test_hash_operations(oracle):                                                                 # tests/mod.rs:214-275
    O = oracle
    single = pad_ops(["begin"], 16) + ["rescr"] * 10 + ["drop"] * 4 + ["noop"]
    expected = O.hasher_digest([1, 2, 3, 4])[::-1]
    prove_and_verify(O, O.Trace.from_ops(single, [], [0, 0, 4, 3, 2, 1]), 2, expected)
    double = (pad_ops(["begin"], 16) + ["rescr"] * 10 + ["drop4", "noop", "pad2", "dup2", "noop", "noop"]
              + ["rescr"] * 10 + ["drop4"] + ["noop"] * 4)
    assert len(double) == 47
    expected = O.hasher_digest(O.hasher_digest([1, 2, 3, 4]))[::-1]
    prove_and_verify(O, O.Trace.from_ops(double, [], [0, 0, 4, 3, 2, 1]), 2, expected)


def test_cmp_operation(oracle):                                                                   # comparisons.rs:31-67
    O = oracle
    rnd = random.Random(31)
    for a, b in ((rnd.randrange(P), rnd.randrange(P)), (12345, 12345)):
        ia, ib = _cmp_inputs(a, b, 128)
        ops = pad_ops(pad_ops(["begin", "pad2"], 8) + ["push"] + ["cmp"] * 128 + ["drop4"], 255)
        t = O.Trace.from_ops(ops, [1 << 127], [0, 0, 0, 0, 0, a, b], ia, ib)
        prove_and_verify(O, t, 4, [int(a > b), int(a < b), b, a])


def test_binacc_operation(oracle):                                                                # comparisons.rs:69-104
    O = oracle
    a = random.Random(69).randrange(P)
    tape = [(a >> (127 - i)) & 1 for i in range(128)][::-1]
    ops = pad_ops(["begin"] + ["binacc"] * 128 + ["drop"] * 3, 255)
    t = O.Trace.from_ops(ops, [], [0, 0, 1, 0, a], tape, [])
    prove_and_verify(O, t, 2, [a, a])


def test_example_conditional(oracle):                                                             # examples/conditional.rs:4-45
    O = oracle
    src = "begin push.3 push.5 read if.true add else mul end end"
    for flag, expected in ((0, 15), (1, 8)):
        prove_and_verify(O, O.Trace(src, [], [flag]), 1, [expected])


def test_example_comparison(oracle):                                                              # examples/comparison.rs:4-50
    O = oracle
    src = "begin push.9 read dup.2 lt.128 if.true mul else add end dup isodd.128 end"
    for value in (6, 11):
        expected = 9 * value % P if value < 9 else (9 + value) % P
        prove_and_verify(O, O.Trace(src, [], [value]), 2, [expected & 1, expected])


def _collatz_steps(value):                                                                        # examples/collatz.rs:48-62
    i = 0
    while value != 1:
        value = value * inv(2) % P if value & 1 == 0 else (value * 3 + 1) % P
        i += 1
    return i


def test_example_collatz(oracle):                                                                 # examples/collatz.rs:4-45
    O = oracle
    src = """begin pad read dup push.1 ne
             while.true
                 swap push.1 add swap dup isodd.128
                 if.true push.3 mul push.1 add else push.2 div end
                 dup push.1 ne
             end
             swap end"""
    t = O.Trace(src, [], [3])
    assert (t.ctx_depth, t.loop_depth) == (2, 1)
    assert t.outputs(1) == [_collatz_steps(3)] == [7]
    if t.length <= 4096:
        prove_and_verify(O, t, 1, [7])
    assert O.Trace(src, [], [1]).outputs(1) == [0]                        # loop not entered
    t = O.Trace(src, [], [27])                                            # the classic long one: 111 steps, 2^15 rows
    assert t.outputs(1) == [_collatz_steps(27)] == [111] and (t.length, t.loop_depth) == (1 << 15, 1) and t.trace_hash() == t.program_hash


def test_example_range(oracle):                                                                   # examples/range.rs:4-71
    O = oracle
    values = [5, (1 << 63) + 17, (1 << 63) - 1, (1 << 64) - 1]
    src = "begin " + "read rc.63 add " * len(values) + "end"
    t = O.Trace(src, [0], values)
    assert t.outputs(1) == [2]
    prove_and_verify(O, O.Trace("begin read rc.63 add read rc.63 add end", [0], values[:2]), 1, [1])


def _merkle_root(O, path, index):                                                                 # examples/merkle.rs:112-145
    n = len(path[0])
    r = index & 1
    v = O.hasher_digest([path[0][r], path[1][r], path[0][1 - r], path[1][1 - r]])
    index = (index + (1 << (n - 1))) >> 1
    for i in range(2, n):
        v = O.hasher_digest([v[0], v[1], path[0][i], path[1][i]] if index & 1 == 0 else [path[0][i], path[1][i], v[0], v[1]])
        index >>= 1
    return v


def test_example_merkle(oracle):                                                                  # examples/merkle.rs:4-110
    O = oracle
    rnd = random.Random(4)
    for depth, leaf_index in ((3, 2), (4, 5)):
        path = [[rnd.randrange(P) for _ in range(depth)] for _ in range(2)]
        src = "begin read.ab dup.2 smpath.%d swap.2 push.%d roll.4 swap swap.2 pmpath.%d end" % (depth, leaf_index, depth)
        a, b = [path[0][0]], [path[1][0]]
        index = leaf_index + (1 << (depth - 1))
        for i in range(1, depth):
            a += [0, path[0][i]]; b += [index & 1, path[1][i]]
            index >>= 1
        for i in range(1, depth):
            a.append(path[0][i]); b.append(path[1][i])
        root = _merkle_root(O, path, leaf_index)
        expected = (root + root)[::-1]
        t = O.Trace(src, [], a, b)
        assert t.outputs(4) == expected
        if depth == 3:
            prove_and_verify(O, t, 4, expected)


def test_committed_isa_fixtures_are_what_the_oracle_produces(oracle):
    """tests/golden/isa_traces.npz / isa_proof_digests.json (the GPU test that uses them runs without the oracle) against a fresh run of the
    script that made them: a change of the oracle's VM or prover shows up here."""
    import importlib.util
    import json
    import os
    import numpy as np
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_isa_golden", os.path.join(here, "make_isa_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    arrays, cases = mod.build()
    committed = json.load(open(os.path.join(here, "isa_proof_digests.json")))["cases"]
    assert cases == committed
    stored = np.load(os.path.join(here, "isa_traces.npz"))
    for name, cols in arrays.items():
        assert (stored[name] == cols).all(), name
