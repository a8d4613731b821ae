"""GPU parity tests proper: every phase of the HIP prover path (through the C-ABI of libdistaff_hip.so) against the CPU
oracle on the same inputs and the same Fiat-Shamir challenges -- bit-exact, as the path is integer arithmetic."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_which_library_is_bound():
    """the parity tests run on the test build (tests/conftest.py) -- or, when tests/test_product_library.py re-runs a selection with
    DISTAFF_PRODUCT_ONLY=1, on the product library"""
    import os
    import distaff_amd as D
    if os.environ.get("DISTAFF_HIP_LIB"):
        return
    assert D.load().dst_test_hooks() == (0 if os.environ.get("DISTAFF_PRODUCT_ONLY") == "1" else 1)


def _ctx(D, trace, **kw):
    log_n = trace.length.bit_length() - 1
    return D.Context(log_n, trace.width, trace.ctx_depth, trace.loop_depth, **kw)


def _check_all_phases(O, D, trace, num_outputs=1, log_blowup=5, num_queries=50, grinding=12):
    ext = 1 << log_blowup
    op = O.Prover.from_trace(trace, num_outputs, ext=ext, num_queries=num_queries, grinding=grinding)
    for k in range(1, 10):
        op.step(k)
    ctx = _ctx(D, trace, log_blowup=log_blowup, num_queries=num_queries, grinding=grinding)
    ctx.upload(trace.columns)
    W, n, N = trace.width, trace.length, trace.length * ext

    # steps 1-2
    root = ctx.commit_trace()
    assert (ctx.read_elements("polys").reshape(W, n, 2) == op.get("polys")).all(), "polys"
    regs = op.get("registers")
    for c in range(W):
        assert (ctx.read_elements("lde", c) == regs[c]).all(), "lde register %d" % c
    assert ctx.read("trace_leaves").tobytes() == op.get_bytes("trace_leaves"), "trace leaves"
    assert ctx.read("trace_nodes").tobytes()[32:] == op.get_bytes("trace_nodes")[32:], "trace nodes"
    assert root == op.get_bytes("roots")[:32]

    # steps 3-5 with the oracle's coefficient draws
    croot = ctx.eval_constraints(trace.public_inputs, op.outputs, op.get("constraint_draws"))
    # the boundary combinations are written in coefficient form (no evaluation vectors) unless DISTAFF_BOUNDARY=eval
    import os
    vectors = (("ceval_i", "i_evaluations"), ("ceval_f", "f_evaluations"), ("ceval_t", "t_evaluations"))
    for name, oname in (vectors if os.environ.get("DISTAFF_BOUNDARY") == "eval" else vectors[2:]):
        assert (ctx.read_elements(name) == op.get(oname)).all(), name
    assert (ctx.read_elements("cpoly") == op.get("constraint_poly")).all(), "constraint poly"
    assert (ctx.read_elements("cevals") == op.get("constraint_evaluations")).all(), "constraint evaluations"
    assert ctx.read("cnodes").tobytes()[32:] == op.get_bytes("constraint_nodes")[32:], "constraint nodes"
    assert croot == op.get_bytes("roots")[32:]

    # step 6
    z1, z2 = ctx.compose(op.get("deep_draws"))
    assert (z1 == op.get("trace_at_z1")).all() and (z2 == op.get("trace_at_z2")).all(), "deep values"
    assert (ctx.read_elements("comp_poly") == op.get("composition_poly")).all(), "composition poly"
    assert (ctx.read_elements("comp_evals") == op.get("composed_evaluations")).all(), "composition evaluations"

    # step 7
    xs = O.to_ints(op.get("fri_special_xs")) if op.get_u64("fri_layers")[0] > 1 else []
    layers = op.get_u64("fri_layers")[0]
    roots = b""
    for d in range(layers):
        r, more = ctx.fri_commit_layer()
        roots += r
        assert ctx.read("fri_nodes", d).tobytes()[32:] == op.get_bytes("fri_nodes", d)[32:], "fri nodes %d" % d
        assert more == (d + 1 < layers)
        if more:
            ctx.fri_fold(xs[d])
            nxt = op.get("fri_values", d + 1)          # rows (e[r], e[r+R], e[r+2R], e[r+3R]) of the next layer
            R = nxt.shape[0]
            got = ctx.read_elements("fri_evals", d + 1).reshape(4, R, 2).transpose(1, 0, 2)
            assert (got == nxt).all(), "fri layer %d" % (d + 1)
    assert roots == op.get_bytes("fri_roots")

    # step 8
    seeds = op.get_bytes("query_seeds")
    seed1, nonce = ctx.pow_grind(seeds[:32], grinding)
    assert nonce == op.get_u64("pow_nonce")[0] and seed1 == seeds[32:]
    assert D.query_positions(seed1, N, ext, num_queries) == op.get_u64("positions")

    # step 9: the serialised proof is byte-identical and the oracle's verifier accepts it
    proof = ctx.build_proof(op.get_u64("positions"), nonce)
    assert proof == op.get_bytes("proof"), "proof bytes"
    assert O.verify(proof, trace.program_hash, trace.public_inputs, op.outputs) == (True, "")

    # dst_prove (library-side Fiat-Shamir) reproduces the same proof
    ctx.upload(trace.columns)
    assert ctx.prove(trace.public_inputs, op.outputs) == proof
    ctx.close()


@pytest.mark.parametrize("log_n", [7, 10, 12])
def test_fibonacci_all_phases(oracle, log_n):
    import distaff_amd as D
    _check_all_phases(oracle, D, oracle.fibonacci_trace(1 << log_n))


def test_trace_in_its_own_buffer(oracle, monkeypatch):
    """A context that owns coset 0 of the extension keeps the trace in the coset-0 slots of the extension buffer (nothing is copied);
    DISTAFF_TRACE_BUFFER=1 gives the trace its own buffer and copies it into coset 0, as contexts of the other ranks of a sharded proof
    do for their interpolation input.  Same intermediates, same proof."""
    import distaff_amd as D
    monkeypatch.setenv("DISTAFF_TRACE_BUFFER", "1")
    _check_all_phases(oracle, D, oracle.fibonacci_trace(1 << 10))


def test_synthetic_division_by_power_tables(oracle, monkeypatch):
    """k_syn_div's first formulation (scale by b^t, additive suffix scan, scale by b^-(i+1)) is kept beside the blocked one; both give the
    oracle's quotients (polynom.rs:190) -- at 2^12 steps the blocked form recurses once (16 chunks of 2048 coefficients)."""
    import distaff_amd as D
    monkeypatch.setenv("DISTAFF_SYN_DIV_TABLES", "1")
    _check_all_phases(oracle, D, oracle.fibonacci_trace(1 << 12))


@pytest.mark.parametrize("instance", ["", "small", "deep", "generic"])
def test_boundary_constraints_by_evaluation(oracle, monkeypatch, instance):
    """The evaluate-and-interpolate route of the two boundary combinations (the reference's own, constraint_table.rs:54-62): its
    evaluation vectors equal the oracle's; the default route writes the same polynomials in coefficient form, and every other test
    compares the constraint polynomial and everything downstream."""
    import distaff_amd as D
    monkeypatch.setenv("DISTAFF_BOUNDARY", "eval")
    if instance:
        monkeypatch.setenv("DISTAFF_AIR", instance)
    _check_all_phases(oracle, D, oracle.fibonacci_trace(1 << 8))
    _check_all_phases(oracle, D, oracle.Trace("begin dup.4 add mul swap.2 add drop drop block push.9 mul end end", [1, 2, 3, 4]), num_outputs=2)


def test_merkle_levels_two_per_launch_on_small_trees(oracle, monkeypatch):
    """The wide levels of every tree are built two per launch (a lane hashes four children into two parents and the grandparent); trees
    of the sizes the oracle can rebuild are below the width at which the library switches to that form, so it is forced here
    (DISTAFF_MERKLE_LEVEL2_LOG): all node arrays of all trees against the oracle, single context and sharded (local levels)."""
    import distaff_amd as D
    monkeypatch.setenv("DISTAFF_MERKLE_LEVEL2_LOG", "3")
    _check_all_phases(oracle, D, oracle.fibonacci_trace(1 << 10))
    t = oracle.fibonacci_trace(1 << 9)
    op = oracle.Prover.from_trace(t, 1, grinding=8)
    expected = op.prove()
    for world in (2, 8):
        ctxs = []
        for r in range(world):
            ctx = D.Context(9, t.width, t.ctx_depth, t.loop_depth, rank=r, world=world, grinding=8)
            ctx.upload(t.columns)
            ctxs.append(ctx)
        assert D.prove_sharded_local(ctxs, t.public_inputs, op.outputs) == expected
        for ctx in ctxs:
            ctx.close()


def test_all_phases_with_register_pre_stages(oracle, monkeypatch):
    """DISTAFF_NTT=pre: every transform of a proof (interpolation, extension, inverse coset transforms of the combination, constraint and
    composition extensions) through the pre-stage instances the library takes by itself at 2^21 / 2^22 steps -- all intermediates
    against the oracle, on one context and sharded over four ranks."""
    import distaff_amd as D
    monkeypatch.setenv("DISTAFF_NTT", "pre")
    _check_all_phases(oracle, D, oracle.fibonacci_trace(1 << 10))
    t = oracle.fibonacci_trace(1 << 10)
    op = oracle.Prover.from_trace(t, 1, grinding=8)
    expected = op.prove()
    ctxs = []
    for r in range(4):
        ctx = D.Context(10, t.width, t.ctx_depth, t.loop_depth, rank=r, world=4, grinding=8)
        ctx.upload_owned(t.columns)
        ctxs.append(ctx)
    assert D.prove_sharded_local(ctxs, t.public_inputs, op.outputs) == expected
    for ctx in ctxs:
        ctx.close()


def test_combination_and_composition_as_whole_array_steps(oracle, monkeypatch):
    """DISTAFF_COMBINE=steps: combine_polys and the DEEP composition as the reference's sequence of whole-array steps (8n-coefficient boundary
    polynomials, three divisions, additions; copy / C(z) / division / multiply-adds) instead of the fused passes the library takes by
    default -- every intermediate and the proof must be the same, on one context and through the sharded prover."""
    import distaff_amd as D
    monkeypatch.setenv("DISTAFF_COMBINE", "steps")
    _check_all_phases(oracle, D, oracle.fibonacci_trace(1 << 9))
    t = oracle.fibonacci_trace(1 << 8)
    op = oracle.Prover.from_trace(t, 1, grinding=8)
    expected = op.prove()
    ctxs = []
    for r in range(4):
        ctx = D.Context(8, t.width, t.ctx_depth, t.loop_depth, rank=r, world=4, grinding=8)
        ctx.upload_owned(t.columns)
        ctxs.append(ctx)
    assert D.prove_sharded_local(ctxs, t.public_inputs, op.outputs) == expected
    for ctx in ctxs:
        ctx.close()


@pytest.mark.parametrize("shape", ["fibonacci", "w17", "w18"])
def test_trace_from_pinned_host_memory(oracle, shape):
    """dst_trace_upload_async: the registers arrive group by group (the first group is short) and every group is interpolated and extended
    as it lands; the proof is the one of the synchronous upload.  Register counts 20 (1 + 4 + 4 + 4 + 4 + 3), 17 (1 + 16 / 4) and 18 (2 + ...)."""
    import distaff_amd as D
    O = oracle
    if shape == "fibonacci":
        t, outs = O.fibonacci_trace(1 << 10), 1
    elif shape == "w17":
        t, outs = O.Trace("begin add push.5 mul push.7 end", [1, 2]), 2
    else:
        t, outs = O.Trace("begin add block push.5 mul push.7 end end", [1, 2]), 2
    op = O.Prover.from_trace(t, outs, ext=32, num_queries=50, grinding=12)
    for k in range(1, 10):
        op.step(k)
    ctx = _ctx(D, t, log_blowup=5, num_queries=50, grinding=12)
    ctx.upload(t.columns)
    expected = ctx.prove(t.public_inputs, op.outputs)
    assert expected == op.get_bytes("proof")
    table, handle = ctx.pinned_trace(t.columns)
    try:
        for _ in range(2):                                   # the second round re-uses the events of the first
            ctx.upload_async(table)
            assert ctx.prove(t.public_inputs, op.outputs) == expected
    finally:
        ctx.release_pinned(handle)
    ctx.close()


def test_other_program_shapes(oracle):
    import distaff_amd as D
    O = oracle
    _check_all_phases(O, D, O.Trace("begin add push.5 mul push.7 end", [1, 2]), num_outputs=2)             # W = 17, no context register
    _check_all_phases(O, D, O.Trace("begin add block push.5 mul push.7 end end", [1, 2]), num_outputs=2)  # W = 18
    _check_all_phases(O, D, O.Trace("begin push.3 push.4 push.5 push.6 push.7 push.8 push.9 push.10 push.11 mul add block swap.2 dup.2 drop add end mul end", [1, 2]),
                      num_outputs=3)                                                                      # stack deeper than 8


DEEP_PROGRAMS = [   # (source, inputs, outputs): stack deeper than 8 and / or more than two context registers -> the any-shape instances
    ("begin push.3 push.4 push.5 push.6 push.7 push.8 push.9 push.10 push.11 mul add block swap.2 dup.2 drop add end mul end", [1, 2], 3),     # depth 11
    ("begin push.3 push.4 push.5 push.6 push.7 push.8 push.9 push.10 push.11 push.12 push.13 push.14 dup.4 dup.4 dup.2 swap.2 mul add block swap.2 dup.2 drop add "
     "block dup.1 mul swap.1 drop.4 block add mul end end end drop.4 mul add end", [1, 2, 3], 4),                                                # depth 25, 3 context registers
    ("begin add block push.5 mul block dup.2 add block mul push.7 end end end add end", [1, 2, 3, 4], 2),                                         # depth 6, 3 context registers
]


@pytest.mark.parametrize("instance", ["", "generic"])
def test_deep_stacks_and_nested_blocks(oracle, monkeypatch, instance):
    """Shapes outside the specialised instances: by default the nested-sum instance whose slots 8.. are seven flag sums times shifted
    differences, and the per-operation formulation (DISTAFF_AIR=generic) as an independent statement of the same constraints; both
    must reproduce the oracle's evaluations on the whole 8n domain (where every operation's flag is non-zero), and the proof."""
    import distaff_amd as D
    if instance:
        monkeypatch.setenv("DISTAFF_AIR", instance)
    for src, inputs, num_outputs in DEEP_PROGRAMS:
        _check_all_phases(oracle, D, oracle.Trace(src, inputs), num_outputs=num_outputs)


def _pad_ops(ops, n):
    return ops + ["noop"] * (n - len(ops))


def _isa_traces(O):
    """Valid traces covering EVERY user operation and every flow operation (BEGIN / TEND / FEND / LOOP / WRAP / BREAK / VOID): the
    reference's own end-to-end fixtures (src/tests/mod.rs:66-315, src/tests/comparisons.rs, src/processor/mod.rs:237-346) and its example
    programs, as tests/test_oracle_isa.py pins them on the oracle.  name -> (trace, number of outputs)"""
    import random
    P = 2**128 - 45 * 2**40 + 1
    rnd = random.Random(5)
    a, b = rnd.randrange(P), rnd.randrange(P)
    bits = lambda v, n: [(v >> i) & 1 for i in range(n)][::-1]
    T = {}
    T["stack_manipulation"] = (O.Trace.from_ops(
        ["begin"] + ["noop"] * 7 + ["swap", "swap2", "swap4", "roll4", "roll8", "dup", "add", "pad2", "push", "swap4", "drop4", "dup2", "swap4", "add", "add", "dup4",
                                    "push", "add", "add", "add", "add", "noop", "noop"], [11, 12], [7, 6, 5, 4, 3, 2, 1, 0]), 8)
    T["choose"] = (O.Trace.from_ops(_pad_ops(["begin", "choose", "choose"], 15), [], [3, 4, 1, 5, 0, 6, 7, 8]), 8)
    T["choose2"] = (O.Trace.from_ops(_pad_ops(_pad_ops(_pad_ops(["begin"], 8) + ["push"], 16) + ["push", "choose2", "choose2"], 31), [3, 4], [5, 6, 1, 0, 7, 8, 0, 0]), 8)
    T["cswap2"] = (O.Trace.from_ops(_pad_ops(["begin", "cswap2", "pad2", "swap4", "cswap2"], 15), [], [3, 4, 1, 2, 1, 0, 5, 6]), 8)
    T["math_inv_neg_not"] = (O.Trace.from_ops(_pad_ops(["begin", "add", "mul", "inv", "neg", "swap", "not"], 15), [], [7, 6, 5, 0, 2, 3]), 2)
    T["bool_and_or"] = (O.Trace.from_ops(_pad_ops(["begin", "not", "or", "or", "and", "and", "not"], 15), [], [1, 0, 1, 1, 0]), 1)
    T["read_read2"] = (O.Trace.from_ops(_pad_ops(_pad_ops(["begin", "read", "read2"], 8) + ["push"], 15), [5], [1], [2, 3], [4]), 5)
    T["assert_asserteq"] = (O.Trace.from_ops(_pad_ops(["begin", "assert", "noop", "asserteq"], 15), [], [1, 3, 3]), 2)
    T["eq"] = (O.Trace.from_ops(_pad_ops(["begin", "read", "eq", "swap2", "read", "eq"], 15), [], [1, 2, 3, 4, 4], [pow(P - 1, P - 2, P), 1]), 3)
    T["rescr_double_hash"] = (O.Trace.from_ops(_pad_ops(["begin"], 16) + ["rescr"] * 10 + ["drop4", "noop", "pad2", "dup2", "noop", "noop"] + ["rescr"] * 10 + ["drop4"]
                                               + ["noop"] * 4, [], [0, 0, 4, 3, 2, 1]), 2)
    T["cmp_128"] = (O.Trace.from_ops(_pad_ops(_pad_ops(["begin", "pad2"], 8) + ["push"] + ["cmp"] * 128 + ["drop4"], 255), [1 << 127], [0, 0, 0, 0, 0, a, b],
                                     bits(a, 128), bits(b, 128)), 4)
    T["binacc_128"] = (O.Trace.from_ops(_pad_ops(["begin"] + ["binacc"] * 128 + ["drop"] * 3, 255), [], [0, 0, 1, 0, a], [(a >> (127 - i)) & 1 for i in range(128)][::-1]), 2)
    T["if_true"] = (O.Trace("begin read if.true add push.3 else push.7 add push.8 end mul end", [5, 3], [1]), 2)
    T["if_false"] = (O.Trace("begin read if.true add push.3 else push.7 add push.8 end mul end", [5, 3], [0]), 2)
    T["while_skipped"] = (O.Trace("begin mul read while.true dup mul read end end", [5, 3], [0]), 1)
    T["while_5_iterations"] = (O.Trace("begin mul read while.true dup mul read end end", [5, 3], [1, 1, 1, 1, 1, 0]), 1)
    T["nested_loops"] = (O.Trace("begin read while.true read while.true push.2 mul read end push.3 add read end end", [1], [1, 1, 1, 0, 1, 0, 0, 0]), 1)
    T["example_comparison"] = (O.Trace("begin push.9 read dup.2 lt.128 if.true mul else add end dup isodd.128 end", [], [6]), 2)
    T["example_collatz"] = (O.Trace("begin pad read dup push.1 ne while.true swap push.1 add swap dup isodd.128 if.true push.3 mul push.1 add else push.2 div end "
                                    "dup push.1 ne end swap end", [], [3]), 1)
    path = [[rnd.randrange(P) for _ in range(3)] for _ in range(2)]
    ta, tb, index = [path[0][0]], [path[1][0]], 2 + 4
    for i in range(1, 3):
        ta += [0, path[0][i]]; tb += [index & 1, path[1][i]]; index >>= 1
    for i in range(1, 3):
        ta.append(path[0][i]); tb.append(path[1][i])
    T["example_merkle"] = (O.Trace("begin read.ab dup.2 smpath.3 swap.2 push.2 roll.4 swap swap.2 pmpath.3 end", [], ta, tb), 4)
    T["example_range"] = (O.Trace("begin read rc.63 add read rc.63 add end", [0], [5, (1 << 63) + 17]), 1)
    return T


ISA_TRACE_NAMES = ["stack_manipulation", "choose", "choose2", "cswap2", "math_inv_neg_not", "bool_and_or", "read_read2", "assert_asserteq", "eq", "rescr_double_hash",
                   "cmp_128", "binacc_128", "if_true", "if_false", "while_skipped", "while_5_iterations", "nested_loops", "example_comparison", "example_collatz",
                   "example_merkle", "example_range"]


@pytest.mark.parametrize("instance", ["", "generic"])
@pytest.mark.parametrize("name", ISA_TRACE_NAMES)
def test_whole_instruction_set_and_flow_blocks(oracle, monkeypatch, name, instance):
    """a7 / a8 on valid traces of the whole instruction set and of if / while blocks (loop registers, LOOP / WRAP / BREAK / FEND rows): every
    intermediate of every phase against the oracle, through the default constraint instance for the shape and through the per-operation
    formulation (DISTAFF_AIR=generic).  The oracle side of these traces is pinned by the reference's literals in tests/test_oracle_isa.py."""
    import distaff_amd as D
    if instance:
        monkeypatch.setenv("DISTAFF_AIR", instance)
    trace, num_outputs = _isa_traces(oracle)[name]
    assert trace.trace_hash() == trace.program_hash
    _check_all_phases(oracle, D, trace, num_outputs=num_outputs, grinding=8)


def test_loops_and_macros_at_2_13_and_2_15(oracle):
    """The same at sizes where every kernel runs its multi-workgroup form: the Collatz example from 27 (111 iterations of a `while` loop with
    a nested if / else and isodd.128: 2^15 rows, 26 registers, loop depth 1, context depth 2 -- examples/collatz.rs) and 100 range checks
    (`read rc.63 add`: binacc x 63, eq, 2^13 rows -- examples/range.rs); every intermediate of every phase against the oracle."""
    import random
    import distaff_amd as D
    O = oracle
    t = O.Trace("begin pad read dup push.1 ne while.true swap push.1 add swap dup isodd.128 if.true push.3 mul push.1 add else push.2 div end "
                "dup push.1 ne end swap end", [], [27])
    assert (t.length, t.loop_depth, t.ctx_depth) == (1 << 15, 1, 2) and t.outputs(1) == [111]
    _check_all_phases(O, D, t, num_outputs=1, grinding=8)
    rnd = random.Random(3)
    values = [rnd.randrange(1 << 64) for _ in range(100)]
    t = O.Trace("begin " + "read rc.63 add " * 100 + "end", [0], values)
    assert t.length == 1 << 13 and t.outputs(1) == [sum(v < (1 << 63) for v in values)]
    _check_all_phases(O, D, t, num_outputs=1, grinding=8)


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_prover_on_loop_and_macro_traces(oracle, world):
    """dst_prove_sharded (thread-ranks) on traces that are not the Fibonacci shape: loop registers, context depth 2, stacks deeper than 8 --
    the constraint evaluation runs per rank on its own cosets, the last trace state travels with rank 0's status record; every rank must
    return the oracle's proof."""
    import distaff_amd as D
    O = oracle
    traces = _isa_traces(O)
    for name in ("while_5_iterations", "nested_loops", "example_comparison", "example_merkle", "cmp_128"):
        t, num_outputs = traces[name]
        op = O.Prover.from_trace(t, num_outputs, grinding=8)
        expected = op.prove()
        ctxs = []
        for r in range(world):
            ctx = D.Context(t.length.bit_length() - 1, t.width, t.ctx_depth, t.loop_depth, rank=r, world=world, grinding=8)
            ctx.upload(t.columns)
            ctxs.append(ctx)
        assert D.prove_sharded_local(ctxs, t.public_inputs, op.outputs) == expected, name
        for ctx in ctxs:
            ctx.close()


def test_isa_traces_cover_every_operation(oracle):
    """the set above executes all 32 user operations and all 8 flow operations, with loop_depth 0, 1 and 2 and ctx_depth up to 3"""
    users, flows, loop_depths = set(), set(), set()
    for trace, _ in _isa_traces(oracle).values():
        cols = trace.columns[:, :, 0].astype(np.int64)
        flows |= set((cols[5] + 2 * cols[6] + 4 * cols[7]).tolist())
        hacc = (cols[5] + cols[6] + cols[7]) == 0
        users |= set(sum(cols[8 + i] << i for i in range(7))[hacc].tolist())
        loop_depths.add(trace.loop_depth)
    assert flows == set(range(8))
    assert users == set(oracle.OPS.values())
    assert loop_depths >= {0, 1, 2}


def test_program_shapes_with_stack_depth_5_to_8(oracle):
    """Stack depth 5..8 (with and without a context register): the depth <= 8 kernel instance evaluates the low-degree stack
    operations as nested sums over all eight slots."""
    import distaff_amd as D
    O = oracle
    _check_all_phases(O, D, O.Trace("begin push.3 push.4 push.5 mul add dup.2 add drop end", [1, 2]), num_outputs=2)                       # depth 5
    _check_all_phases(O, D, O.Trace("begin dup.4 add mul swap.2 add drop drop block push.9 mul end end", [1, 2, 3, 4]), num_outputs=2)     # depth 8
    _check_all_phases(O, D, O.Trace("begin dup.4 add mul swap.2 add block dup.2 add drop push.9 mul end drop end", [1, 2, 3, 4]), num_outputs=3)
    _check_all_phases(O, D, O.Trace("begin dup.2 dup.2 dup.2 add mul block swap.2 dup.2 drop add end mul end", [1, 2]), num_outputs=4)


def test_blowup_16_and_64(oracle):
    import distaff_amd as D
    _check_all_phases(oracle, D, oracle.fibonacci_trace(256), log_blowup=4, num_queries=100)       # config 5's options
    _check_all_phases(oracle, D, oracle.fibonacci_trace(128), log_blowup=6, num_queries=30)


@pytest.mark.parametrize("log_n", [12, 16])
def test_config2_random_columns_lde_and_merkle(oracle, log_n):
    """BASELINE config 2 (2^16-step trace, LDE + Merkle commit only, bit-exact against the CPU): 20 uniform columns from splitmix64 as
    SURVEY.md section 8(d) specifies, at the stated size and at 2^12 (the size the CPU run of this test on the emulated build uses)."""
    import distaff_amd as D
    O = oracle
    n, W = 1 << log_n, 20
    P = O.P
    def splitmix(seed):
        x = seed
        while True:
            x = (x + 0x9E3779B97F4A7C15) & (2**64 - 1)
            z = x
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
            yield z ^ (z >> 31)
    cols = []
    for c in range(W):
        g = splitmix(0x44697374616666 + c)
        col = []
        while len(col) < n:
            v = next(g) | (next(g) << 64)
            if v < P:
                col.append(v)
        cols.append(col)
    columns = O.to_arr(cols)
    op = O.Prover(columns, 1, 0, [], [], ext=32)
    op.step(1); op.step(2)
    ctx = D.Context(log_n, W, 1, 0)
    ctx.upload(columns)
    root = ctx.commit_trace()
    assert root == op.get_bytes("roots")[:32]
    assert ctx.read("trace_leaves").tobytes() == op.get_bytes("trace_leaves")
    regs = op.get("registers")
    for c in (0, 7, 19):
        assert (ctx.read_elements("lde", c) == regs[c]).all()
    ctx.close()


def test_tiny_traces_of_32_and_16_rows(oracle):
    """The reference starts traces at MIN_TRACE_LENGTH = 16 rows (src/lib.rs:82); the shortest programs the VM can run fill 32 rows
    (root span + closing block).  Those go through every phase; a 16-row table (no program produces one) through extension + commitment."""
    import distaff_amd as D
    O = oracle
    for src, inputs in [("begin noop end", [1]), ("begin add end", [1, 2]), ("begin push.3 mul end", [2])]:
        t = O.Trace(src, inputs)
        assert t.columns.shape[1] == 32
        _check_all_phases(O, D, t, num_outputs=1)
    rng = np.random.default_rng(16)
    W, n = 17, 16
    cols = rng.integers(0, 2**62, size=(W, n, 2), dtype=np.uint64)
    for log_blowup in (5, 4):
        op = O.Prover(cols, 0, 0, [], [], ext=1 << log_blowup)
        op.step(1); op.step(2)
        ctx = D.Context(4, W, 0, 0, log_blowup=log_blowup)
        ctx.upload(cols)
        assert ctx.commit_trace() == op.get_bytes("roots")[:32]
        assert ctx.read("trace_leaves").tobytes() == op.get_bytes("trace_leaves")
        regs = op.get("registers")
        for c in (0, 16):
            assert (ctx.read_elements("lde", c) == regs[c]).all()
        ctx.close()


def test_repeated_query_positions_are_refused(oracle):
    """MerkleTree::prove_batch asserts 'repeating indexes detected' (merkle.rs:69); the C-ABI returns an argument error instead"""
    import distaff_amd as D
    t = oracle.fibonacci_trace(128)
    ctx = _ctx(D, t, grinding=8)
    ctx.upload(t.columns)
    ctx.prove(t.public_inputs, [int(oracle.to_ints(t.columns[16, -1:])[0])])
    with pytest.raises(D.DistaffError) as e:
        ctx.build_proof([5, 9, 5], 0)
    assert e.value.code == D.DST_ERR_ARG and "repeating indexes" in str(e.value)
    ctx.close()


def test_invalid_trace_reports_air_error(oracle):
    import distaff_amd as D
    O = oracle
    t = O.fibonacci_trace(128)
    cols = t.columns.copy()
    cols[16, 40, 0] += 1
    ctx = _ctx(D, t)
    ctx.upload(cols)
    root = ctx.commit_trace()
    with pytest.raises(D.DistaffError) as e:
        ctx.eval_constraints([1, 0], [1], D.prng_vector(root, 344))
    assert e.value.code == -3 and ctx.bad_step in (39, 40)
    ctx.close()


def test_wide_rows_two_chunk_leaves(oracle):
    """W > 64 registers: a trace row spans two BLAKE3 chunks (W*16 > 1024 bytes)."""
    import distaff_amd as D
    O = oracle
    rng = np.random.default_rng(1)
    W, n = 71, 64                     # 15 decoder + 16 context + 8 loop + 32 stack registers: 1136-byte rows
    cols = rng.integers(0, 2**62, size=(W, n, 2), dtype=np.uint64)
    op = O.Prover(cols, 16, 8, [], [], ext=16)
    op.step(1); op.step(2)
    ctx = D.Context(6, W, 16, 8, log_blowup=4)
    ctx.upload(cols)
    assert ctx.commit_trace() == op.get_bytes("roots")[:32]
    assert ctx.read("trace_leaves").tobytes() == op.get_bytes("trace_leaves")
    ctx.close()


def test_device_field_arithmetic(oracle):
    """fe.h on the device (incl. the gfx950-specific multiplication) against Python big integers, edge cases included."""
    import random
    import distaff_amd as D
    P = oracle.P
    rnd = random.Random(11)
    edge = [0, 1, 2, P - 1, P - 2, (P + 1) // 2, 2**64, 2**64 - 1, 2**127, 2**96, P - 2**40, 45 * 2**40 - 1, 45 * 2**40, 2**128 - 2**88, 2**32 - 1, 2**96 - 1]
    edge = [e % P for e in edge]
    a = [x for x in edge for _ in edge] + [rnd.randrange(P) for _ in range(60000)] + [P - 1 - rnd.randrange(2**20) for _ in range(4000)]
    b = [y for _ in edge for y in edge] + [rnd.randrange(P) for _ in range(60000)] + [P - 1 - rnd.randrange(2**70) for _ in range(4000)]
    A, B = D.ints_to_arr(a), D.ints_to_arr(b)
    ctx = D.Context(6, 17, 0, 0, log_blowup=4)
    assert D.arr_to_ints(ctx.field_op("mul", A, B)) == [x * y % P for x, y in zip(a, b)]
    assert D.arr_to_ints(ctx.field_op("mul_portable", A, B)) == [x * y % P for x, y in zip(a, b)]
    assert D.arr_to_ints(ctx.field_op("add", A, B)) == [(x + y) % P for x, y in zip(a, b)]
    assert D.arr_to_ints(ctx.field_op("sub", A, B)) == [(x - y) % P for x, y in zip(a, b)]
    assert D.arr_to_ints(ctx.field_op("inv", A[:300], B[:300])) == [pow(x, P - 2, P) for x in a[:300]]
    assert D.arr_to_ints(ctx.field_op("pow", A[:300], B[:300])) == [pow(x, y, P) if x else 0 for x, y in zip(a[:300], b[:300])]
    # sums of 40 products + one element with a single reduction (fe_acc), incl. the all-maximal case
    n = 20000
    want = [(sum(a[(i + j) % n] * b[(i + 7 * j) % n] for j in range(40)) + a[i]) % P for i in range(n)]
    assert D.arr_to_ints(ctx.field_op("dot40", A[:n], B[:n])) == want
    top = D.ints_to_arr([P - 1] * 256)
    assert D.arr_to_ints(ctx.field_op("dot40", top, top)) == [(40 * (P - 1) * (P - 1) + P - 1) % P] * 256
    ctx.close()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_prover_equals_single_gpu(oracle, monkeypatch, world):
    """Coset-sharded proving (distaff_amd/sharded.py) with `world` ranks as threads on one GPU: every rank returns the oracle's proof.
    At these sizes the whole FRI commit phase is in the replicated tail (layers of <= 2^17 elements); lowering the limit makes the
    first one to three layers sharded (coset-major, boundary-node exchanges), which is what a 2^20-step proof does with four."""
    import distaff_amd as D
    from distaff_amd import sharded
    O = oracle
    for log_n, log_b, nq, replicate_log in ((8, 5, 50, None), (10, 5, 50, None), (8, 4, 100, None), (10, 5, 50, 9), (10, 5, 50, 13), (8, 6, 60, 10), (7, 7, 40, 0)):
        if world > min(8, (1 << log_b) // 2):              # two cosets per rank is the minimum (blowup 16 on 8 ranks: BASELINE config 5's shape)
            continue
        if replicate_log is None:
            monkeypatch.delenv("DISTAFF_FRI_REPLICATE_LOG", raising=False)
        else:
            monkeypatch.setenv("DISTAFF_FRI_REPLICATE_LOG", str(replicate_log))
        t = O.fibonacci_trace(1 << log_n)
        op = O.Prover.from_trace(t, 1, ext=1 << log_b, num_queries=nq, grinding=10)
        expected = op.prove()
        for python_openings in (False, True):                 # openings planned behind the C-ABI / by the Python statement of the plan
            proofs = sharded.prove_local(t.columns, log_n, t.width, t.ctx_depth, t.loop_depth, t.public_inputs, op.outputs, world,
                                         python_openings=python_openings, log_blowup=log_b, num_queries=nq, grinding=10)
            assert all(p == expected for p in proofs), (world, log_n, log_b, replicate_log, python_openings)


def test_sharded_prover_reports_invalid_trace(oracle):
    import distaff_amd as D
    from distaff_amd import sharded
    t = oracle.fibonacci_trace(256)
    cols = t.columns.copy()
    cols[17, 100, 0] += 1
    with pytest.raises(D.DistaffError):
        sharded.prove_local(cols, 8, t.width, t.ctx_depth, t.loop_depth, [1, 0], [1], 2, grinding=8)


def test_sharded_fri_protocol_state_errors(oracle, monkeypatch):
    """The begin / end / roots protocol of the FRI commit phase refuses calls out of order with DST_ERR_STATE (one rank, host buffers)."""
    import ctypes
    import distaff_amd as D
    from distaff_amd import sharded
    monkeypatch.setenv("DISTAFF_FRI_REPLICATE_LOG", "9")                       # 2^13, 2^11 sharded, then the replicated tail
    t = oracle.fibonacci_trace(256)
    ctx = D.Context(8, t.width, t.ctx_depth, t.loop_depth, grinding=8)
    ctx.upload(t.columns)
    prover = sharded.ShardedProver(ctx, sharded.LocalComm.create(1)[0])
    expected = prover.prove(t.public_inputs, [3])                               # a complete run first: all buffers exist
    roots, rep_from = ctx.shard_fri_roots()
    assert len(roots) == 4 and rep_from == 2
    cap = ctx.shard_export_size(sharded.SH_FRI_SEND_CAP, 0)
    buf = np.zeros(cap, dtype=np.uint8)

    def state_error(call):
        with pytest.raises(D.DistaffError) as e:
            call()
        assert e.value.code == D.DST_ERR_STATE, e.value

    # a new proof up to the composition; then FRI calls out of order
    ctx.shard_commit_trace()
    root = prover._exchange(sharded.SH_TRACE_TREE)
    assert ctx.shard_eval_constraints(t.public_inputs, [3], D.lib.prng_vector(root, 344)) == -1
    prover._exchange(sharded.SH_CEVAL)
    ctx.shard_combine()
    croot = prover._exchange(sharded.SH_CONSTRAINT_TREE)
    state_error(lambda: ctx.shard_fri_begin(buf.ctypes.data, False, cap))      # composition not built
    ctx.compose(D.lib.prng_vector(croot, 516))
    state_error(lambda: ctx.shard_fri_roots())                                 # commit phase not finished
    state_error(lambda: ctx.shard_fri_end(buf.ctypes.data, False))             # nothing in flight
    size, more = ctx.shard_fri_begin(buf.ctypes.data, False, cap)
    assert more and size == ctx.shard_export_size(sharded.SH_FRI_TREE, 0)
    state_error(lambda: ctx.shard_fri_begin(buf.ctypes.data, False, cap))      # layer 0 not folded yet
    ctx.shard_fri_end(buf.ctypes.data, False)                                  # one rank: what it sent is what the all-gather delivers
    size, more = ctx.shard_fri_begin(buf.ctypes.data, False, cap)
    ctx.shard_fri_end(buf.ctypes.data, False)
    state_error(lambda: ctx.shard_fri_layer())                                 # layer 2 belongs to the replicated tail
    size, more = ctx.shard_fri_begin(buf.ctypes.data, False, cap)              # the tail: this rank's cosets of layer 2
    assert not more and size == (1 << 9) * 16
    state_error(lambda: ctx.shard_fri_begin(buf.ctypes.data, False, cap))      # exchange pending
    ctx.shard_fri_end(buf.ctypes.data, False)
    assert ctx.shard_fri_roots() == (roots, rep_from)                          # same trace, same challenges: same commitments
    state_error(lambda: ctx.shard_fri_begin(buf.ctypes.data, False, cap))      # commit phase over
    assert prover.prove(t.public_inputs, [3]) == expected
    ctx.close()


def test_sharded_prover_over_torch_distributed_world1(oracle, monkeypatch):
    """The torch.distributed (RCCL) transport of the sharded prover on one GPU, both shard-transfer modes: staged through the host and
    directly between libdistaff_hip.so's buffers and torch tensors on the device; with the FRI limit lowered so that two layers go
    through the boundary-node exchange before the replicated tail."""
    monkeypatch.setenv("DISTAFF_FRI_REPLICATE_LOG", "9")
    import socket
    import torch
    import torch.distributed as dist
    import distaff_amd as D
    from distaff_amd import sharded
    O = oracle
    t = O.fibonacci_trace(256)
    op = O.Prover.from_trace(t, 1, grinding=8)
    expected = op.prove()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        for device_path in (False, True):
            comm = sharded.TorchComm(dist, torch.device("cuda", 0), device_path=device_path)
            ctx = D.Context(8, t.width, t.ctx_depth, t.loop_depth, grinding=8)
            ctx.upload(t.columns)
            proof = sharded.ShardedProver(ctx, comm).prove(t.public_inputs, op.outputs)
            assert proof == expected, device_path
            ctx.close()
    finally:
        dist.destroy_process_group()


def _random_columns(W, n, seed):
    rng = np.random.default_rng(seed)
    cols = rng.integers(0, 2**63, size=(W, n, 2), dtype=np.uint64)      # high limb < 2^63: always below the modulus
    return cols


@pytest.mark.parametrize("family,log_n,log_blowup", [("reg", 13, 5), ("reg", 14, 5), ("reg", 15, 4), ("reg", 16, 5), ("reg", 17, 4), ("reg", 18, 4),
                                                      ("reg", 19, 4), ("reg", 20, 4), ("reg", 21, 4), ("reg", 22, 4), ("reg", 23, 4), ("reg", 24, 4),
                                                      ("lds", 13, 5), ("lds", 16, 5), ("lds", 19, 4), ("lds", 22, 4), ("auto", 21, 4), ("auto", 22, 4), ("auto", 23, 4), ("auto", 24, 4), ("3pass", 20, 5), ("dif", 16, 5), ("dif", 21, 4), ("dit", 16, 5), ("dit", 20, 5), ("dit2", 13, 5), ("dit2", 20, 5),
                                                      ("waves4", 20, 5), ("waves8", 13, 5), ("waves8", 16, 5), ("order0", 16, 5), ("order0", 20, 5), ("generic", 20, 5),
                                                      ("pre", 13, 5), ("pre", 16, 5), ("pre", 20, 5), ("3pass", 21, 4), ("3pass", 22, 4)])
def test_lde_every_tile_length(oracle, monkeypatch, family, log_n, log_blowup):
    """Both NTT kernel families (register-radix: every tile length 2^6 .. 2^12 in both passes; LDS radix-2) and the default per-pass
    choice at the largest size, through size-independent properties that pin the
    transforms point-wise: the interpolated polynomial evaluated by the oracle (Horner) at trace-domain points gives the trace, and at
    LDE-domain points gives the device's extension -- including the largest trace length (BASELINE config 5: 2^24, blowup 16).
    "auto" at 2^21 / 2^22 is the two-pass plan with a register pre-stage (2048-point factors on 1024 x 4 tiles); "pre" forces pre-stages
    onto smaller transforms, "3pass" the three-pass plan onto sizes that would not take it."""
    import distaff_amd as D
    O = oracle
    n, B, W = 1 << log_n, 1 << log_blowup, 16
    cols = _random_columns(W, n, 1000 + log_n)
    monkeypatch.delenv("DISTAFF_NTT_DIF", raising=False)
    monkeypatch.delenv("DISTAFF_NTT_WAVES", raising=False)
    monkeypatch.delenv("DISTAFF_NTT_ORDER", raising=False)
    monkeypatch.delenv("DISTAFF_NTT_FIXED", raising=False)
    if family in ("waves4", "waves8"):                                   # LDS family forced to 512 lanes + register prefetch / to 1024 lanes at 8 waves per SIMD
        monkeypatch.delenv("DISTAFF_NTT", raising=False)                # (default: the latter for 1024-point tiles only)
        monkeypatch.setenv("DISTAFF_NTT_WAVES", family[-1])
    elif family == "generic":                                            # the any-shape instances instead of the ones compiled for 1024 x 4 tiles
        monkeypatch.delenv("DISTAFF_NTT", raising=False)
        monkeypatch.setenv("DISTAFF_NTT_FIXED", "0")
    elif family == "order0":                                             # coset-slow block order of the first pass (default: every coset of a tile group first)
        monkeypatch.delenv("DISTAFF_NTT", raising=False)
        monkeypatch.setenv("DISTAFF_NTT_ORDER", "0")
    elif family in ("dif", "dit", "dit2"):                                 # first pass forced to pre-scale + DIF / to the coset DIT with its table in LDS
        monkeypatch.delenv("DISTAFF_NTT", raising=False)                # (1024-lane instance at 2^20) / to the DIT whose last-stage twiddles stay in global memory
        monkeypatch.setenv("DISTAFF_NTT_DIF", {"dif": "1", "dit": "0", "dit2": "2"}[family])
    elif family == "auto":
        monkeypatch.delenv("DISTAFF_NTT", raising=False)
    else:
        monkeypatch.setenv("DISTAFF_NTT", family)                       # read by dst_ctx_create
    ctx = D.Context(log_n, W, 0, 0, log_blowup=log_blowup)
    ctx.upload(cols)
    ctx.commit_trace()
    c = 5
    poly = ctx.read_elements("polys").reshape(W, n, 2)[c]
    lde = ctx.read_elements("lde", c)
    ctx.close()
    assert lde.shape == (n * B, 2)
    g_n, g_N = O.root_of_unity(n), O.root_of_unity(n * B)
    rng = np.random.default_rng(7 + log_n)
    samples = 6 if log_n <= 20 else 3
    for k in [0, 1, n - 1] + [int(v) for v in rng.integers(0, n, size=samples)]:
        assert O.poly_eval(poly, O.exp(g_n, k)) == O.to_ints(cols[c, k:k + 1])[0], ("interpolation", k)
    for i in [1, B - 1, B, n * B - 1] + [int(v) for v in rng.integers(0, n * B, size=samples)]:
        assert O.poly_eval(poly, O.exp(g_N, i)) == O.to_ints(lde[i:i + 1])[0], ("extension", i)
    assert (lde[::B] == cols[c]).all()                                 # coset 0 of the extension is the trace itself


def test_config3_full_size_proof_is_accepted_and_tamper_evident(oracle):
    """BASELINE config 3 at full size (2^20-step Fibonacci trace, default ProofOptions).  The oracle cannot produce this proof in
    seconds, so the checks are size-independent: the oracle's restatement of the reference verifier accepts the GPU proof for the
    right public data, and rejects it for a wrong output, a wrong program hash and a flipped byte, with the reference's error strings."""
    import distaff_amd as D
    O = oracle
    cols, program_hash, result = _fib(20)
    ctx = D.Context(20, 20, 1, 0)
    ctx.upload(cols)
    proof = ctx.prove([1, 0], [result])
    again = ctx.prove([1, 0], [result])
    ctx.close()
    assert proof == again                                            # deterministic
    ok, err = O.verify(proof, program_hash, [1, 0], [result])
    assert ok, err
    ok, err = O.verify(proof, program_hash, [1, 0], [result + 1])
    assert not ok and "verification of low-degree proof failed" in err
    ok, err = O.verify(proof, bytes(32), [1, 0], [result])
    assert not ok
    bad = bytearray(proof); bad[40] ^= 1
    ok, err = O.verify(bytes(bad), program_hash, [1, 0], [result])
    assert not ok


_FIB_CACHE = {}


def _fib(log_n):
    """the library's host-side trace generator, once per size and test session (14 s at 2^20, one thread: the sponge is a chain)"""
    import distaff_amd as D
    if log_n not in _FIB_CACHE:
        if log_n >= 22:
            _FIB_CACHE.clear()                                        # 1.3 GiB at 2^22: keep one
        import os
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        _FIB_CACHE[log_n] = bench.fibonacci_trace_cached(D, log_n)     # BENCH_TRACE_CACHE=<dir>: shared with bench.py runs on the same box
    return _FIB_CACHE[log_n]


def _accepts_and_rejects(O, proof, program_hash, result):
    ok, err = O.verify(proof, program_hash, [1, 0], [result])
    assert ok, err
    ok, err = O.verify(proof, program_hash, [1, 0], [result + 1])
    assert not ok and "verification of low-degree proof failed" in err
    bad = bytearray(proof); bad[len(proof) // 2] ^= 1
    ok, err = O.verify(bytes(bad), program_hash, [1, 0], [result])
    assert not ok


def _sampled_parity(O, D, log_n, log_blowup=5, num_queries=50, points=32, horner_rows=32, deep_registers=20, seed=1, trace=None, num_outputs=1):
    """Oracle POINT computations against intermediates of a full-size proof (sizes at which the oracle's whole-domain loops take hours):
      * trace polynomials: Horner at trace-domain points = the trace (interpolation), at LDE points = sampled LDE rows (trace_table.rs:143-169);
      * the reference evaluator on sampled row pairs of the 8n-point domain = the transition combination the constraint kernels
        wrote (evaluator.rs:139-162), incl. trace steps (must vanish) and the excepted last step;
      * the constraint polynomial at those points, rebuilt from the oracle's boundary / transition values and the divisors of
        combine_polys (constraint_table.rs:54-88), = the constraint LDE;
      * DEEP values = Horner at z and z*g; the composition (trace_table.rs:206-261, constraint_poly.rs:39-52) at sampled LDE points,
        rebuilt with Python integers from those rows, = the composition LDE;
      * every FRI layer: sampled rows hashed by the oracle's BLAKE3 = the leaf, folded by the oracle's quartic interpolation /
        evaluation at x = prng(root) (fri/prover.rs:25-49, quartic.rs:20-60) = the entry of the next layer.
    Challenges are the library's own Fiat-Shamir draws (pinned against the oracle's PRNG in test_host_logic).
    `trace`: any oracle.Trace instead of the Fibonacci trace of 2^log_n steps (other register counts, context / loop / stack depths)."""
    P = O.P
    if trace is None:
        cols, program_hash, result = _fib(log_n)
        W, ctx_depth, loop_depth, stack_depth, inputs, outputs = 20, 1, 0, 4, [1, 0], [result]
    else:
        cols, log_n = trace.columns, trace.length.bit_length() - 1
        W, ctx_depth, loop_depth, stack_depth = trace.width, trace.ctx_depth, trace.loop_depth, trace.stack_depth
        inputs, outputs = trace.public_inputs, trace.outputs(num_outputs)
    n, B = 1 << log_n, 1 << log_blowup
    N = n * B
    rng = np.random.default_rng(seed)
    ctx = D.Context(log_n, W, ctx_depth, loop_depth, log_blowup=log_blowup, num_queries=num_queries)
    ctx.upload(cols)
    ints = lambda a: [int(lo) | (int(hi) << 64) for lo, hi in np.asarray(a, dtype=np.uint64).reshape(-1, 2)]      # noqa: E731
    elems = lambda b: ints(np.frombuffer(b, dtype=np.uint64))                                                     # noqa: E731
    cm = lambda pos: (pos % B) * n + pos // B            # natural LDE position -> coset-major index                 # noqa: E731
    g_n, g_N, g_8n = O.root_of_unity(n), O.root_of_unity(N), O.root_of_unity(8 * n)

    # ---- steps 1-2
    root = ctx.commit_trace()
    polys = ctx.read_elements("polys").reshape(W, n, 2)
    for c, k in ((0, 0), (4, 1), (W - 1, n - 1), (int(rng.integers(W)), int(rng.integers(n)))):
        assert O.poly_eval(polys[c], O.exp(g_n, k)) == ints(cols[c, k])[0], ("interpolation", c, k)
    steps = [0, 8, 3, 8 * n - 8, 8 * n - 1, 8 * n - 5, 8 * 17 + 5] + [int(v) for v in rng.integers(0, 8 * n, size=max(points - 7, 0))]
    steps = steps[:max(points, 7)]
    pos = [s_ * (B // 8) for s_ in steps]
    rows_cur = np.frombuffer(ctx.shard_read(10, 0, pos), dtype=np.uint64).reshape(len(pos), W, 2)
    rows_nxt = np.frombuffer(ctx.shard_read(10, 0, [(p_ + B) % N for p_ in pos]), dtype=np.uint64).reshape(len(pos), W, 2)
    checked = 0
    for i in range(len(pos)):                                          # Horner over n coefficients per (row, register)
        for rows, p_ in ((rows_cur, pos[i]), (rows_nxt, (pos[i] + B) % N)):
            if checked >= horner_rows:
                break
            x = O.exp(g_N, p_)
            for c in range(W):
                assert O.poly_eval(polys[c], x) == ints(rows[i, c])[0], ("LDE row", p_, c)
            checked += 1

    # ---- steps 3-5
    coeffs = D.prng_vector(root, 344)
    croot = ctx.eval_constraints(inputs, outputs, coeffs)
    tv = elems(ctx.shard_read(11, 0, [(s_ % 8) * n + s_ // 8 for s_ in steps]))
    cv = elems(ctx.shard_read(3, 0, [cm(p_) for p_ in pos]))
    op_count, ph = ints(cols[0, n - 1])[0], [ints(cols[1, n - 1])[0], ints(cols[2, n - 1])[0]]
    x_last = pow(g_n, n - 1, P)
    for i, s_ in enumerate(steps):
        x = O.exp(g_8n, s_)
        t, bi, bf, ok = O.evaluate_at(n, ctx_depth, loop_depth, stack_depth, coeffs, ph, op_count, inputs, outputs, s_, x, ints(rows_cur[i]), ints(rows_nxt[i]))
        assert ok and t == tv[i], ("transition combination", s_)
        if s_ % 8:                                                     # off the trace domain the divisors are invertible
            c_x = (bi * pow(x - 1, -1, P) + bf * pow(x - x_last, -1, P) + t * (x - x_last) % P * pow(pow(x, n, P) - 1, -1, P)) % P
            assert c_x == cv[i], ("constraint polynomial", s_)
    cpoly = ctx.read_elements("cpoly")
    for i in (1, 2):
        assert O.poly_eval(cpoly, O.exp(g_N, pos[i])) == cv[i], ("constraint LDE", pos[i])

    # ---- step 6
    draws = D.prng_vector(croot, 516)
    z1, z2 = ctx.compose(draws)
    dr = ints(draws)
    z, k1, k2, k3 = dr[0], dr[513], dr[514], dr[515]
    zg = z * g_n % P
    z1i, z2i = ints(z1), ints(z2)
    for c in list(range(W))[:deep_registers]:
        assert O.poly_eval(polys[c], z) == z1i[c] and O.poly_eval(polys[c], zg) == z2i[c], ("DEEP value", c)
    c_z = O.poly_eval(cpoly, z)
    comp = elems(ctx.shard_read(6, 0, [cm(p_) for p_ in pos]))
    inc = 6 * n + 1                                                    # utils/mod.rs:20 get_incremental_trace_degree
    for i, p_ in enumerate(pos):
        x = pow(g_N, p_, P)
        row = ints(rows_cur[i])
        a = sum(dr[1 + c] * (row[c] - z1i[c]) for c in range(W)) % P
        b = sum(dr[257 + c] * (row[c] - z2i[c]) for c in range(W)) % P
        t1 = (a * pow(x - z, -1, P) + b * pow(x - zg, -1, P)) % P
        want = (t1 * (k1 + k2 * pow(x, inc, P)) + k3 * (cv[i] - c_z) % P * pow(x - z, -1, P)) % P
        assert want == comp[i], ("composition", p_)
    del polys, cpoly

    # ---- step 7
    d, size = 0, N
    while True:
        lroot, more = ctx.fri_commit_layer()
        R = size // 4
        rs = [0, 1, R - 1] + [int(v) for v in rng.integers(0, R, size=5)]
        idx = [r_ + q * R for r_ in rs for q in range(4)]
        vals = elems(ctx.shard_read(6, d, [cm(v) for v in idx] if d == 0 else idx))
        leaves = ctx.shard_read(7, d, rs)
        for j, r_ in enumerate(rs):
            row4 = vals[4 * j:4 * j + 4]
            assert O.blake3(b"".join(v.to_bytes(16, "little") for v in row4)) == leaves[32 * j:32 * j + 32], ("FRI leaf", d, r_)
        if not more:
            break
        x = ints(D.prng_vector(lroot, 1))[0]
        ctx.fri_fold(x)
        nxt = elems(ctx.shard_read(6, d + 1, rs))
        for j, r_ in enumerate(rs):
            xs = [pow(g_N, (4 ** d) * (r_ + q * R), P) for q in range(4)]
            poly = O.quartic_interpolate_batch(O.to_arr([xs]), O.to_arr([vals[4 * j:4 * j + 4]]))
            assert O.to_ints(O.quartic_evaluate_batch(poly, x))[0] == nxt[j], ("FRI fold", d, r_)
        d, size = d + 1, R
    ctx.close()


@pytest.mark.parametrize("log_n,log_blowup", [(7, 5), (10, 4)])
def test_sampled_oracle_parity_small(oracle, log_n, log_blowup):
    """the sampled checks of the full-size tests at sizes where every intermediate is ALSO compared in full (test_fibonacci_all_phases):
    pins the sampling arithmetic itself, and runs on the CPU-emulated build"""
    import distaff_amd as D
    _sampled_parity(oracle, D, log_n, log_blowup=log_blowup, points=24, horner_rows=48)


def test_sampled_oracle_parity_on_a_long_loop_trace(oracle):
    """The same sampled checks on a trace that is NOT the Fibonacci shape, at a size where the oracle's whole-domain loops take minutes: the
    Collatz example from 837 799 (524 iterations of a `while` loop with a nested if / else, isodd.128, eq: 2^17 rows, 26 registers, loop
    depth 1, context depth 2 -- the any-shape constraint instances, 7-block leaves, 26-register transform batches), and the complete
    proof accepted by the oracle's verifier."""
    import distaff_amd as D
    O = oracle
    t = O.Trace("begin pad read dup push.1 ne while.true swap push.1 add swap dup isodd.128 if.true push.3 mul push.1 add else push.2 div end "
                "dup push.1 ne end swap end", [], [837799])
    assert (t.length, t.width, t.loop_depth, t.ctx_depth) == (1 << 17, 26, 1, 2) and t.outputs(1) == [524]
    _sampled_parity(O, D, None, points=24, horner_rows=8, deep_registers=26, trace=t)
    ctx = _ctx(D, t)
    ctx.upload(t.columns)
    proof = ctx.prove(t.public_inputs, t.outputs(1))
    assert O.verify(proof, t.program_hash, t.public_inputs, t.outputs(1)) == (True, "")
    assert O.verify(proof, t.program_hash, t.public_inputs, [523])[0] is False
    ctx.close()


def test_config3_sampled_oracle_parity_at_full_size(oracle):
    """BASELINE config 3 (2^20 steps, default options): 32 sampled points of every phase against oracle point computations."""
    import distaff_amd as D
    _sampled_parity(oracle, D, 20, points=32, horner_rows=32)


def test_config4_trace_full_size_on_one_gpu(oracle):
    """BASELINE config 4's trace (2^22 steps, default ProofOptions) on ONE GPU: three-pass transforms, 40 GiB of extension; the
    oracle's restatement of the reference verifier accepts the proof and rejects a wrong output / a flipped byte."""
    import distaff_amd as D
    cols, program_hash, result = _fib(22)
    ctx = D.Context(22, 20, 1, 0)
    ctx.upload(cols)
    proof = ctx.prove([1, 0], [result], cap=1 << 22)
    ctx.close()
    _accepts_and_rejects(oracle, proof, program_hash, result)


def test_config4_sampled_oracle_parity_at_full_size(oracle):
    """BASELINE config 4's trace (2^22 steps, three-pass transforms): sampled points of every phase against oracle point computations
    (32 evaluator points; 8 LDE rows and 8 DEEP registers by Horner -- 4 M coefficients each)."""
    import distaff_amd as D
    _sampled_parity(oracle, D, 22, points=32, horner_rows=8, deep_registers=8, seed=2)


def test_config5_full_size_on_one_gpu(oracle):
    """BASELINE config 5 (2^24 steps, blowup 16, 100 queries = 120-bit security) on ONE GPU (~190 GiB of the 288): the proof is accepted
    by the oracle's restatement of the reference verifier and rejected after tampering.  Most of the test's time is the host-side VM
    that generates the 2^24-step trace."""
    import distaff_amd as D
    cols, program_hash, result = _fib(24)                               # kept for the sampled test below (the VM takes minutes at 2^24)
    ctx = D.Context(24, 20, 1, 0, log_blowup=4, num_queries=100, grinding=20)
    ctx.upload(cols)
    del cols
    proof = ctx.prove([1, 0], [result], cap=1 << 24)
    ctx.close()
    _accepts_and_rejects(oracle, proof, program_hash, result)


def test_config5_sampled_oracle_parity_at_full_size(oracle):
    """BASELINE config 5 (2^24 steps, blowup 16, 100 queries): sampled points of every phase against oracle point computations
    (32 evaluator points; 2 LDE rows and 4 DEEP registers by Horner -- 16 M coefficients each)."""
    import distaff_amd as D
    _sampled_parity(oracle, D, 24, log_blowup=4, num_queries=100, points=32, horner_rows=2, deep_registers=4, seed=3)


def _sharded_local_equals_single_context(log_n, world, **options):
    """dst_prove_sharded_local (collectives behind the C-ABI, one thread per rank, all ranks sharing the one GPU of the test box) against
    the single-context proof of the same trace -- which the oracle's verifier judges in the config 4 / config 5 tests"""
    import distaff_amd as D
    cols, program_hash, result = _fib(log_n)
    ctx = D.Context(log_n, 20, 1, 0, **options)
    ctx.upload(cols)
    expected = ctx.prove([1, 0], [result], cap=1 << 24)
    ctx.close()
    ctxs = []
    try:
        for r in range(world):
            c = D.Context(log_n, 20, 1, 0, rank=r, world=world, **options)
            c.upload_owned(cols)                                     # 1 / world of the trace per rank: the interpolation is split by columns
            ctxs.append(c)
        assert D.prove_sharded_local(ctxs, [1, 0], [result], cap=1 << 24) == expected
    finally:
        for c in ctxs:
            c.close()


def test_config4_sharded_over_8_ranks_equals_single_context():
    """BASELINE config 4 (2^22-step trace, default ProofOptions, 8 ranks): the sharded prover behind the C-ABI with 8 thread-ranks --
    four LDE cosets per rank, k-range tree exchange of 2^22 boundary nodes per tree, three-pass transforms -- returns the single-context
    proof.  (On the driver's 8-GPU node the ranks are processes over RCCL; here they share the box's one GPU: ~20 GiB per rank.)"""
    _sharded_local_equals_single_context(22, 8)


def test_config5_shape_sharded_over_8_ranks_equals_single_context():
    """BASELINE config 5's shape (blowup 16, 100 queries, 8 ranks = two LDE cosets per rank: the constraint tree has no rank-local
    level) at 2^22 steps.  At the stated 2^24 steps one rank's buffers are ~60 GiB: eight of them do not fit the ONE GPU of this box
    (the single-context proof at 2^24 is test_config5_full_size_on_one_gpu); the partitioning does not depend on the size."""
    _sharded_local_equals_single_context(22, 8, log_blowup=4, num_queries=100)


def test_sharded_world8_at_bench_size_equals_single_context():
    """Config 3's trace (2^20 steps, default options) through the sharded path with 8 thread-ranks sharing the GPU: every rank must
    return the single-context proof (which the oracle's verifier accepts in test_config3_full_size_proof_is_accepted_and_tamper_evident)."""
    import distaff_amd as D
    from distaff_amd import sharded
    cols, program_hash, result = _fib(20)
    ctx = D.Context(20, 20, 1, 0)
    ctx.upload(cols)
    expected = ctx.prove([1, 0], [result])
    ctx.close()
    proofs = sharded.prove_local(cols, 20, 20, 1, 0, [1, 0], [result], 8)
    assert all(p == expected for p in proofs)


def test_small_fri_layers_in_one_launch_equal_the_per_layer_path(oracle, monkeypatch):
    """fri::reduce from the first layer of at most 2^13 evaluations on runs as ONE launch (k_fri_tail: rows hashed, trees built,
    x = field::prng(root) drawn by the device's own ChaCha20 / Uniform, folds) -- the proof must equal the oracle's and the one made with
    a launch and a root read-back per layer (DISTAFF_FRI_TAIL=0); blowup 16 makes the remainder 128 elements instead of 256."""
    import distaff_amd as D
    O = oracle
    for log_n, log_b in ((8, 5), (10, 5), (9, 4), (12, 5)):
        t = O.fibonacci_trace(1 << log_n)
        op = O.Prover.from_trace(t, 1, ext=1 << log_b, grinding=8)
        expected = op.prove()
        def stats_of_a_proof(**switches):
            # the library reads its switches once, when a context is created: one context per setting
            for k in ("DISTAFF_FRI_TAIL", "DISTAFF_FRI_CHAIN"):
                monkeypatch.delenv(k, raising=False)
            for k, v in switches.items():
                monkeypatch.setenv(k, v)
            ctx = D.Context(log_n, t.width, t.ctx_depth, t.loop_depth, log_blowup=log_b, grinding=8)
            ctx.upload(t.columns)
            ctx.set_profiling(1); ctx.kernel_stats(reset=True)
            assert ctx.prove(t.public_inputs, op.outputs) == expected
            stats = ctx.kernel_stats(reset=True)
            ctx.close()
            return stats
        assert stats_of_a_proof().get("fri_tail_kernel", {}).get("launches") == 1
        stats = stats_of_a_proof(DISTAFF_FRI_TAIL="0")
        assert "fri_tail_kernel" not in stats and stats.get("fri_draw_kernel", {}).get("launches", 0) >= 2      # every layer's x drawn on the device
        # dst_prove commits the layers above the tail without host round trips (x = prng(root) drawn by fri_draw_kernel, the fold reads it
        # from device memory); DISTAFF_FRI_CHAIN=0 keeps one root read-back and one host draw per layer: the same proof either way
        assert "fri_draw_kernel" not in stats_of_a_proof(DISTAFF_FRI_TAIL="0", DISTAFF_FRI_CHAIN="0")
        assert "fri_draw_kernel" not in stats_of_a_proof(DISTAFF_FRI_CHAIN="0")
    for k in ("DISTAFF_FRI_TAIL", "DISTAFF_FRI_CHAIN"):
        monkeypatch.delenv(k, raising=False)


def test_fibonacci_2_16_proof_bytes_equal_oracle(oracle):
    """Largest size at which the oracle's own prover finishes in about ten seconds: byte-identical proofs, default ProofOptions."""
    import distaff_amd as D
    O = oracle
    t = O.fibonacci_trace(1 << 16)
    expected = O.Prover.from_trace(t, 1).prove()
    cols, program_hash, result = D.fibonacci_trace(16)
    assert (cols == t.columns).all()                                 # the library's own trace generator (host_vm.h) against the oracle VM
    ctx = D.Context(16, 20, 1, 0)
    ctx.upload(cols)
    assert ctx.prove([1, 0], [result]) == expected
    ctx.close()


def test_sharded_world8_at_2_16_equals_single_context():
    """Buffer sizing of the sharded path at a non-toy size: 8 thread-ranks on one GPU, 2^16-step Fibonacci trace, default options;
    the single-context proof of the same trace is the reference (it equals the oracle's in test_fibonacci_2_16_proof_bytes_equal_oracle)."""
    import distaff_amd as D
    from distaff_amd import sharded
    cols, program_hash, result = D.fibonacci_trace(16)
    ctx = D.Context(16, 20, 1, 0)
    ctx.upload(cols)
    expected = ctx.prove([1, 0], [result])
    ctx.close()
    proofs = sharded.prove_local(cols, 16, 20, 1, 0, [1, 0], [result], 8)
    assert all(p == expected for p in proofs)


@pytest.mark.parametrize("world,log_n,replicate_log,gather", [(2, 8, None, False), (4, 8, 9, False), (8, 8, 9, False), (8, 10, 9, False), (4, 10, 9, True), (8, 7, None, False)])
def test_prove_sharded_behind_the_c_abi(oracle, monkeypatch, world, log_n, replicate_log, gather):
    """dst_prove_sharded_local: the whole sharded proof inside the library (collectives included, in-process transport, one thread per
    rank): k-range Merkle exchange (all-to-all of boundary nodes, subtree per rank, all-gather of the subtree roots), all-gathers of the
    constraint evaluations and of the first small FRI layer, openings from three kinds of owners.  Every variant must return the
    oracle's proof bytes; `gather` forces the all-gather form of the tree exchange instead."""
    import distaff_amd as D
    O = oracle
    if replicate_log is None:
        monkeypatch.delenv("DISTAFF_FRI_REPLICATE_LOG", raising=False)
    else:
        monkeypatch.setenv("DISTAFF_FRI_REPLICATE_LOG", str(replicate_log))
    if gather:
        monkeypatch.setenv("DISTAFF_SHARD_TREE_GATHER", "1")
    else:
        monkeypatch.delenv("DISTAFF_SHARD_TREE_GATHER", raising=False)
    t = O.fibonacci_trace(1 << log_n)
    op = O.Prover.from_trace(t, 1, grinding=8)
    expected = op.prove()
    ctxs = []
    for r in range(world):
        ctx = D.Context(log_n, t.width, t.ctx_depth, t.loop_depth, rank=r, world=world, grinding=8)
        # the interpolation is split by columns: a rank needs only the registers r (mod world) of the trace on its device
        # (dst_trace_upload_owned); a whole trace works as well (every other configuration)
        if (world + log_n) % 2 == 0:
            ctx.upload_owned(t.columns)
        else:
            ctx.upload(t.columns)
        ctxs.append(ctx)
    for _ in range(2):                                                # twice: the buffers of the first proof are reused
        assert D.prove_sharded_local(ctxs, t.public_inputs, op.outputs) == expected
    for ctx in ctxs:
        ctx.close()


@pytest.mark.parametrize("world", [2, 8])
def test_prove_sharded_with_collectives_on_their_own_stream(oracle, monkeypatch, world):
    """The stream / event choreography dst_prove_sharded uses on a stream-ordered transport (RCCL with several ranks: the coefficient
    all-gathers of round k + 1 on the collective stream while round k is extended, the exchange of the constraint evaluations while the
    boundary combinations are written) -- forced here over the in-process transport, which a one-GPU box can run with any number of ranks."""
    import distaff_amd as D
    O = oracle
    monkeypatch.setenv("DISTAFF_SHARD_FORCE_OVERLAP", "1")
    monkeypatch.delenv("DISTAFF_FRI_REPLICATE_LOG", raising=False)
    t = O.fibonacci_trace(1 << 9)
    op = O.Prover.from_trace(t, 1, grinding=8)
    expected = op.prove()
    ctxs = []
    for r in range(world):
        ctx = D.Context(9, t.width, t.ctx_depth, t.loop_depth, rank=r, world=world, grinding=8)
        ctx.upload_owned(t.columns)
        ctxs.append(ctx)
    for _ in range(2):
        assert D.prove_sharded_local(ctxs, t.public_inputs, op.outputs) == expected
    for ctx in ctxs:
        ctx.close()


def test_prove_sharded_fri_layers_with_host_draws(oracle, monkeypatch):
    """The sharded FRI layers are committed without host waits by default (x drawn on the device from the replicated root, the ranks'
    records looked at after the commit phase); DISTAFF_FRI_CHAIN=0 keeps a root read-back, a host draw and a status check per layer --
    the same proof, with sharded layers forced at a small size (DISTAFF_FRI_REPLICATE_LOG)."""
    import distaff_amd as D
    O = oracle
    monkeypatch.setenv("DISTAFF_FRI_REPLICATE_LOG", "9")
    t = O.fibonacci_trace(1 << 10)
    op = O.Prover.from_trace(t, 1, grinding=8)
    expected = op.prove()
    for chain in ("0", None):
        if chain is None:
            monkeypatch.delenv("DISTAFF_FRI_CHAIN", raising=False)
        else:
            monkeypatch.setenv("DISTAFF_FRI_CHAIN", chain)
        ctxs = []
        for r in range(4):
            ctx = D.Context(10, t.width, t.ctx_depth, t.loop_depth, rank=r, world=4, grinding=8)
            ctx.upload_owned(t.columns)
            ctxs.append(ctx)
        for _ in range(2):
            assert D.prove_sharded_local(ctxs, t.public_inputs, op.outputs) == expected
        for ctx in ctxs:
            ctx.close()


def test_prove_sharded_over_rccl_with_one_rank(oracle):
    """The RCCL transport of dst_prove_sharded (librccl.so bound at run time: ncclCommInitRank, all-gathers of host values and of the
    constraint evaluations / FRI layer, the all-to-all as grouped send / receive) with the one rank a single-GPU box offers; more
    ranks run the same calls (the in-process transport covers the partitioning, the driver's 8-GPU run the links)."""
    import distaff_amd as D
    O = oracle
    t = O.fibonacci_trace(1 << 10)
    op = O.Prover.from_trace(t, 1, grinding=8)
    comm = D.Comm.rccl(D.Comm.unique_id(), 0, 1, 0)
    info = comm.describe()                                    # what RCCL itself reports for the live communicator (bench.py prints it for N > 1)
    assert info["transport"] == "rccl" and (info["rccl_ranks"], info["rccl_rank"], info["device"]) == (1, 0, 0) and info["rccl_version"] > 0
    comm.trace(1)
    ctx = D.Context(10, t.width, t.ctx_depth, t.loop_depth, grinding=8)
    ctx.upload(t.columns)
    assert ctx.prove_sharded(comm, t.public_inputs, op.outputs) == op.prove()
    assert [k for k, _, _ in comm.trace(0)][-2:] == ["H", "H"]
    ctx.close()
    comm.close()


@pytest.mark.parametrize("transport", ["ordered", "blocking"])
@pytest.mark.parametrize("world", [2, 8])
def test_thread_rank_transport_issue_order_and_peer_access(oracle, monkeypatch, world, transport):
    """The in-process transport with one Python thread per rank (what dst_prove_sharded_local does with C++ threads): every rank issues
    the same collective sequence, and each reports the peer-access picture of its device -- on this box all ranks share GPU 0, so no
    peer is on another device (on a multi-GPU node every other rank's device is enabled with hipDeviceEnablePeerAccess once, so that the
    device-to-device copies of the exchanges cross xGMI instead of being staged through the host)."""
    import threading
    import distaff_amd as D
    O = oracle
    t = O.fibonacci_trace(1 << 9)
    op = O.Prover.from_trace(t, 1, grinding=8)
    expected = op.prove()
    # `ordered` = the default: collectives only enqueued, events between the ranks' streams, and therefore the two-stream choreography of the
    # RCCL transport; `blocking` = every rank's stream drained around each collective, everything on the main stream
    if transport == "blocking":
        monkeypatch.setenv("DISTAFF_LOCAL_TRANSPORT", "blocking")
    comms = D.Comm.local(world)
    ctxs = []
    for r in range(world):
        ctx = D.Context(9, t.width, t.ctx_depth, t.loop_depth, rank=r, world=world, grinding=8)
        ctx.upload(t.columns)
        ctxs.append(ctx)
        comms[r].trace(1)
    proofs, errors = [None] * world, []

    def run(r):
        try:
            proofs[r] = ctxs[r].prove_sharded(comms[r], t.public_inputs, op.outputs)
        except Exception as e:                                # noqa: BLE001
            errors.append((r, e))
    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    assert all(p == expected for p in proofs)
    records = [cm.trace(0) for cm in comms]
    assert all(rec == records[0] for rec in records) and len(records[0]) > 10
    assert {st for _, _, st in records[0] if st is not None} == ({0, 1} if transport == "ordered" else {0})
    for r, cm in enumerate(comms):
        info = cm.describe()
        assert info["transport"] == "local" and (info["rank"], info["world"]) == (r, world)
        assert info["device"] == 0 and info["peers_other_device"] == 0 and info["peers_enabled"] == 0
    for ctx in ctxs:
        ctx.close()
    for cm in comms:
        cm.close()


def _thread_ranks(ctxs, comms, inputs, outputs, skip=()):
    """one Python thread per rank calling prove_sharded -> (proofs, {rank: DistaffError}, seconds per rank)"""
    import threading
    import time
    import distaff_amd as D
    world = len(ctxs)
    proofs, errors, took = [None] * world, {}, [0.0] * world

    def run(r):
        t0 = time.time()
        try:
            proofs[r] = ctxs[r].prove_sharded(comms[r], inputs, outputs)
        except D.DistaffError as e:
            errors[r] = e
        took[r] = time.time() - t0
    threads = [threading.Thread(target=run, args=(r,)) for r in range(world) if r not in skip]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    return proofs, errors, took


def test_prove_sharded_peer_that_never_arrives(oracle):
    """Containment (the reference panics, lib.rs:32,49,56; the C-ABI promises an error code on every rank): a rank whose peer never enters
    dst_prove_sharded does not wait for ever -- after the communicator's limit (dst_comm_set_timeout) it aborts the communicator and returns
    DST_ERR_COMM, naming the collective it was stuck in; the dead communicator refuses further work at once; with fresh communicators the
    SAME contexts prove."""
    import time
    import distaff_amd as D
    O = oracle
    t = O.fibonacci_trace(128)
    op = O.Prover.from_trace(t, 1, grinding=8)
    expected = op.prove()
    ctxs = []
    for r in range(2):
        ctx = D.Context(7, t.width, t.ctx_depth, t.loop_depth, rank=r, world=2, grinding=8)
        ctx.upload(t.columns)
        ctxs.append(ctx)
    comms = D.Comm.local(2)
    for cm in comms:
        cm.set_timeout(1.5)
    proofs, errors, took = _thread_ranks(ctxs, comms, t.public_inputs, op.outputs, skip=(1,))        # rank 1 never calls
    assert 0 in errors and errors[0].code == D.DST_ERR_COMM, errors
    assert "did not reach collective #0 (all-gather" in str(errors[0]) and "within 1.5" in str(errors[0]), str(errors[0])
    assert 1.0 < took[0] < 20.0, took
    # the group is dead for every rank: the late peer is refused at once instead of waiting in a barrier nobody will complete
    t0 = time.time()
    with pytest.raises(D.DistaffError) as e:
        ctxs[1].prove_sharded(comms[1], t.public_inputs, op.outputs)
    assert e.value.code == D.DST_ERR_COMM and time.time() - t0 < 5.0, str(e.value)
    with pytest.raises(D.DistaffError) as e:
        ctxs[0].prove_sharded(comms[0], t.public_inputs, op.outputs)
    assert e.value.code == D.DST_ERR_COMM and time.time() - t0 < 5.0
    for cm in comms:
        cm.close()
    # a host-side abort (a watchdog that learnt of a dead peer) has the same effect
    comms = D.Comm.local(2)
    comms[0].abort()
    proofs, errors, took = _thread_ranks(ctxs, comms, t.public_inputs, op.outputs)
    assert set(errors) == {0, 1} and all(err.code == D.DST_ERR_COMM for err in errors.values()), errors
    assert max(took) < 5.0
    for cm in comms:
        cm.close()
    comms = D.Comm.local(2)
    proofs, errors, took = _thread_ranks(ctxs, comms, t.public_inputs, op.outputs)
    assert not errors and proofs[0] == expected and proofs[1] == expected
    for cm in comms:
        cm.close()
    for ctx in ctxs:
        ctx.close()


@pytest.mark.parametrize("transport", ["ordered", "blocking"])
@pytest.mark.parametrize("stall_at", [0, 3, 11])
def test_prove_sharded_stalled_collective_is_aborted(oracle, monkeypatch, stall_at, transport):
    """What a collective whose peer never arrives does to a rank -- its stream stops -- reproduced on ONE GPU by the test build's fault
    injection (DISTAFF_TEST_STALL_COLLECTIVE=k@r: before rank r's device collective number k a kernel is queued that holds the stream
    until the communicator is aborted).  The rank's next host wait is a bounded poll: after the limit it aborts the communicator (which
    releases the stream, as ncclCommAbort ends RCCL's kernels), returns DST_ERR_COMM and names the wait and the last collective it issued;
    its peer returns DST_ERR_COMM as well.  With 2 ranks: collectives 0 - 9 = the coefficient all-gathers, 10 = the all-to-all of the trace
    tree's boundary nodes, 11 = the all-gather of its root records, then the first host wait (the trace root).
    `ordered` (the default in-process transport, stream-ordered like RCCL): the collectives behind the stalled one are ENQUEUED regardless, the
    rank notices at the root wait and the last collective it ISSUED is 11 whichever stalled; the peer's stream waits for the stalled rank's
    events, so it times out by itself or is woken by the abort.  `blocking` (DISTAFF_LOCAL_TRANSPORT=blocking): the rank drains its stream
    inside the stalled collective and names exactly that one."""
    import distaff_amd as D
    if not D.load().dst_test_hooks():
        pytest.skip("fault injection exists in the test build only")
    O = oracle
    t = O.fibonacci_trace(128)
    op = O.Prover.from_trace(t, 1, grinding=8)
    expected = op.prove()
    ctxs = []
    for r in range(2):
        ctx = D.Context(7, t.width, t.ctx_depth, t.loop_depth, rank=r, world=2, grinding=8)
        ctx.upload(t.columns)
        ctxs.append(ctx)
    monkeypatch.setenv("DISTAFF_TEST_STALL_COLLECTIVE", "%d@1" % stall_at)
    if transport == "blocking":
        monkeypatch.setenv("DISTAFF_LOCAL_TRANSPORT", "blocking")
    comms = D.Comm.local(2)
    monkeypatch.delenv("DISTAFF_TEST_STALL_COLLECTIVE")
    for cm in comms:
        cm.set_timeout(2.0)
    proofs, errors, took = _thread_ranks(ctxs, comms, t.public_inputs, op.outputs)
    assert set(errors) == {0, 1} and all(e.code == D.DST_ERR_COMM for e in errors.values()), errors
    msg = comms[1].last_error()
    assert "no completion within 2.0 s" in msg and "the communicator was aborted" in msg, msg
    named = stall_at if transport == "blocking" else 11
    assert ("collective #%d (%s" % (named, "all-gather" if named != 10 else "all-to-all")) in msg, msg
    if transport == "ordered":
        assert "the root of the trace tree" in msg, msg
    peer = comms[0].last_error()
    assert "left the group" in peer or "did not reach" in peer or "no completion within" in peer, peer
    assert 1.5 < max(took) < 25.0, took
    for cm in comms:
        cm.close()
    comms = D.Comm.local(2)
    proofs, errors, took = _thread_ranks(ctxs, comms, t.public_inputs, op.outputs)
    assert not errors and proofs[0] == expected and proofs[1] == expected
    for cm in comms:
        cm.close()
    for ctx in ctxs:
        ctx.close()


def test_sharded_phase_and_exchange_times(oracle):
    """dst_phase_ms of the sharded prover comes from events on the prover's stream (the host only enqueues between two waits, its own clock says
    nothing about the phases) and dst_shard_exchange_ms from events around every collective: every phase is accounted for, the nine add up to
    the call's wall time, and the collectives counted are exactly those the communicator's issue-order record lists."""
    import time
    import distaff_amd as D
    O = oracle
    t = O.fibonacci_trace(1 << 12)
    op = O.Prover.from_trace(t, 1, grinding=8)
    expected = op.prove()
    comms = D.Comm.local(2)
    ctxs = []
    for r in range(2):
        ctx = D.Context(12, t.width, t.ctx_depth, t.loop_depth, rank=r, world=2, grinding=8)
        ctx.upload(t.columns)
        ctxs.append(ctx)
    _thread_ranks(ctxs, comms, t.public_inputs, op.outputs)                 # warm-up: buffers, events, code
    for cm in comms:
        cm.trace(1)
    t0 = time.time()
    proofs, errors, took = _thread_ranks(ctxs, comms, t.public_inputs, op.outputs)
    wall_ms = (time.time() - t0) * 1e3
    assert not errors and proofs[0] == expected and proofs[1] == expected
    for r in range(2):
        ph = ctxs[r].phase_ms()
        assert len(ph) == 9 and all(v >= 0 for v in ph), ph
        assert ph[0] > 0 and ph[1] > 0 and ph[2] > 0 and ph[4] > 0 and ph[5] > 0 and ph[6] > 0, ph      # extension, trees, constraints, DEEP, FRI all have device time
        assert 0.5 * took[r] * 1e3 < sum(ph) < 1.05 * wall_ms, (sum(ph), took[r] * 1e3, wall_ms)
        ex = ctxs[r].shard_exchange_ms()
        rec = comms[r].trace(0)
        assert ex["collectives"] == len(rec) and ex["timed_by_events"] == sum(1 for k, _, _ in rec if k != "H"), (ex, len(rec))
        assert ex["coefficients"] > 0 and ex["tree_all_to_all"] > 0 and ex["tree_all_gather"] > 0 and ex["constraint_evaluations"] > 0 and ex["host_values"] > 0, ex
        assert sum(v for k, v in ex.items() if k not in ("collectives", "timed_by_events")) < sum(ph), (ex, ph)
    for cm in comms:
        cm.close()
    for ctx in ctxs:
        ctx.close()


def test_host_side_abort_from_another_thread(oracle, monkeypatch):
    """dst_comm_abort is the host's way out before the limit expires (a watchdog that learnt of a dead peer): called from ANOTHER thread while
    the ranks are inside dst_prove_sharded -- one waiting in a barrier for a peer that never comes, one (test build) polling a stream that a
    stalled collective holds -- it ends both within a moment although their limits are a minute away."""
    import threading
    import time
    import distaff_amd as D
    O = oracle
    t = O.fibonacci_trace(128)
    op = O.Prover.from_trace(t, 1, grinding=8)
    expected = op.prove()
    ctxs = []
    for r in range(2):
        ctx = D.Context(7, t.width, t.ctx_depth, t.loop_depth, rank=r, world=2, grinding=8)
        ctx.upload(t.columns)
        ctxs.append(ctx)
    hooks = bool(D.load().dst_test_hooks()) and "emu" not in D.library_path()
    for case in (["absent peer"] + (["stalled stream"] if hooks else [])):
        if case == "stalled stream":
            monkeypatch.setenv("DISTAFF_TEST_STALL_COLLECTIVE", "4@1")
        comms = D.Comm.local(2)
        monkeypatch.delenv("DISTAFF_TEST_STALL_COLLECTIVE", raising=False)
        for cm in comms:
            cm.set_timeout(60.0)
        victim = 0 if case == "absent peer" else 1
        timer = threading.Timer(1.0, comms[victim].abort)
        t0 = time.time()
        timer.start()
        proofs, errors, took = _thread_ranks(ctxs, comms, t.public_inputs, op.outputs, skip=((1,) if case == "absent peer" else ()))
        timer.join()
        assert errors and all(e.code == D.DST_ERR_COMM for e in errors.values()), (case, errors)
        assert victim in errors and 0.5 < time.time() - t0 < 20.0, (case, took)
        assert "dst_comm_abort" in comms[victim].last_error() or "left the group" in comms[victim].last_error(), comms[victim].last_error()
        for cm in comms:
            cm.close()
    comms = D.Comm.local(2)
    proofs, errors, took = _thread_ranks(ctxs, comms, t.public_inputs, op.outputs)
    assert not errors and proofs[0] == expected and proofs[1] == expected
    for cm in comms:
        cm.close()
    for ctx in ctxs:
        ctx.close()


_STALLED_PEER_WORKER = r"""
import datetime, json, os, sys, time
sys.path.insert(0, %r)
import torch.distributed as dist
import distaff_amd as D
dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=6))       # the HOST's channel carries its own limit
rank, world = dist.get_rank(), dist.get_world_size()
cols, program_hash, result = D.fibonacci_trace(10)
ctx = D.Context(10, 20, 1, 0, device=0, rank=rank, world=world)
ctx.upload(cols)
if rank == 1:
    os.environ["DISTAFF_TEST_STALL_COLLECTIVE"] = "12"                        # somewhere behind the trace tree
comm = D.Comm.over_torch(dist)
comm.set_timeout(3.0)
t0 = time.time()
out = {"rank": rank}
try:
    ctx.prove_sharded(comm, [1, 0], [result])
    out["code"] = 0
except D.DistaffError as e:
    out["code"], out["message"] = e.code, str(e)
out["seconds"] = time.time() - t0
out["comm_error"] = comm.last_error()
json.dump(out, open(os.path.join(sys.argv[1], "result_%%d.json" %% rank), "w"))
if rank == 1:
    time.sleep(9)                                                             # the stalled rank stops answering: rank 0 has only its own limits
os._exit(0)                                                                   # no orderly shutdown of a broken group
"""


def test_stalled_peer_process_over_the_callback_transport(tmp_path):
    """Two OS processes on GPU 0 over the callback transport (gloo carries the collectives): rank 1's stream stalls in the middle of the
    proof (fault injection of the test build) and the process then stops answering.  Rank 1 returns DST_ERR_COMM after ITS communicator's
    limit (3 s: bounded poll, abort); rank 0, inside the host's collective callback, returns DST_ERR_COMM when the host's channel gives up
    (gloo's 6 s) -- nobody hangs, both name the collective."""
    import json
    import os
    import socket
    import subprocess
    import sys
    import distaff_amd as D
    if not D.load().dst_test_hooks():
        pytest.skip("fault injection exists in the test build only")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    script = tmp_path / "worker.py"
    script.write_text(_STALLED_PEER_WORKER % root)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    res = [json.load(open(tmp_path / ("result_%d.json" % r))) for r in range(2)]
    assert res[1]["code"] == D.DST_ERR_COMM and "no completion within 3.0 s" in res[1]["comm_error"] and "collective #12 (all-gather" in res[1]["comm_error"], (res, outs)
    assert 2.5 < res[1]["seconds"] < 20.0, res
    assert res[0]["code"] == D.DST_ERR_COMM and "callback returned" in res[0]["comm_error"] and "collective #12" in res[0]["comm_error"], (res, outs)
    assert res[0]["seconds"] < 60.0, res


def test_prove_sharded_reports_an_invalid_trace_on_every_rank(oracle):
    import distaff_amd as D
    O = oracle
    t = O.fibonacci_trace(128)
    cols = t.columns.copy()
    cols[16, 40, 0] += 1
    op = O.Prover.from_trace(t, 1, grinding=8)
    ctxs = []
    for r in range(4):
        ctx = D.Context(7, t.width, t.ctx_depth, t.loop_depth, rank=r, world=4, grinding=8)
        ctx.upload(cols)
        ctxs.append(ctx)
    with pytest.raises(D.DistaffError) as e:
        D.prove_sharded_local(ctxs, t.public_inputs, op.outputs)
    assert e.value.code == -3 and "step" in str(e.value)
    for ctx in ctxs:
        ctx.close()


def test_prove_sharded_rank_without_a_trace_on_fresh_contexts(oracle):
    """A rank that fails before its first sharded proof (nothing uploaded on a context that has never proved) still takes part in
    every exchange -- its buffers exist from context creation on -- and every rank returns its error instead of waiting for it."""
    import distaff_amd as D
    O = oracle
    t = O.fibonacci_trace(128)
    op = O.Prover.from_trace(t, 1, grinding=8)
    expected = op.prove()
    for missing in (0, 2):
        ctxs = []
        for r in range(4):
            ctx = D.Context(7, t.width, t.ctx_depth, t.loop_depth, rank=r, world=4, grinding=8)
            if r != missing:
                ctx.upload(t.columns)
            ctxs.append(ctx)
        with pytest.raises(D.DistaffError) as e:
            D.prove_sharded_local(ctxs, t.public_inputs, op.outputs)
        assert e.value.code == D.DST_ERR_STATE, str(e.value)
        assert "no trace uploaded" in str(e.value) and "rank %d reported error -4" % missing in str(e.value)
        # the same contexts prove once the rank has its trace
        ctxs[missing].upload(t.columns)
        assert D.prove_sharded_local(ctxs, t.public_inputs, op.outputs) == expected
        for ctx in ctxs:
            ctx.close()
    # a communicator of another shape than the context's is refused before any collective
    ctx = D.Context(7, t.width, t.ctx_depth, t.loop_depth, rank=1, world=2, grinding=8)
    ctx.upload(t.columns)
    comms = D.Comm.local(4)
    with pytest.raises(D.DistaffError) as e:
        ctx.prove_sharded(comms[1], t.public_inputs, op.outputs)
    assert e.value.code == D.DST_ERR_ARG and "rank / world" in str(e.value)
    for cm in comms:
        cm.close()
    ctx.close()


def test_plain_c_host_produces_the_oracle_proof(oracle, tmp_path):
    """examples/prove_fibonacci.c (C99, only include/distaff_hip.h and the shared library) writes the same bytes as the oracle."""
    import subprocess
    from test_host_logic import _compile_c_host
    exe = _compile_c_host(tmp_path)
    out = tmp_path / "proof.bin"
    subprocess.check_call([exe, "10", str(out)])
    t = oracle.fibonacci_trace(1 << 10)
    assert out.read_bytes() == oracle.Prover.from_trace(t, 1).prove()


def test_plain_c_multi_gpu_host_produces_the_oracle_proof(oracle, tmp_path):
    """examples/prove_sharded.c: one process with four thread-ranks on the one GPU of this box (in-process transport), and one process
    per rank over RCCL with the single rank the box offers; both write the oracle's bytes."""
    import subprocess
    from test_host_logic import _compile_c_host
    exe = _compile_c_host(tmp_path, "prove_sharded")
    expected = oracle.Prover.from_trace(oracle.fibonacci_trace(1 << 10), 1).prove()
    out = tmp_path / "local.bin"
    subprocess.check_call([exe, "local", "4", "10", "1", str(out)])
    assert out.read_bytes() == expected
    out = tmp_path / "rccl.bin"
    subprocess.check_call([exe, "rccl", "0", "1", "10", str(tmp_path / "unique.id"), str(out)])
    assert out.read_bytes() == expected


def test_gpu_proofs_match_golden_digests():
    """Committed fixtures (tests/golden/proof_digests.json, made by the oracle): no oracle run involved on the GPU box."""
    import json
    import os
    import distaff_amd as D
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "proof_digests.json")) as f:
        cases = json.load(f)["cases"]
    for c in cases:
        cols, program_hash, result = D.fibonacci_trace(c["log_n"])
        assert program_hash.hex() == c["program_hash"]
        ctx = D.Context(c["log_n"], 20, 1, 0, log_blowup=c["extension_factor"].bit_length() - 1, num_queries=c["num_queries"], grinding=c["grinding_factor"])
        ctx.upload(cols)
        proof = ctx.prove([1, 0], [result])
        ctx.close()
        assert len(proof) == c["proof_bytes"] and D.blake3(proof).hex() == c["proof_blake3"], c


def test_gpu_proofs_of_loop_and_macro_traces_match_golden_digests():
    """Committed fixtures of traces with a `while` loop, a taken else-branch and the lt / isodd macros (tests/golden/isa_traces.npz, made by
    the oracle's VM) and the digests of the oracle's proofs for them: no oracle run involved on the GPU box."""
    import json
    import os
    import distaff_amd as D
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    traces = np.load(os.path.join(here, "isa_traces.npz"))
    for c in json.load(open(os.path.join(here, "isa_proof_digests.json")))["cases"]:
        cols = traces[c["name"]]
        assert cols.shape == (c["width"], c["length"], 2) and D.blake3(np.ascontiguousarray(cols).tobytes()).hex() == c["columns_blake3"]
        ctx = D.Context(c["length"].bit_length() - 1, c["width"], c["ctx_depth"], c["loop_depth"], grinding=c["grinding_factor"])
        ctx.upload(cols)
        proof = ctx.prove([int(v) for v in c["public_inputs"]], [int(v) for v in c["outputs"]])
        ctx.close()
        assert len(proof) == c["proof_bytes"] and D.blake3(proof).hex() == c["proof_blake3"], c["name"]


_GLOO_WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
import torch.distributed as dist
import distaff_amd as D
from distaff_amd import sharded
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
cols, program_hash, result = D.fibonacci_trace(12)
ctx = D.Context(12, 20, 1, 0, device=0, rank=rank, world=world)       # both ranks share GPU 0; shards travel through the host (gloo)
ctx.upload(cols)
comm = sharded.TorchComm(dist, None, device_path=False)
proof = sharded.ShardedProver(ctx, comm).prove([1, 0], [result])
open(os.path.join(%r, "proof_%%d.bin" %% rank), "wb").write(proof)
ctx.close()
dist.destroy_process_group()
print("ok")
'''


def test_sharded_prover_two_processes_over_gloo(tmp_path):
    """The multi-process form of the sharded prover (one process per rank, torch.distributed collectives, host-staged shard hand-off)
    with both processes on the one GPU of the test box: every rank writes the single-context proof."""
    import os
    import socket
    import subprocess
    import sys
    import distaff_amd as D
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER % (root, str(tmp_path)))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    cols, program_hash, result = D.fibonacci_trace(12)
    ctx = D.Context(12, 20, 1, 0)
    ctx.upload(cols)
    expected = ctx.prove([1, 0], [result])
    ctx.close()
    for r in range(2):
        assert (tmp_path / ("proof_%d.bin" % r)).read_bytes() == expected


_CALLBACK_WORKER = r"""
import os, sys
sys.path.insert(0, %r)
import torch.distributed as dist
import distaff_amd as D
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
log_n = int(sys.argv[1])
if sys.argv[3] == "overlap":
    os.environ["DISTAFF_SHARD_FORCE_OVERLAP"] = "1"                         # the stream / event choreography of a stream-ordered transport (RCCL)
cols, program_hash, result = D.fibonacci_trace(log_n)
ctx = D.Context(log_n, 20, 1, 0, device=0, rank=rank, world=world)       # every rank on GPU 0
if rank %% 2:
    ctx.upload(cols)
else:
    ctx.upload_owned(cols)                                                  # only the registers this rank interpolates
comm = D.Comm.over_torch(dist)                                              # dst_comm_init_callbacks, gloo carrying the library's collectives
info = comm.describe()
assert info["transport"] == "callbacks" and info["rank"] == rank and info["world"] == world
comm.trace(1)
for k in range(2):                                                          # twice: buffers of the first proof are reused
    proof = ctx.prove_sharded(comm, [1, 0], [result])
    open(os.path.join(sys.argv[2], "proof_%%d_%%d.bin" %% (rank, k)), "wb").write(proof)
stages = ctx.shard_stage_ms()
assert stages["tree_exchanges"] >= 2 and stages["transport_calls"] > 0
# RCCL deadlocks when two ranks issue the collectives of a communicator in different orders; gloo forgives it.  Every rank's record
# of what it issued (kind, bytes per rank, stream) must be the same sequence.
mine = comm.trace(0)
every = [None] * world
dist.all_gather_object(every, mine)
assert all(e == every[0] for e in every), "ranks issued different collective sequences"
assert len(mine) %% 2 == 0 and mine[:len(mine) // 2] == mine[len(mine) // 2:], "the second proof issued another sequence than the first"
if rank == 0:
    import json
    json.dump(mine[:len(mine) // 2], open(os.path.join(sys.argv[2], "collectives.json"), "w"))
ctx.close(); comm.close()
dist.barrier()
dist.destroy_process_group()
print("ok")
"""


@pytest.mark.parametrize("world,log_n,streams", [(2, 12, "one"), (4, 12, "overlap"), (2, 16, "overlap"), (4, 16, "one"), (8, 12, "one"), (8, 12, "overlap")])
def test_prove_sharded_in_separate_processes_sharing_the_gpu(tmp_path, world, log_n, streams):
    """`world` OS processes, each with its own HIP runtime, context and communicator handle, all on GPU 0, each calling
    dst_prove_sharded: the library's own orchestration (column-split interpolation, k-range tree exchange, status records) with its
    all-gathers / all-to-alls on DEVICE buffers carried between the processes by gloo through the callback transport
    (dst_comm_init_callbacks + dst_comm_copy).  RCCL refuses several ranks on one device, so this is the closest a one-GPU box gets to
    the driver's N-process run; every rank must write the single-context proof, twice.  The ranks also compare the ORDER in which they issued
    the collectives (dst_comm_trace): kind, size and stream of every one -- with `streams` = overlap in the two-stream choreography the
    RCCL transport uses (coefficient all-gathers and the exchange of the constraint evaluations on the collective stream)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    import distaff_amd as D
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    script = tmp_path / "worker.py"
    script.write_text(_CALLBACK_WORKER % root)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), str(log_n), str(tmp_path), streams], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    seq = json.load(open(tmp_path / "collectives.json"))
    kinds = "".join(k for k, _, _ in seq)
    rounds = (20 + world - 1) // world
    assert kinds.startswith("G" * rounds + "AG") and kinds.endswith("HH") and kinds.count("A") >= 2      # coefficient rounds, trace tree (all-to-all + records), ..., openings
    assert {st for _, _, st in seq if st is not None} == ({0, 1} if streams == "overlap" else {0})
    cols, program_hash, result = D.fibonacci_trace(log_n)
    ctx = D.Context(log_n, 20, 1, 0)
    ctx.upload(cols)
    expected = ctx.prove([1, 0], [result])
    ctx.close()
    for r in range(world):
        for k in range(2):
            assert (tmp_path / ("proof_%d_%d.bin" % (r, k))).read_bytes() == expected, (r, k)


@pytest.mark.parametrize("ranks", [2, 8])
def test_bench_with_n_processes_on_one_device(ranks):
    """bench.py launched exactly as the driver launches it for N > 1 (`python -m torch.distributed.run ...`) on a box with ONE GPU: the
    ranks share the device, the collectives of dst_prove_sharded travel through the callback transport over gloo, and rank 0 prints one
    well-formed line that says so (a functional run of the N-process path, not a scaling measurement)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    harness = os.environ.get("DISTAFF_BENCH_ENTRY", os.path.join(root, "bench.py"))      # the CPU run of this test goes through tests/emu/bench_harness.py
    env = dict(os.environ, BENCH_LOG_N="12", BENCH_CPU_LOG_N="8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port), harness, "--gpus", str(ranks), "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert "error" not in d, d
    assert d["n_gpus"] == ranks and d["steps"] == 2 and d["scaling"] == "strong" and d["value"] > 0 and d["proof_verified"]
    assert d["devices"]["shared"] and d["devices"]["ranks"] == ranks and "callback transport" in d["config"]["parallelism"]
    assert d["comm"]["transport"] == "callbacks" and d["comm"]["all_ranks_same_transport"] and len(d["comm"]["device_per_rank"]) == ranks
    assert d["library"]["path"].endswith(".so") and isinstance(d["library"]["env"], dict)
    assert set(d["phase_ms"]) >= {"lde", "trace_merkle", "constraint_eval", "fri", "openings"} and all(v >= 0 for v in d["phase_ms"].values())
    st = d["shard_stage_ms_rank0"]
    assert st and st["transport_calls"] > 0 and st["tree_exchanges"] >= 2
    assert abs(d["value"] - d["config"]["trace_steps"] * 20 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


def test_bench_falls_back_to_the_callback_transport_over_nccl():
    """`bench.py --force-sharded` with the library's own RCCL communicator failing on a rank: the ranks agree on it and the collectives
    of dst_prove_sharded go through the callback transport over the torch.distributed group instead -- here the nccl group, i.e. the
    device-tensor form of Comm.over_torch (all_gather_into_tensor / all_to_all_single on staged device tensors), which the gloo
    tests do not reach.  One rank is all a one-GPU box offers; the calls are the ones N ranks make."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_LOG_N="12", BENCH_SIMULATE_RCCL_FAILURE="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-sharded", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], cwd=root, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    d = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][0])
    assert "error" not in d and d["proof_verified"] and "simulated failure" in d["transport_note"]
    assert "callback transport (torch.distributed nccl" in d["config"]["parallelism"] and d["shard_stage_ms_rank0"]["tree_exchanges"] >= 2


def test_bench_prints_an_error_line_instead_of_hanging():
    """a run that cannot proceed (here: a size the library refuses) ends with ONE JSON line carrying `error` and the stage, and a
    non-zero exit code; a stalled run is ended the same way by the watchdog (BENCH_TIMEOUT_S)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    entry = os.environ.get("DISTAFF_BENCH_ENTRY", os.path.join(root, "bench.py"))
    r = subprocess.run([sys.executable, entry, "--gpus", "1", "--steps", "1", "--warmup", "0", "--log-n", "7", "--log-blowup", "9", "--no-cpu-baseline"],
                       cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert r.returncode != 0 and len(lines) == 1, (r.returncode, r.stdout.decode()[-1000:], r.stderr.decode()[-1000:])
    d = json.loads(lines[0])
    assert d["value"] is None and "extension factor" in d["error"] and d["stage"] == "context"
    env = dict(os.environ, BENCH_TIMEOUT_S="1", BENCH_LOG_N="12")
    r = subprocess.run([sys.executable, entry, "--gpus", "1", "--steps", "2000", "--warmup", "0", "--no-cpu-baseline"], cwd=root, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert r.returncode == 3 and len(lines) == 1 and "no progress" in json.loads(lines[0])["error"], (r.returncode, r.stdout.decode()[-1000:])


def test_bench_config2_line(tmp_path):
    """`bench.py --workload commit`: BASELINE config 2 (LDE + Merkle commit only on random columns) as a bench line of its own, the
    trace root checked against the oracle's on the same columns in the CPU leg"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    entry = os.environ.get("DISTAFF_BENCH_ENTRY", os.path.join(root, "bench.py"))
    r = subprocess.run([sys.executable, entry, "--workload", "commit", "--log-n", "10", "--steps", "2", "--warmup", "1"], cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    d = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][0])
    assert "config 2" in d["config"]["workload"] and set(d["phase_ms"]) == {"lde", "trace_merkle"} and d["cpu_baseline"]["root_hex"] == d["trace_root_hex"]
    assert d["roofline"]["bound"] == "hbm" and "commit" in d["phase_hbm"] and d["cpu_baseline"]["reference_published"]


@pytest.mark.parametrize("instance", ["small", "deep", "generic"])
def test_general_constraint_instances_on_the_fibonacci_trace(oracle, monkeypatch, instance):
    """The depth <= 8 and the two any-shape instances of the constraint kernel, forced onto a trace the depth-4 instance would take:
    same evaluations, same proof (DISTAFF_AIR is read when the context is created)."""
    import distaff_amd as D
    monkeypatch.setenv("DISTAFF_AIR", instance)
    _check_all_phases(oracle, D, oracle.fibonacci_trace(1 << 10))


def test_bench_line_contract(tmp_path):
    """bench.py prints ONE JSON line with the fields the driver and the judge read (small trace, short CPU sample)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_LOG_N="12", BENCH_CPU_LOG_N="8")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["trace_steps"] * d["config"]["registers"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] == 1 and c["value"] > 0 and "sample" in c and c["unit"] == d["unit"]
    # round 6: the CPU figure at the HEADLINE size is quoted beside the bounded sample, and the box says what clock it sustains under this arithmetic
    ref = c["same_size_reference"]
    assert ref["log_n"] == 20 and ref["value"] > 0 and ref["source"].startswith("profiles/") and ref["kind"] == "port"
    assert d["box"]["sclk_nominal_MHz"] == 2400.0 and 500.0 < d["box"]["sclk_under_load_MHz"] <= 2500.0, d["box"]


@pytest.mark.parametrize("steps,log_blowup", [(128, 7), (256, 8)])
def test_blowup_128_and_256(oracle, steps, log_blowup):
    """The largest extension factors the C-ABI accepts, on trace lengths for which the reference's own FRI parameters verify (with a
    256-element remainder bound the reference rejects its own proofs for many (length, extension) pairs beyond 64, e.g. 512 x 256).
    Includes FRI layers with fewer leaves than the extension factor (tree geometry of the batched openings)."""
    import distaff_amd as D
    _check_all_phases(oracle, D, oracle.fibonacci_trace(steps), log_blowup=log_blowup, num_queries=20)
