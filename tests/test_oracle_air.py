"""Pins the oracle's VM subset, trace-row decoding, op flags and constraint pieces against the literals of the
reference's own tests: src/programs/blocks/tests.rs:4-50, src/processor/mod.rs:190-233,
src/stark/trace/trace_state.rs:397-579, src/stark/constraints/decoder/{op_bits,sponge,flow_ops}.rs test modules,
src/stark/constraints/utils.rs:120-160."""
import numpy as np

P = 2**128 - 45 * 2**40 + 1
NOOP, PUSH = 0x7F, 0x1F


def test_sponge_span_hash_kats(oracle):
    O = oracle
    def span_hash(ops, values):
        st = [0, 0, 0, 0]
        for i, (op, v) in enumerate(zip(ops, values)):
            st = O.sponge_round(st, op, v, i)
        return st
    assert span_hash([NOOP] * 15, [0] * 15) == [                # blocks/tests.rs:13-17
        283855050660402859567809346597024356257, 290430270201175202384178252750741838599,
        33642161455895506272337605785278290375, 114906032113415280284656928780040029722]
    ops = [NOOP] * 8 + [PUSH] + [NOOP] * 6
    assert span_hash(ops, [0] * 8 + [1] + [0] * 6) == [         # blocks/tests.rs:29-33
        309939768290184920181146334415666126639, 189522128575407709345588553132211127638,
        300449513105356487315600679523377528535, 201241536410685268433124688525928056833]
    assert span_hash(ops, [0] * 8 + [2] + [0] * 6) == [         # blocks/tests.rs:45-49
        238085520613464573032580920836572617149, 98362585914038709664139524327351111560,
        159064915881679512167348007665307977960, 152057468867502483682425300737565245134]


def test_vm_traces(oracle):
    O = oracle
    t = O.Trace("begin add push.5 mul push.7 end", [1, 2])     # processor/mod.rs:190-210
    assert (t.length, t.width, t.ctx_depth, t.loop_depth) == (64, 17, 0, 0)
    last = t.row(63)
    assert last[0] == 46 and last[5:15] == [1] * 10 and last[15:] == [7, 15]
    assert O.to_arr(last[1:3]).tobytes() == t.program_hash
    t = O.Trace("begin add block push.5 mul push.7 end end", [1, 2])   # processor/mod.rs:212-233
    assert (t.length, t.width, t.ctx_depth, t.loop_depth) == (64, 18, 1, 0)
    last = t.row(63)
    assert last[0] == 60 and last[5:15] == [1] * 10 and last[15] == 0 and last[16:] == [7, 15]
    assert O.to_arr(last[1:3]).tobytes() == t.program_hash


def test_fibonacci_trace_shape(oracle):
    O = oracle
    for n in (128, 1024):
        t = O.fibonacci_trace(n)
        k = n // 16 - 3
        assert (t.width, t.length, t.ctx_depth, t.loop_depth, t.stack_depth) == (20, n, 1, 0, 4)
        last = t.row(n - 1)
        assert last[0] == 16 * k + 44 and last[5:15] == [1] * 10
        a, b = 1, 0                                             # inputs [1, 0]; K iterations of swap dup.2 drop add
        for _ in range(k):
            a, b = (a + b) % P, a
        assert last[16] == a and last[17] == b and last[18:] == [0, 0]
        assert O.to_arr(last[1:3]).tobytes() == t.program_hash
        first = t.row(0)
        assert first[:15] == [0] * 15 and first[16:18] == [1, 0]


def test_op_flags(oracle):
    O = oracle
    f = O.op_flags(1, 0, 2, [101, 1, 2, 3, 4, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 15, 16, 17])      # trace_state.rs:503-516
    assert f["cf"] == [1, 0, 0, 0, 0, 0, 0, 0] and f["ld"] == [0] * 32 and f["hd"] == [0] * 4 and (f["begin"], f["noop"]) == (1, 0)
    f = O.op_flags(1, 0, 2, [101, 1, 2, 3, 4, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 15, 16, 17])      # trace_state.rs:518-530
    assert f["cf"] == [0] * 7 + [1] and f["ld"] == [0] * 31 + [1] and f["hd"] == [0, 0, 0, 1] and (f["begin"], f["noop"]) == (0, 1)
    f = O.op_flags(1, 0, 2, [101, 1, 2, 3, 4, 1, 0, 0, 1, 0, 0, 0, 0, 1, 0, 15, 16, 17])      # trace_state.rs:532-544
    assert f["cf"] == [0, 1, 0, 0, 0, 0, 0, 0] and f["ld"] == [0, 1] + [0] * 30 and f["hd"] == [0, 1, 0, 0]
    f = O.op_flags(1, 0, 2, [101, 1, 2, 3, 4, 1, 1, 0, 1, 1, 0, 0, 0, 0, 1, 15, 16, 17])      # trace_state.rs:546-556
    assert f["cf"] == [0, 0, 0, 1, 0, 0, 0, 0] and f["ld"] == [0, 0, 0, 1] + [0] * 28 and f["hd"] == [0, 0, 1, 0]
    for bits, code in (([0] * 7, 0), ([1] * 7, 127), ([1, 1, 1, 1, 1, 1, 0], 63), ([1, 0, 0, 0, 0, 1, 1], 97)):   # trace_state.rs:559-579
        assert O.op_flags(1, 0, 2, [101, 1, 2, 3, 4, 1, 1, 1] + bits + [15, 16, 17])["op_code"] == code


def new_state(step, flow_op, sponge, ctx, loop):
    # helper of flow_ops.rs tests: decoder registers only, user op = NOOP, one stack register = 0
    row = [step] + list(sponge) + [(flow_op >> i) & 1 for i in range(3)] + [1] * 7 + list(ctx) + list(loop) + [0]
    return row


def test_flow_op_constraints(oracle):
    O = oracle
    BEGIN, TEND, FEND, LOOP, VOID = 1, 2, 3, 4, 7
    def run(which, s1, s2, ctx, lp=0):
        return O.constraint_piece(which, ctx, lp, 1, s1, s2)
    eq = lambda a, b: (a - b) % P
    # op_begin (flow_ops.rs:175-210)
    assert run("begin", new_state(15, BEGIN, [3, 5, 7, 9], [0], []), new_state(16, VOID, [0, 0, 0, 0], [3], []), 1) == [0] * 7
    assert run("begin", new_state(15, BEGIN, [3, 5, 7, 9], [2, 0], []), new_state(16, VOID, [0, 0, 0, 0], [3, 2], []), 2) == [0] * 8
    assert run("begin", new_state(15, BEGIN, [3, 5, 7, 9], [0], []), new_state(16, VOID, [1, 2, 3, 4], [5], []), 1) == [1, 2, 3, 4, 0, eq(3, 5), 0]
    assert run("begin", new_state(15, BEGIN, [3, 5, 7, 9], [2, 0], []), new_state(16, VOID, [1, 2, 3, 4], [5, 6], []), 2) == [1, 2, 3, 4, 0, eq(3, 5), eq(2, 6), 0]
    # op_tend (flow_ops.rs:212-247)
    assert run("tend", new_state(15, TEND, [3, 5, 7, 9], [8], []), new_state(16, VOID, [8, 3, 4, 0], [0], []), 1) == [0] * 7
    assert run("tend", new_state(15, TEND, [3, 5, 7, 9], [8], []), new_state(16, VOID, [1, 2, 3, 4], [8], []), 1) == [7, 1, 0, 4, 0, 8, 0]
    assert run("tend", new_state(15, TEND, [3, 5, 7, 9], [4, 6], []), new_state(16, VOID, [1, 2, 3, 4], [5, 6], []), 2) == [3, 1, 0, 4, 0, 1, 6, 0]
    # op_fend (flow_ops.rs:249-284)
    assert run("fend", new_state(15, FEND, [3, 5, 7, 9], [8, 2], []), new_state(16, VOID, [8, 6, 3, 0], [2, 0], []), 2) == [0] * 8
    assert run("fend", new_state(15, FEND, [3, 5, 7, 9], [8], []), new_state(16, VOID, [1, 3, 2, 4], [8], []), 1) == [7, 0, 1, 4, 0, 8, 0]
    assert run("fend", new_state(15, FEND, [3, 5, 7, 9], [4, 6], []), new_state(16, VOID, [1, 6, 2, 4], [5, 6], []), 2) == [3, 0, 1, 4, 0, 1, 6, 0]
    # op_loop (flow_ops.rs:286-303)
    assert run("loop", new_state(15, LOOP, [3, 5, 7, 9], [0], [0]), new_state(16, VOID, [0, 0, 0, 0], [3], [11]), 1, 1) == [0] * 7
    assert run("loop", new_state(15, LOOP, [3, 5, 7, 9], [0], [0]), new_state(16, VOID, [1, 2, 3, 4], [3], [11]), 1, 1) == [1, 2, 3, 4, 0, 0, 0]


def test_op_bits_constraints(oracle):
    O = oracle
    def state(flow, user, counter):
        return [counter, 0, 0, 0, 0] + [(flow >> i) & 1 for i in range(3)] + [(user >> i) & 1 for i in range(7)] + [0, 0]
    def ev(s, masks, inc):
        nxt = state(7, 0x7F, s[0] + (1 if inc else 0))
        return O.constraint_piece("op_bits", 1, 0, 1, s, nxt, masks)
    ok = [0] * 15
    assert ev(state(7, 0x7F, 1), [0, 0, 0], False) == ok       # op_bits.rs:88-94
    for i in range(3):                                          # op_bits.rs:96-105
        s = state(7, 0x7F, 0); s[5 + i] = 3
        exp = [0] * 10; exp[i] = 6
        assert ev(s, [0, 0, 0], False)[:10] == exp
    for cf in range(1, 8):                                      # op_bits.rs:129-139
        assert ev(state(cf, 0x7F, 1), [0, 0, 0], False) == ok
        assert ev(state(cf, 0x68, 1), [0, 0, 0], False) != ok
    assert ev(state(2, 0x7F, 1), [0, 0, 0], False) == ok and ev(state(2, 0x7F, 1), [1, 0, 0], False) != ok     # op_bits.rs:147-150
    assert ev(state(1, 0x7F, 1), [0, 1, 0], False) != ok                                                         # op_bits.rs:157-159
    assert ev(state(0, 0x1F, 1), [0, 0, 0], True) == ok and ev(state(0, 0x1F, 1), [0, 0, 1], True) != ok         # op_bits.rs:173-176
    # op_bits.rs:179-204: VOID may only be followed by VOID
    assert O.constraint_piece("op_bits", 1, 0, 1, state(0, 0x68, 1), state(7, 0x7F, 2), [0, 0, 0]) == ok
    assert O.constraint_piece("op_bits", 1, 0, 1, state(7, 0x7F, 1), state(0, 0x68, 1), [0, 0, 0]) != ok


def test_hacc_constraint(oracle):
    O = oracle
    tables = O.periodic_tables(1)                               # ext = 1: the raw 16-step constants
    ark0 = O.to_ints(tables[0, :8])
    s1 = [0, 1, 2, 3, 4, 0, 0, 0, 1, 1, 1, 1, 1, 0, 0, 0, 0]    # decoder/sponge.rs:62: push.7
    sponge = O.sponge_round([1, 2, 3, 4], 0x1F, 7, 0)
    s2 = [0] + sponge + [1] * 10 + [0, 7]
    assert O.constraint_piece("hacc", 1, 0, 1, s1, s2, ark0) == [0, 0, 0, 0]
    s2bad = [0] + sponge + [1] * 10 + [0, 6]
    assert O.constraint_piece("hacc", 1, 0, 1, s1, s2bad, ark0) == [0, P - 1, 0, 0]   # sponge.rs:95
    s1 = [0, 1, 2, 3, 4, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 0, 0]    # non-push op (op code 96)
    sponge = O.sponge_round([1, 2, 3, 4], 96, 9, 0)
    s2 = [0] + sponge + [1] * 10 + [0, 9]
    assert O.constraint_piece("hacc", 1, 0, 1, s1, s2, ark0) == [0, P - 9, 0, 0]      # sponge.rs:106


def test_constraints_vanish_on_trace_rows(oracle):
    # AIR self-check (evaluator.rs:152-158): every transition constraint is 0 on consecutive rows of a valid trace
    O = oracle
    for src, inp in (("begin add push.5 mul push.7 end", [1, 2]), ("begin add block push.5 mul push.7 end end", [1, 2]),
                     ("begin swap dup.2 drop add swap dup.2 drop add swap dup.2 drop add end", [1, 0]), (O.fibonacci_source(6), [1, 0])):
        t = O.Trace(src, inp)
        tables = O.periodic_tables(1)
        for step in range(t.length - 1):
            c = tables[step % 16]
            cur, nxt = t.row(step), t.row(step + 1)
            args = (t.ctx_depth, t.loop_depth, t.stack_depth, cur, nxt)
            assert set(O.constraint_piece("op_bits", *args, O.to_ints(c[20:23]))) == {0}, (src, step)
            flags = O.op_flags(t.ctx_depth, t.loop_depth, t.stack_depth, cur)
            total = [0] * (5 + max(t.ctx_depth, 1) + max(t.loop_depth, 1))
            for k, name in enumerate(("hacc", "begin", "tend", "fend", "loop", "wrap", "break", "void")):
                consts = O.to_ints(c[:8]) if name == "hacc" else ()
                part = O.constraint_piece(name, *args, consts, flags["cf"][k])
                total = [(a + b) % P for a, b in zip(total, part + [0] * (len(total) - len(part)))]
            assert set(total) == {0}, (src, step)
            assert set(O.constraint_piece("stack", *args, O.to_ints(c[8:20]))) == {0}, (src, step)


def test_periodic_tables(oracle):
    O = oracle
    t1 = O.periodic_tables(1)
    t8 = O.periodic_tables(8)
    assert t8.shape == (128, 23, 2)
    assert (t8[::8] == t1).all()                                # extension agrees with the raw cycle on the trace sub-domain
    masks = O.to_ints(t1[:, 20:23])
    assert [m[0] for m in masks] == [0] + [1] * 15 and [m[1] for m in masks] == [1] * 15 + [0]
    assert [m[2] for m in masks] == [0] + [1] * 7 + [0] + [1] * 7     # decoder/mod.rs:219-223
