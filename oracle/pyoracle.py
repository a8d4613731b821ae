"""ctypes binding of oracle/_build/liboracle.so (ORACLE = test infrastructure, not product code).

Field elements are Python ints on this side; bulk data are numpy ``uint64`` arrays whose last axis has length 2
(``[..., 0]`` = low 64 bits, ``[..., 1]`` = high 64 bits), i.e. the 16-byte little-endian layout the reference
uses for ``u128`` (``/root/reference/src/utils/mod.rs:35-41``).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")

P = 2**128 - 45 * 2**40 + 1
G = 23953097886125630542083529559205016746


def build(force=False):
    """Compile the C++ restatement (gcc only; a few seconds)."""
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
            for f in os.listdir(_HERE) if f.endswith((".hpp", ".cpp"))):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        L = _lib
        L.orc_last_error.restype = ctypes.c_char_p
        for name in ("orc_vm_execute", "orc_vm_execute_inputs", "orc_vm_execute_ops", "orc_prover_new"):
            getattr(L, name).restype = ctypes.c_void_p
        for name in ("orc_merkle_prove_batch", "orc_prover_get", "orc_program_debug", "orc_stack_run", "orc_program_traverse"):
            getattr(L, name).restype = ctypes.c_long
        L.orc_infer_degree.restype = ctypes.c_size_t
        L.orc_poly_div.restype = ctypes.c_size_t
    return _lib


def last_error():
    return lib().orc_last_error().decode()


# ---- conversions -----------------------------------------------------------------------------------------------
def to_arr(values):
    """ints (any nesting) -> uint64 array [..., 2]"""
    a = np.asarray(values, dtype=object)
    flat = a.reshape(-1)
    out = np.empty((flat.size, 2), dtype=np.uint64)
    for i, v in enumerate(flat):
        v = int(v)
        out[i, 0] = v & 0xFFFFFFFFFFFFFFFF
        out[i, 1] = v >> 64
    return out.reshape(a.shape + (2,))


def to_ints(arr):
    arr = np.asarray(arr, dtype=np.uint64)
    flat = arr.reshape(-1, 2)
    vals = [int(lo) | (int(hi) << 64) for lo, hi in flat]
    return np.array(vals, dtype=object).reshape(arr.shape[:-1]).tolist() if arr.ndim > 1 else vals[0]


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def _el(v):
    return to_arr([v])


# ---- field / polynomials -------------------------------------------------------------------------------------------
_OPS = {"add": 0, "sub": 1, "mul": 2, "inv": 3, "exp": 4, "neg": 5}


def field_op(op, a, b=None):
    a = _c(a)
    b = _c(b) if b is not None else np.zeros_like(a)
    out = np.empty_like(a)
    lib().orc_field_op(_OPS[op], _p(a), _p(b), _p(out), ctypes.c_size_t(a.size // 2))
    return out


def add(a, b): return to_ints(field_op("add", _el(a), _el(b)))[0]
def sub(a, b): return to_ints(field_op("sub", _el(a), _el(b)))[0]
def mul(a, b): return to_ints(field_op("mul", _el(a), _el(b)))[0]
def inv(a): return to_ints(field_op("inv", _el(a)))[0]
def exp(a, e): return to_ints(field_op("exp", _el(a), _el(e)))[0]
def neg(a): return to_ints(field_op("neg", _el(a)))[0]


def inv_many(values):
    a = _c(values); out = np.empty_like(a)
    lib().orc_inv_many(_p(a), _p(out), ctypes.c_size_t(a.size // 2))
    return out


def root_of_unity(order):
    out = np.zeros((1, 2), dtype=np.uint64)
    lib().orc_root_of_unity(ctypes.c_uint64(order), _p(out))
    return to_ints(out)[0]


def power_series(b, n):
    out = np.zeros((n, 2), dtype=np.uint64)
    lib().orc_power_series(_p(_el(b)), _p(out), ctypes.c_size_t(n))
    return out


def fft_eval(p):
    a = _c(p).copy()
    lib().orc_fft_eval(_p(a), ctypes.c_size_t(a.shape[0]))
    return a


def fft_interpolate(v):
    a = _c(v).copy()
    lib().orc_fft_interpolate(_p(a), ctypes.c_size_t(a.shape[0]))
    return a


def poly_eval(p, x):
    a = _c(p); out = np.zeros((1, 2), dtype=np.uint64)
    lib().orc_poly_eval(_p(a), ctypes.c_size_t(a.shape[0]), _p(_el(x)), _p(out))
    return to_ints(out)[0]


def syn_div(a, b):
    a = _c(a).copy()
    lib().orc_syn_div(_p(a), ctypes.c_size_t(a.shape[0]), _p(_el(b)))
    return a


def syn_div_expanded(a, degree, exceptions):
    a = _c(a).copy(); e = to_arr(list(exceptions))
    lib().orc_syn_div_expanded(_p(a), ctypes.c_size_t(a.shape[0]), ctypes.c_size_t(degree), _p(e), ctypes.c_size_t(len(exceptions)))
    return a


def poly_mul(a, b):
    a = _c(a); b = _c(b); out = np.zeros((a.shape[0] + b.shape[0] - 1, 2), dtype=np.uint64)
    lib().orc_poly_mul(_p(a), ctypes.c_size_t(a.shape[0]), _p(b), ctypes.c_size_t(b.shape[0]), _p(out))
    return out


def poly_div(a, b):
    a = _c(a); b = _c(b); out = np.zeros((a.shape[0], 2), dtype=np.uint64)
    n = lib().orc_poly_div(_p(a), ctypes.c_size_t(a.shape[0]), _p(b), ctypes.c_size_t(b.shape[0]), _p(out))
    return out[:n]


def poly_interpolate(xs, ys):
    xs = _c(xs); ys = _c(ys); out = np.zeros_like(xs)
    lib().orc_poly_interpolate(_p(xs), _p(ys), ctypes.c_size_t(xs.shape[0]), _p(out))
    return out


def infer_degree(ev):
    a = _c(ev)
    return lib().orc_infer_degree(_p(a), ctypes.c_size_t(a.shape[0]))


def quartic_transpose(v, stride):
    a = _c(v); rows = a.shape[0] // (4 * stride); out = np.zeros((rows, 4, 2), dtype=np.uint64)
    lib().orc_quartic_transpose(_p(a), ctypes.c_size_t(a.shape[0]), ctypes.c_size_t(stride), _p(out))
    return out


def quartic_interpolate_batch(xs, ys):
    xs = _c(xs); ys = _c(ys); out = np.zeros_like(xs)
    lib().orc_quartic_interpolate_batch(_p(xs), _p(ys), ctypes.c_size_t(xs.shape[0]), _p(out))
    return out


def quartic_evaluate_batch(polys, x):
    a = _c(polys); out = np.zeros((a.shape[0], 2), dtype=np.uint64)
    lib().orc_quartic_evaluate_batch(_p(a), ctypes.c_size_t(a.shape[0]), _p(_el(x)), _p(out))
    return out


# ---- hashing / Merkle --------------------------------------------------------------------------------------------------
def blake3(data):
    data = bytes(data); out = ctypes.create_string_buffer(32)
    lib().orc_blake3(data, ctypes.c_size_t(len(data)), out)
    return out.raw


def merkle_nodes(leaves):
    """leaves: bytes of nleaves*32 -> nodes bytes (nleaves*32; nodes[0] = 0, root = nodes[1])"""
    leaves = bytes(leaves); n = len(leaves) // 32
    out = ctypes.create_string_buffer(n * 32)
    lib().orc_merkle_nodes(leaves, ctypes.c_size_t(n), out)
    return out.raw


def _parse_batch_proof(b):
    import struct
    o = 0
    def u64():
        nonlocal o
        v = struct.unpack_from("<Q", b, o)[0]; o += 8; return v
    def hv():
        nonlocal o
        k = u64(); r = [b[o + 32 * i:o + 32 * (i + 1)] for i in range(k)]; o += 32 * k; return r
    values = hv()
    nodes = [hv() for _ in range(u64())]
    depth = b[o]
    return {"values": values, "nodes": nodes, "depth": depth}


def merkle_prove_batch(leaves, indexes, raw=False):
    leaves = bytes(leaves); n = len(leaves) // 32
    idx = np.asarray(indexes, dtype=np.uint64)
    cap = 64 + 32 * (len(indexes) * (2 + n.bit_length() + 2))
    out = ctypes.create_string_buffer(cap)
    k = lib().orc_merkle_prove_batch(leaves, ctypes.c_size_t(n), _p(idx), ctypes.c_size_t(len(indexes)), out, ctypes.c_size_t(cap))
    if k < 0:
        raise RuntimeError(last_error())
    return out.raw[:k] if raw else _parse_batch_proof(out.raw[:k])


def merkle_verify_batch(root, indexes, proof_bytes):
    idx = np.asarray(indexes, dtype=np.uint64)
    return lib().orc_merkle_verify_batch(bytes(root), _p(idx), ctypes.c_size_t(len(indexes)), bytes(proof_bytes), ctypes.c_size_t(len(proof_bytes))) == 1


# ---- Fiat-Shamir ----------------------------------------------------------------------------------------------------------
def chacha20_words(seed, nwords):
    out = np.zeros(nwords, dtype=np.uint32)
    lib().orc_chacha20_words(bytes(seed), _p(out), ctypes.c_size_t(nwords))
    return out


def prng_vector(seed, n):
    out = np.zeros((n, 2), dtype=np.uint64)
    lib().orc_prng_vector(bytes(seed), _p(out), ctypes.c_size_t(n))
    return out


def query_positions(seed, domain_size, ext, nq):
    out = np.zeros(nq, dtype=np.uint64)
    k = lib().orc_query_positions(bytes(seed), ctypes.c_uint64(domain_size), ctypes.c_uint64(ext), ctypes.c_uint64(nq), _p(out))
    if k < 0:
        raise RuntimeError(last_error())
    return [int(x) for x in out[:k]]


def pow_find(seed, grinding):
    out = ctypes.create_string_buffer(32); nonce = ctypes.c_uint64(0)
    lib().orc_pow_find(bytes(seed), ctypes.c_uint32(grinding), out, ctypes.byref(nonce))
    return out.raw, nonce.value


# ---- Rescue / VM ------------------------------------------------------------------------------------------------------------
def sponge_round(state4, op_code, op_value, step):
    s = to_arr(list(state4))
    lib().orc_sponge_round(_p(s), _p(_el(op_code)), _p(_el(op_value)), ctypes.c_uint64(step))
    return to_ints(s)


def hasher_round(state6, step):
    s = to_arr(list(state6))
    lib().orc_hasher_round(_p(s), ctypes.c_uint64(step))
    return to_ints(s)


# operation codes (processor/opcodes.rs:42-86) and execution hint kinds (opcodes.rs:168-176) as the C entry points take them
OPS = {"assert": 0x60, "asserteq": 0x61, "eq": 0x62, "drop": 0x63, "drop4": 0x64, "choose": 0x65, "choose2": 0x66, "cswap2": 0x67,
       "add": 0x68, "mul": 0x69, "and": 0x6A, "or": 0x6B, "inv": 0x6C, "neg": 0x6D, "not": 0x6E,
       "read": 0x70, "read2": 0x71, "dup": 0x72, "dup2": 0x73, "dup4": 0x74, "pad2": 0x75,
       "swap": 0x78, "swap2": 0x79, "swap4": 0x7A, "roll4": 0x7B, "roll8": 0x7C, "binacc": 0x7D,
       "push": 0x1F, "cmp": 0x3F, "rescr": 0x5F, "begin": 0x00, "noop": 0x7F}
HINTS = {"none": 0, "eq_start": 1, "rc_start": 2, "cmp_start": 3, "pmpath_start": 4, "push_value": 5}


def _vec(values):
    return to_arr(list(values)) if len(values) else np.zeros((0, 2), dtype=np.uint64)


def _opcodes(ops):
    return np.array([OPS[o] if isinstance(o, str) else int(o) for o in ops], dtype=np.uint8)


class Trace:
    """Execution trace of a program (any block kind, the whole instruction set): columns [W, n, 2], ctx/loop depths, program hash.
    ``Trace(source, public_inputs, secret_a, secret_b)`` compiles assembly; ``Trace.from_ops`` builds the single-Span program of the
    reference's ``build_program`` test helper (src/tests/mod.rs:315-335)."""

    def __init__(self, source, public_inputs, secret_a=(), secret_b=(), _handle=None):
        if _handle is None:
            pub, sa, sb = _vec(public_inputs), _vec(secret_a), _vec(secret_b)
            _handle = lib().orc_vm_execute_inputs(source.encode(), _p(pub), ctypes.c_size_t(len(public_inputs)), _p(sa), ctypes.c_size_t(len(secret_a)),
                                                  _p(sb), ctypes.c_size_t(len(secret_b)))
        h = _handle
        if not h:
            raise RuntimeError(last_error())
        W = ctypes.c_uint64(); n = ctypes.c_uint64(); ctx = ctypes.c_uint64(); lp = ctypes.c_uint64()
        lib().orc_trace_dims(ctypes.c_void_p(h), ctypes.byref(W), ctypes.byref(n), ctypes.byref(ctx), ctypes.byref(lp))
        self.width, self.length, self.ctx_depth, self.loop_depth = W.value, n.value, ctx.value, lp.value
        self.columns = np.zeros((self.width, self.length, 2), dtype=np.uint64)
        lib().orc_trace_copy(ctypes.c_void_p(h), _p(self.columns))
        ph = np.zeros((2, 2), dtype=np.uint64)
        lib().orc_trace_program_hash(ctypes.c_void_p(h), _p(ph))
        self.program_hash = ph.tobytes()
        lib().orc_trace_free(ctypes.c_void_p(h))
        self.stack_depth = self.width - 15 - self.ctx_depth - self.loop_depth
        self.public_inputs = list(public_inputs)

    @classmethod
    def from_ops(cls, ops, push_values, public_inputs, secret_a=(), secret_b=()):
        code = _opcodes(ops)
        pv, pub, sa, sb = _vec(push_values), _vec(public_inputs), _vec(secret_a), _vec(secret_b)
        h = lib().orc_vm_execute_ops(_p(code), ctypes.c_size_t(len(code)), _p(pv), ctypes.c_size_t(len(push_values)), _p(pub), ctypes.c_size_t(len(public_inputs)),
                                     _p(sa), ctypes.c_size_t(len(secret_a)), _p(sb), ctypes.c_size_t(len(secret_b)))
        return cls(None, public_inputs, _handle=h or 0)

    def row(self, step):
        return to_ints(self.columns[:, step, :])

    def user_stack(self, step):
        """user stack registers of a row, zero-padded to 8 like TraceState::user_stack() (trace_state.rs:60-66)"""
        v = self.row(step)[15 + self.ctx_depth + self.loop_depth:]
        return v + [0] * (8 - len(v))

    def outputs(self, num_outputs):
        """lib.rs:44-46: the top of the user stack at the last step"""
        return self.user_stack(self.length - 1)[:num_outputs]

    def trace_hash(self):
        """the program hash the VM accumulated (sponge registers 0, 1 of the last row; lib.rs:55)"""
        return to_arr(self.row(self.length - 1)[1:3]).tobytes()


def program_debug(source):
    """``format!("{:?}", program)`` of the compiled program (programs/mod.rs:63-73)"""
    out = ctypes.create_string_buffer(1 << 20)
    k = lib().orc_program_debug(source.encode(), out, ctypes.c_size_t(1 << 20))
    if k < 0:
        raise RuntimeError(last_error())
    return out.value.decode()


def program_traverse(source, conditions):
    """programs/tests/utils.rs traverse + close_block: returns (hash state [4], program hash bytes, step count)"""
    c = _vec(conditions); h = np.zeros((4, 2), dtype=np.uint64); ph = np.zeros((2, 2), dtype=np.uint64)
    k = lib().orc_program_traverse(source.encode(), _p(c), ctypes.c_size_t(len(conditions)), _p(h), _p(ph))
    if k < 0:
        raise RuntimeError(last_error())
    return to_ints(h), ph.tobytes(), k


def hasher_digest(values):
    """utils/hasher.rs:12-26"""
    out = np.zeros((2, 2), dtype=np.uint64)
    lib().orc_hasher_digest(_p(_vec(values)), ctypes.c_size_t(len(values)), _p(out))
    return to_ints(out)


def stack_run(public_inputs, secret_a, secret_b, ops, init_len=16, max_regs=32):
    """The user stack alone (processor/stack/mod.rs).  ``ops``: names / codes or (op, hint_kind, hint_value) tuples.
    Returns (states, depth, max_depth): states[k] = register file after k operations (as wide as the reference's ``registers``)."""
    code, kinds, vals = [], [], []
    for o in ops:
        if isinstance(o, tuple):
            code.append(o[0]); kinds.append(HINTS[o[1]]); vals.append(o[2] if len(o) > 2 else 0)
        else:
            code.append(o); kinds.append(0); vals.append(0)
    code = _opcodes(code); kinds = np.array(kinds, dtype=np.uint8); vals = to_arr(vals) if len(vals) else np.zeros((1, 2), dtype=np.uint64)
    n = len(code)
    states = np.zeros((n + 1, max_regs, 2), dtype=np.uint64); depth = np.zeros(n + 1, dtype=np.uint64); md = np.zeros(n + 1, dtype=np.uint64)
    pub, sa, sb = _vec(public_inputs), _vec(secret_a), _vec(secret_b)
    k = lib().orc_stack_run(_p(pub), ctypes.c_size_t(len(public_inputs)), _p(sa), ctypes.c_size_t(len(secret_a)), _p(sb), ctypes.c_size_t(len(secret_b)),
                            ctypes.c_size_t(init_len), _p(code), _p(kinds), _p(vals), ctypes.c_size_t(n), _p(states), ctypes.c_size_t(max_regs), _p(depth), _p(md))
    if k < 0:
        raise RuntimeError(last_error())
    return [to_ints(states[i, :k]) for i in range(n + 1)], [int(x) for x in depth], [int(x) for x in md]


def fibonacci_source(n_terms):
    return "begin repeat.%d swap dup.2 drop add end end" % (n_terms - 1)


def fibonacci_trace(n_steps):
    """Fibonacci trace that fills exactly n_steps rows (K = n/16 - 3 iterations)."""
    return Trace(fibonacci_source(n_steps // 16 - 3 + 1), [1, 0])


# ---- AIR pieces ---------------------------------------------------------------------------------------------------------------
def op_flags(ctx, lp, st, row):
    out = np.zeros((47, 2), dtype=np.uint64)
    lib().orc_op_flags(ctypes.c_size_t(ctx), ctypes.c_size_t(lp), ctypes.c_size_t(st), _p(to_arr(list(row))), _p(out))
    v = to_ints(out)
    return {"cf": v[:8], "ld": v[8:40], "hd": v[40:44], "begin": v[44], "noop": v[45], "op_code": v[46]}


_PIECES = {"op_bits": 0, "hacc": 1, "begin": 2, "tend": 3, "fend": 4, "loop": 5, "wrap": 6, "break": 7, "void": 8, "stack": 9}


def constraint_piece(which, ctx, lp, st, cur_row, nxt_row, consts=(), flag=1):
    out = np.zeros((64, 2), dtype=np.uint64)
    c = to_arr(list(consts)) if len(consts) else np.zeros((1, 2), dtype=np.uint64)
    k = lib().orc_constraint_piece(_PIECES[which], ctypes.c_size_t(ctx), ctypes.c_size_t(lp), ctypes.c_size_t(st),
                                   _p(to_arr(list(cur_row))), _p(to_arr(list(nxt_row))), _p(c), _p(_el(flag)), _p(out))
    if k < 0:
        raise RuntimeError(last_error())
    return to_ints(out[:k])


def evaluate_at(trace_length, ctx, lp, st, coeffs344, program_hash2, op_count, inputs, outputs, step, x, cur_row, nxt_row):
    """The evaluator at one point of the 8n-point constraint evaluation domain (evaluator.rs:139-162, 181-326): returns
    (transition combination, first-step boundary combination, last-step boundary combination, constraints_vanish)."""
    out = np.zeros((4, 2), dtype=np.uint64)
    i = to_arr(list(inputs)) if len(inputs) else np.zeros((1, 2), dtype=np.uint64)
    o = to_arr(list(outputs)) if len(outputs) else np.zeros((1, 2), dtype=np.uint64)
    r = lib().orc_evaluate_at(ctypes.c_size_t(trace_length), ctypes.c_size_t(ctx), ctypes.c_size_t(lp), ctypes.c_size_t(st), _p(_c(coeffs344)),
                              _p(to_arr(list(program_hash2))), _p(_el(op_count)), _p(i), ctypes.c_size_t(len(inputs)), _p(o), ctypes.c_size_t(len(outputs)),
                              ctypes.c_size_t(step), _p(_el(x)), _p(to_arr(list(cur_row))), _p(to_arr(list(nxt_row))), _p(out))
    if r != 0:
        raise RuntimeError(last_error())
    v = to_ints(out)
    return v[0], v[1], v[2], bool(v[3])


def periodic_tables(ext):
    out = np.zeros((16 * ext, 23, 2), dtype=np.uint64)
    lib().orc_periodic_tables(ctypes.c_size_t(ext), _p(out))
    return out


# ---- prover ---------------------------------------------------------------------------------------------------------------------
GET = {"polys": 0, "registers": 1, "trace_leaves": 2, "trace_nodes": 3, "constraint_draws": 4, "i_evaluations": 5, "f_evaluations": 6,
       "t_evaluations": 7, "constraint_poly": 8, "constraint_evaluations": 9, "constraint_nodes": 10, "deep_draws": 11, "trace_at_z1": 12,
       "trace_at_z2": 13, "composition_poly": 14, "composed_evaluations": 15, "fri_values": 16, "fri_nodes": 17, "fri_special_xs": 18,
       "query_seeds": 19, "pow_nonce": 20, "positions": 21, "proof": 22, "lde_domain": 23, "constraints_ok": 24, "fri_layers": 25,
       "roots": 26, "fri_roots": 27}


class Prover:
    """Step-wise CPU prover keeping every intermediate (oracle/prover.hpp)."""

    def __init__(self, columns, ctx_depth, loop_depth, inputs, outputs, ext=32, num_queries=50, grinding=20):
        cols = _c(columns)
        self.W, self.n = cols.shape[0], cols.shape[1]
        self.ext = ext
        self.N = self.n * ext
        i = to_arr(list(inputs)) if len(inputs) else np.zeros((0, 2), dtype=np.uint64)
        o = to_arr(list(outputs)) if len(outputs) else np.zeros((0, 2), dtype=np.uint64)
        self._h = lib().orc_prover_new(_p(cols), ctypes.c_size_t(self.W), ctypes.c_size_t(self.n), ctypes.c_size_t(ctx_depth), ctypes.c_size_t(loop_depth),
                                       _p(i), ctypes.c_size_t(len(inputs)), _p(o), ctypes.c_size_t(len(outputs)),
                                       ctypes.c_size_t(ext), ctypes.c_size_t(num_queries), ctypes.c_uint32(grinding))
        if not self._h:
            raise RuntimeError(last_error())

    @classmethod
    def from_trace(cls, trace, num_outputs=1, **kw):
        outputs = to_ints(trace.columns[15 + trace.ctx_depth + trace.loop_depth:, trace.length - 1, :])[:num_outputs]
        p = cls(trace.columns, trace.ctx_depth, trace.loop_depth, trace.public_inputs, outputs, **kw)
        p.outputs = outputs
        return p

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_prover_free(ctypes.c_void_p(self._h)); self._h = None

    def step(self, k, override=None):
        if override is not None:
            ov = _c(override)
            r = lib().orc_prover_step(ctypes.c_void_p(self._h), k, _p(ov), ctypes.c_size_t(ov.shape[0]))
        else:
            r = lib().orc_prover_step(ctypes.c_void_p(self._h), k, None, ctypes.c_size_t(0))
        if r != 0:
            raise RuntimeError(last_error())

    def prove(self):
        ms = (ctypes.c_double * 9)()
        if lib().orc_prover_prove(ctypes.c_void_p(self._h), ms) != 0:
            raise RuntimeError(last_error())
        self.phase_ms = list(ms)
        return self.get_bytes("proof")

    def get_bytes(self, what, arg=0):
        k = lib().orc_prover_get(ctypes.c_void_p(self._h), GET[what], ctypes.c_size_t(arg), None, ctypes.c_size_t(0))
        if k < 0:
            raise RuntimeError("oracle: no such intermediate %s[%d]" % (what, arg))
        buf = np.zeros(max(k, 1), dtype=np.uint8)
        lib().orc_prover_get(ctypes.c_void_p(self._h), GET[what], ctypes.c_size_t(arg), _p(buf), ctypes.c_size_t(k))
        return buf[:k].tobytes()

    def get(self, what, arg=0):
        """field-element intermediates as uint64 [..., 2] arrays"""
        b = np.frombuffer(self.get_bytes(what, arg), dtype=np.uint64).reshape(-1, 2)
        if what == "polys":
            return b.reshape(self.W, self.n, 2)
        if what == "registers":
            return b.reshape(self.W, self.N, 2)
        if what == "fri_values":
            return b.reshape(-1, 4, 2)
        return b

    def get_u64(self, what):
        return [int(x) for x in np.frombuffer(self.get_bytes(what), dtype=np.uint64)]


def verify(proof_bytes, program_hash, inputs, outputs):
    """Returns (ok, error_string) -- error strings are the reference's (verifier.rs)."""
    i = to_arr(list(inputs)) if len(inputs) else np.zeros((0, 2), dtype=np.uint64)
    o = to_arr(list(outputs)) if len(outputs) else np.zeros((0, 2), dtype=np.uint64)
    err = ctypes.create_string_buffer(512)
    r = lib().orc_verify(bytes(proof_bytes), ctypes.c_size_t(len(proof_bytes)), bytes(program_hash), _p(i), ctypes.c_size_t(len(inputs)),
                         _p(o), ctypes.c_size_t(len(outputs)), err, ctypes.c_size_t(512))
    return r == 1, err.value.decode()


def fri_prove_verify(evaluations, claimed_degree, drop_first_evaluation=False):
    ev = _c(evaluations); err = ctypes.create_string_buffer(512)
    r = lib().orc_fri_prove_verify(_p(ev), ctypes.c_size_t(ev.shape[0]), ctypes.c_size_t(claimed_degree), int(drop_first_evaluation), err, ctypes.c_size_t(512))
    if r < 0:
        raise RuntimeError(last_error())
    return r == 1, err.value.decode()
