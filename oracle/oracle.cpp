// ORACLE (test infrastructure, NOT product code) -- C entry points over the CPU restatement, loaded with ctypes by
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg (and by nothing else).
// Field elements cross this boundary as 16-byte little-endian integers.
#include "verifier.hpp"
#include <chrono>
#include <cstdio>

using namespace orc;

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static thread_local std::string g_error;
static int fail(const std::exception& e) { g_error = e.what(); return -1; }

extern "C" {

const char* orc_last_error() { return g_error.c_str(); }

// ---- field / polynomials ------------------------------------------------------------------------
// op: 0 add, 1 sub, 2 mul, 3 inv (b ignored), 4 exp (b = exponent), 5 neg
void orc_field_op(int op, const u128* a, const u128* b, u128* out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        switch (op) {
            case 0: out[i] = add(a[i], b[i]); break;
            case 1: out[i] = sub(a[i], b[i]); break;
            case 2: out[i] = mul(a[i], b[i]); break;
            case 3: out[i] = inv(a[i]); break;
            case 4: out[i] = exp(a[i], b[i]); break;
            case 5: out[i] = neg(a[i]); break;
        }
    }
}
void orc_inv_many(const u128* v, u128* out, size_t n) { inv_many_fill(v, out, n); }
void orc_root_of_unity(uint64_t order, u128* out) { *out = get_root_of_unity(order); }
void orc_power_series(const u128* b, u128* out, size_t n) { vec r = get_power_series(*b, n); std::copy(r.begin(), r.end(), out); }
void orc_fft_eval(u128* p, size_t n) { vec v(p, p + n); eval_fft(v); std::copy(v.begin(), v.end(), p); }
void orc_fft_interpolate(u128* p, size_t n) { vec v(p, p + n); interpolate_fft(v); std::copy(v.begin(), v.end(), p); }
void orc_poly_eval(const u128* p, size_t n, const u128* x, u128* out) { *out = poly_eval(p, n, *x); }
void orc_syn_div(u128* a, size_t n, const u128* b) { syn_div_in_place(a, n, *b); }
void orc_syn_div_expanded(u128* a, size_t n, size_t degree, const u128* exceptions, size_t ne) {
    vec v(a, a + n); syn_div_expanded_in_place(v, degree, vec(exceptions, exceptions + ne)); std::copy(v.begin(), v.end(), a);
}
void orc_poly_mul(const u128* a, size_t na, const u128* b, size_t nb, u128* out) { vec r = poly_mul(vec(a, a + na), vec(b, b + nb)); std::copy(r.begin(), r.end(), out); }
size_t orc_poly_div(const u128* a, size_t na, const u128* b, size_t nb, u128* out) { vec r = poly_div(vec(a, a + na), vec(b, b + nb)); std::copy(r.begin(), r.end(), out); return r.size(); }
void orc_poly_interpolate(const u128* xs, const u128* ys, size_t n, u128* out) { vec r = poly_interpolate(vec(xs, xs + n), vec(ys, ys + n)); std::copy(r.begin(), r.end(), out); }
size_t orc_infer_degree(const u128* ev, size_t n) { return infer_degree(vec(ev, ev + n)); }
void orc_quartic_transpose(const u128* v, size_t len, size_t stride, u128* out) {
    auto r = quartic_transpose(v, len, stride);
    for (size_t i = 0; i < r.size(); i++) for (int k = 0; k < 4; k++) out[4 * i + k] = r[i][k];
}
void orc_quartic_interpolate_batch(const u128* xs, const u128* ys, size_t rows, u128* out) {
    std::vector<quad> x(rows), y(rows);
    for (size_t i = 0; i < rows; i++) for (int k = 0; k < 4; k++) { x[i][k] = xs[4 * i + k]; y[i][k] = ys[4 * i + k]; }
    auto r = quartic_interpolate_batch(x, y);
    for (size_t i = 0; i < rows; i++) for (int k = 0; k < 4; k++) out[4 * i + k] = r[i][k];
}
void orc_quartic_evaluate_batch(const u128* polys, size_t rows, const u128* x, u128* out) {
    for (size_t i = 0; i < rows; i++) { quad q = {polys[4 * i], polys[4 * i + 1], polys[4 * i + 2], polys[4 * i + 3]}; out[i] = quartic_eval(q, *x); }
}

// ---- hashing / Merkle -------------------------------------------------------------------------------
void orc_blake3(const uint8_t* in, size_t len, uint8_t* out) { blake3(in, len, out); }
void orc_merkle_nodes(const uint8_t* leaves, size_t nleaves, uint8_t* nodes_out) {
    std::vector<hash32> l(nleaves);
    memcpy(l.data(), leaves, nleaves * 32);
    auto nodes = build_merkle_nodes(l);
    memcpy(nodes_out, nodes.data(), nodes.size() * 32);
}
static void write_batch_proof(const BatchMerkleProof& pr, ByteWriter& w) { w.hv(pr.values); w.hvv(pr.nodes); w.u8(pr.depth); }
static BatchMerkleProof read_batch_proof(ByteReader& r) { BatchMerkleProof p; p.values = r.hv(); p.nodes = r.hvv(); p.depth = r.u8(); return p; }
// serialised BatchMerkleProof: values (u64 count + 32-byte items), nodes (u64 count of lists, each u64 count + items), depth u8
long orc_merkle_prove_batch(const uint8_t* leaves, size_t nleaves, const uint64_t* idx, size_t nidx, uint8_t* out, size_t cap) {
    try {
        std::vector<hash32> l(nleaves);
        memcpy(l.data(), leaves, nleaves * 32);
        MerkleTree t(l);
        ByteWriter w;
        write_batch_proof(t.prove_batch(std::vector<size_t>(idx, idx + nidx)), w);
        if (w.b.size() > cap) return -(long)w.b.size();
        memcpy(out, w.b.data(), w.b.size());
        return (long)w.b.size();
    } catch (const std::exception& e) { return fail(e); }
}
int orc_merkle_verify_batch(const uint8_t* root, const uint64_t* idx, size_t nidx, const uint8_t* proof, size_t len) {
    try {
        ByteReader r(proof, len);
        BatchMerkleProof p = read_batch_proof(r);
        hash32 rt; memcpy(rt.data(), root, 32);
        return MerkleTree::verify_batch(rt, std::vector<size_t>(idx, idx + nidx), p) ? 1 : 0;
    } catch (const std::exception& e) { return fail(e); }
}

// ---- Fiat-Shamir randomness ------------------------------------------------------------------------------
void orc_chacha20_words(const uint8_t* seed, uint32_t* out, size_t nwords) { ChaCha20Rng g(seed); for (size_t i = 0; i < nwords; i++) out[i] = g.next_u32(); }
void orc_prng_vector(const uint8_t* seed, u128* out, size_t n) { vec r = prng_vector(seed, n); std::copy(r.begin(), r.end(), out); }
int orc_query_positions(const uint8_t* seed, uint64_t domain_size, uint64_t ext, uint64_t nq, uint64_t* out) {
    try {
        hash32 s; memcpy(s.data(), seed, 32);
        ProofOptions o; o.extension_factor = ext; o.num_queries = nq;
        auto p = compute_query_positions(s, domain_size, o);
        for (size_t i = 0; i < p.size(); i++) out[i] = p[i];
        return (int)p.size();
    } catch (const std::exception& e) { return fail(e); }
}
void orc_pow_find(const uint8_t* seed, uint32_t grinding, uint8_t* out_seed, uint64_t* nonce) {
    hash32 s; memcpy(s.data(), seed, 32);
    ProofOptions o; o.grinding_factor = grinding;
    auto r = find_pow_nonce(s, o);
    memcpy(out_seed, r.first.data(), 32); *nonce = r.second;
}

// ---- Rescue rounds, VM, program hash -----------------------------------------------------------------------------
void orc_sponge_round(u128* state4, const u128* op_code, const u128* op_value, uint64_t step) { sponge_apply_round(state4, *op_code, *op_value, step); }
void orc_hasher_round(u128* state6, uint64_t step) { hasher_apply_round(state6, step); }

struct TraceHandle { ExecutionTrace t; Program p; };
static void* run_program(Program&& p, const u128* pub, size_t npub, const u128* sa, size_t na, const u128* sb, size_t nb) {
    auto h = new TraceHandle();
    try {
        h->p = std::move(p);
        h->t = vm_execute(h->p, ProgramInputs(vec(pub, pub + npub), vec(sa, sa + na), vec(sb, sb + nb)));
        return h;
    } catch (...) { delete h; throw; }
}
void* orc_vm_execute(const char* source, const u128* inputs, size_t nin) {
    try { Assembler a; return run_program(a.compile(source), inputs, nin, nullptr, 0, nullptr, 0); }
    catch (const std::exception& e) { fail(e); return nullptr; }
}
// assembly source + public inputs + the two secret tapes (programs/inputs.rs:12)
void* orc_vm_execute_inputs(const char* source, const u128* pub, size_t npub, const u128* sa, size_t na, const u128* sb, size_t nb) {
    try { Assembler a; return run_program(a.compile(source), pub, npub, sa, na, sb, nb); }
    catch (const std::exception& e) { fail(e); return nullptr; }
}
// a single-Span program from op codes, PUSH values assigned in order (the reference tests' build_program, src/tests/mod.rs:315-335)
void* orc_vm_execute_ops(const uint8_t* ops, size_t nops, const u128* push_values, size_t npush, const u128* pub, size_t npub,
                         const u128* sa, size_t na, const u128* sb, size_t nb) {
    try {
        std::vector<UserOp> code; HintMap hints; size_t j = 0;
        for (size_t i = 0; i < nops; i++) {
            if (!is_user_op(ops[i])) throw std::runtime_error("not a user operation code");
            code.push_back((UserOp)ops[i]);
            if (ops[i] == OP_PUSH) { if (j >= npush) throw std::runtime_error("not enough push values"); hints[i] = OpHint::push(push_values[j++]); }
        }
        if (j != npush) throw std::runtime_error("too many push values");
        return run_program(Program::from_root_blocks({make_span(code, hints)}), pub, npub, sa, na, sb, nb);
    } catch (const std::exception& e) { fail(e); return nullptr; }
}
void orc_trace_dims(void* h, uint64_t* W, uint64_t* n, uint64_t* ctx, uint64_t* lp) {
    auto t = (TraceHandle*)h; *W = t->t.registers.size(); *n = t->t.registers[0].size(); *ctx = t->t.ctx_depth; *lp = t->t.loop_depth;
}
void orc_trace_copy(void* h, u128* out) { auto t = (TraceHandle*)h; size_t n = t->t.registers[0].size(); for (size_t c = 0; c < t->t.registers.size(); c++) std::copy(t->t.registers[c].begin(), t->t.registers[c].end(), out + c * n); }
void orc_trace_program_hash(void* h, u128* out2) { auto t = (TraceHandle*)h; out2[0] = t->p.hash[0]; out2[1] = t->p.hash[1]; }
void orc_trace_free(void* h) { delete (TraceHandle*)h; }

// `{:?}` of the compiled program (programs/mod.rs:63-73), for the reference's assembler tests
long orc_program_debug(const char* source, char* out, size_t cap) {
    try { Assembler a; std::string s = a.compile(source).debug(); if (s.size() + 1 > cap) return -(long)s.size(); memcpy(out, s.c_str(), s.size() + 1); return (long)s.size(); }
    catch (const std::exception& e) { return fail(e); }
}
void orc_hasher_digest(const u128* values, size_t n, u128* out2) { auto d = hasher_digest(vec(values, values + n)); out2[0] = d[0]; out2[1] = d[1]; }

// The user stack alone (processor/stack/mod.rs), for the reference's stack unit tests: executes ops[i] with hint (kind, value) i and
// writes after every operation the full register file (states: (nops + 1) x max_regs, row 0 = initial state) plus depth / max_depth.
// Returns the number of registers, or -1 with the reference's panic text in orc_last_error().
long orc_stack_run(const u128* pub, size_t npub, const u128* sa, size_t na, const u128* sb, size_t nb, size_t init_len,
                   const uint8_t* ops, const uint8_t* hint_kinds, const u128* hint_values, size_t nops,
                   u128* states, size_t max_regs, uint64_t* depth, uint64_t* max_depth) {
    try {
        StackVM s(ProgramInputs(vec(pub, pub + npub), vec(sa, sa + na), vec(sb, sb + nb)), init_len);
        auto snap = [&](size_t k) {
            for (size_t r = 0; r < max_regs; r++) states[k * max_regs + r] = r < s.registers.size() ? s.registers[r][s.step] : 0;
            depth[k] = s.depth; max_depth[k] = s.max_depth;
        };
        snap(0);
        for (size_t i = 0; i < nops; i++) {
            if (!is_user_op(ops[i])) throw std::runtime_error("not a user operation code");
            s.execute((UserOp)ops[i], OpHint::of((HintKind)hint_kinds[i], hint_values[i]));
            if (s.registers.size() > max_regs) throw std::runtime_error("orc_stack_run: register file larger than max_regs");
            snap(i + 1);
        }
        return (long)s.registers.size();
    } catch (const std::exception& e) { return fail(e); }
}

// The reference's independent walk over a program (its test utility programs/tests/utils.rs:10-165: traverse, traverse_span,
// close_block, traverse_loop) restated, so that programs/tests/mod.rs can be transcribed: `conditions` is the stack the walk pops
// branch / loop decisions from (last element first).  Writes the hash state after the final close_block and the step count.
namespace {
struct Walk {
    vec stack;
    u128 pop() { if (stack.empty()) throw std::runtime_error("traverse: condition stack is empty"); u128 v = stack.back(); stack.pop_back(); return v; }
    size_t span(const Span& b, u128* hash, bool is_first, size_t step) {
        if (!is_first) { hash_op(hash, OP_NOOP, 0, step); step++; }
        for (size_t i = 0; i < b.length(); i++) { hash_op(hash, b.ops[i], b.get_hint(i).value(), step); step++; }
        return step;
    }
    size_t close(u128* hash, u128 parent_hash, u128 sibling_hash, bool is_true_branch, size_t step) {
        hash_op(hash, OP_NOOP, 0, step); step++;
        step++;  // TEND / FEND
        if (is_true_branch) { hash[1] = hash[0]; hash[0] = parent_hash; hash[2] = sibling_hash; hash[3] = 0; }
        else { hash[2] = hash[0]; hash[0] = parent_hash; hash[1] = sibling_hash; hash[3] = 0; }
        for (size_t i = 0; i < HACC_NUM_ROUNDS; i++) { hash_op(hash, OP_NOOP, 0, step); step++; }
        return step;
    }
    size_t loop(const Block& b, u128* hash, size_t step) {
        step++;  // LOOP
        u128 state[4] = {0, 0, 0, 0};
        for (;;) {
            step = blocks(b.body, state, step);
            u128 c = pop();
            if (c > 1) throw std::runtime_error("cannot exit loop based on a non-binary condition");
            if (state[0] != loop_image(b)) throw std::runtime_error("loop image didn't match loop body hash");
            step++;  // BREAK / WRAP
            if (c == 0) break;
            for (auto& v : state) v = 0;
        }
        step = span(b.alt[0].span, state, true, step);
        step = close(state, hash[0], loop_skip_hash(b), true, step);
        std::copy(state, state + 4, hash);
        return step;
    }
    size_t blocks(const std::vector<Block>& bl, u128* hash, size_t step) {
        if (!bl[0].is_span()) throw std::runtime_error("first block in a sequence must be a Span block");
        step = span(bl[0].span, hash, true, step);
        for (size_t k = 1; k < bl.size(); k++) {
            const Block& b = bl[k];
            u128 state[4] = {0, 0, 0, 0};
            switch (b.kind) {
                case B_SPAN: step = span(b.span, hash, false, step); break;
                case B_GROUP:
                    step++;  // BEGIN
                    step = blocks(b.body, state, step);
                    step = close(state, hash[0], 0, true, step);
                    std::copy(state, state + 4, hash); break;
                case B_SWITCH: {
                    step++;
                    u128 c = pop();
                    if (c == 0) { step = blocks(b.alt, state, step); step = close(state, hash[0], seq_hash(b.body), false, step); }
                    else if (c == 1) { step = blocks(b.body, state, step); step = close(state, hash[0], seq_hash(b.alt), true, step); }
                    else throw std::runtime_error("cannot select a branch based on a non-binary condition");
                    std::copy(state, state + 4, hash); break;
                }
                case B_LOOP: {
                    u128 c = pop();
                    if (c == 0) {
                        step++;
                        step = blocks(b.alt, state, step);
                        step = close(state, hash[0], loop_body_hash(b), false, step);
                        std::copy(state, state + 4, hash);
                    } else if (c == 1) step = loop(b, hash, step);
                    else throw std::runtime_error("cannot enter loop based on a non-binary condition");
                    break;
                }
            }
        }
        return step;
    }
};
}
long orc_program_traverse(const char* source, const u128* conditions, size_t ncond, u128* hash4, u128* program_hash2) {
    try {
        Assembler a; Program p = a.compile(source);
        Walk w; w.stack.assign(conditions, conditions + ncond);
        u128 h[4] = {0, 0, 0, 0};
        size_t step = w.blocks(p.root.body, h, 0);
        step = w.close(h, 0, 0, true, step);
        std::copy(h, h + 4, hash4);
        program_hash2[0] = p.hash[0]; program_hash2[1] = p.hash[1];
        return (long)step;
    } catch (const std::exception& e) { return fail(e); }
}

// ---- AIR pieces --------------------------------------------------------------------------------------------------------
// out: cf[8] ld[32] hd[4] begin noop op_code  (47 elements)
void orc_op_flags(size_t ctx, size_t lp, size_t st, const u128* row, u128* out) {
    TraceState s = TraceState::from_vec(ctx, lp, st, vec(row, row + 15 + ctx + lp + st));
    s.set_op_flags();
    std::copy(s.cf_flags, s.cf_flags + 8, out); std::copy(s.ld_flags, s.ld_flags + 32, out + 8); std::copy(s.hd_flags, s.hd_flags + 4, out + 40);
    out[44] = s.begin_flag; out[45] = s.noop_flag; out[46] = s.op_code();
}
// which: 0 op_bits (masks = 3 values, out 15), 1 hacc (ark = 8 values, out 4), 2..8 begin,tend,fend,loop,wrap,break,void
// (out = 5 + max(ctx,1) + max(loop,1)), 9 stack constraints (ark = 12 values, out = 2 + stack_depth)
int orc_constraint_piece(int which, size_t ctx, size_t lp, size_t st, const u128* cur_row, const u128* nxt_row, const u128* consts, const u128* flag, u128* out) {
    try {
        size_t w = 15 + ctx + lp + st;
        TraceState c = TraceState::from_vec(ctx, lp, st, vec(cur_row, cur_row + w));
        TraceState n = TraceState::from_vec(ctx, lp, st, vec(nxt_row, nxt_row + w));
        size_t cl = std::max(ctx, MIN_CONTEXT_DEPTH), ll = std::max(lp, MIN_LOOP_DEPTH);
        if (which == 0) { vec r(15, 0); enforce_op_bits(r.data(), c, n, consts); std::copy(r.begin(), r.end(), out); return 15; }
        if (which == 1) { vec r(4, 0); enforce_hacc(r.data(), c, n, consts, *flag); std::copy(r.begin(), r.end(), out); return 4; }
        if (which >= 2 && which <= 8) {
            vec r(5 + cl + ll, 0);
            FlowCtx fc(r.data(), c, n);
            switch (which) { case 2: fc.begin(*flag); break; case 3: fc.tend(*flag); break; case 4: fc.fend(*flag); break; case 5: fc.loop(*flag); break;
                             case 6: fc.wrap(*flag); break; case 7: fc.brk(*flag); break; case 8: fc.vd(*flag); break; }
            std::copy(r.begin(), r.end(), out); return (int)r.size();
        }
        if (which == 9) { vec r(2 + st, 0); enforce_stack_constraints(c, n, consts, r.data(), r.size()); std::copy(r.begin(), r.end(), out); return (int)r.size(); }
        return -1;
    } catch (const std::exception& e) { return fail(e); }
}
// periodic constant tables over a cycle of 16*ext steps: out = [cycle][8 sponge ark | 12 hasher ark | 3 masks]
void orc_periodic_tables(size_t ext, u128* out) {
    DecoderAir d(16, ext, 1, 0); StackAir s(16, ext, 1);
    size_t cyc = 16 * ext;
    for (size_t i = 0; i < cyc; i++) {
        for (int k = 0; k < 8; k++) out[i * 23 + k] = d.ark_evals[k][i];
        for (int k = 0; k < 12; k++) out[i * 23 + 8 + k] = s.ark_evals[k][i];
        for (int k = 0; k < 3; k++) out[i * 23 + 20 + k] = d.mask_evals[k][i];
    }
}

// The evaluator at ONE point of the constraint evaluation domain (evaluator.rs:139-162 evaluate_transition, :181-326 evaluate_boundaries,
// as driven by prover.rs:53-64): `step` = index in the 8n-point domain, x = w_{8n}^step, cur / nxt = the LDE rows at positions
// step * (B/8) and step * (B/8) + B.  out = [transition combination, first-step boundary combination, last-step boundary combination,
// 1 when all transition constraints vanish (only meaningful on trace steps)].  For sampled parity at sizes the whole-domain loop of the
// oracle cannot finish in seconds.
int orc_evaluate_at(size_t trace_length, size_t ctx, size_t lp, size_t st, const u128* coeffs344, const u128* prog_hash2, const u128* op_count,
                    const u128* inputs, size_t nin, const u128* outputs, size_t nout, size_t step, const u128* x,
                    const u128* cur_row, const u128* nxt_row, u128* out4) {
    try {
        ConstraintCoefficients cc;
        cc.init(vec(coeffs344, coeffs344 + 2 * NUM_CONSTRAINTS), ctx, lp, st);
        Evaluator ev(trace_length, MAX_CONSTRAINT_DEGREE, ctx, lp, st, trace_length * MAX_CONSTRAINT_DEGREE, cc,
                     vec(prog_hash2, prog_hash2 + PROGRAM_DIGEST_SIZE), *op_count, vec(inputs, inputs + nin), vec(outputs, outputs + nout));
        const size_t w = 15 + ctx + lp + st;
        TraceState c = TraceState::from_vec(ctx, lp, st, vec(cur_row, cur_row + w));
        TraceState n = TraceState::from_vec(ctx, lp, st, vec(nxt_row, nxt_row + w));
        bool ok = true;
        out4[0] = ev.evaluate_transition(c, n, *x, step, &ok);
        ev.evaluate_boundaries(c, *x, out4[1], out4[2]);
        out4[3] = ok ? 1 : 0;
        return 0;
    } catch (const std::exception& e) { return fail(e); }
}

// ---- prover object ------------------------------------------------------------------------------------------------------
struct ProverHandle { Prover* p; StarkProof proof; std::vector<uint8_t> proof_bytes; bool has_proof = false; };

void* orc_prover_new(const u128* cols, size_t W, size_t n, size_t ctx, size_t lp, const u128* inputs, size_t nin, const u128* outputs, size_t nout,
                     size_t ext, size_t nq, uint32_t grind) {
    try {
        std::vector<vec> t(W);
        for (size_t c = 0; c < W; c++) t[c].assign(cols + c * n, cols + (c + 1) * n);
        ProofOptions o; o.extension_factor = ext; o.num_queries = nq; o.grinding_factor = grind;
        auto h = new ProverHandle();
        h->p = new Prover(std::move(t), ctx, lp, vec(inputs, inputs + nin), vec(outputs, outputs + nout), o);
        return h;
    } catch (const std::exception& e) { fail(e); return nullptr; }
}
void orc_prover_free(void* hv) { auto h = (ProverHandle*)hv; delete h->p; delete h; }

// step: 1..9 as in prover.rs; `override`/`nover`: caller-supplied challenges for steps 3 (344 draws) and 6 (517 draws incl. z)
int orc_prover_step(void* hv, int step, const u128* over, size_t nover) {
    auto h = (ProverHandle*)hv;
    try {
        vec ov; if (over) ov.assign(over, over + nover);
        switch (step) {
            case 1: h->p->step1_extend(); break;
            case 2: h->p->step2_trace_tree(); break;
            case 3: h->p->step3_evaluate_constraints(over ? &ov : nullptr); break;
            case 4: h->p->step4_combine(); break;
            case 5: h->p->step5_constraint_tree(); break;
            case 6: h->p->step6_deep_composition(over ? &ov : nullptr); break;
            case 7: h->p->step7_fri(); break;
            case 8: h->p->step8_queries(); break;
            case 9: h->proof = h->p->step9_build_proof(); h->proof_bytes = serialize_proof(h->proof); h->has_proof = true; break;
            default: return -1;
        }
        return 0;
    } catch (const std::exception& e) { return fail(e); }
}
int orc_prover_prove(void* hv, double* phase_ms9) {
    auto h = (ProverHandle*)hv;
    try {
        h->proof = h->p->prove(now_ms);
        h->proof_bytes = serialize_proof(h->proof); h->has_proof = true;
        if (phase_ms9) for (int i = 0; i < 9; i++) phase_ms9[i] = h->p->phase_ms[i];
        return 0;
    } catch (const std::exception& e) { return fail(e); }
}

static long put(const void* src, size_t bytes, uint8_t* out, size_t cap) { if (out) { if (bytes > cap) return -(long)bytes; memcpy(out, src, bytes); } return (long)bytes; }
static long put_cols(const std::vector<vec>& cols, uint8_t* out, size_t cap) {
    size_t n = cols.empty() ? 0 : cols[0].size(), bytes = cols.size() * n * 16;
    if (!out) return (long)bytes;
    if (bytes > cap) return -(long)bytes;
    for (size_t c = 0; c < cols.size(); c++) memcpy(out + c * n * 16, cols[c].data(), n * 16);
    return (long)bytes;
}
// Copies an intermediate into `out` (or returns its size in bytes when out == NULL). `what`:
//  0 polys (W x n)            1 registers / LDE (W x N)      2 trace leaves (N x 32)      3 trace tree nodes (N x 32)
//  4 constraint coeff draws   5 i_evaluations               6 f_evaluations             7 t_evaluations (8n each)
//  8 constraint poly (8n)     9 constraint evaluations (N)  10 constraint tree nodes     11 deep draws (517, [0] = z)
// 12 trace_at_z1             13 trace_at_z2                 14 composition poly (8n)    15 composed evaluations (N)
// 16 FRI layer `arg` values (rows x 4)   17 FRI layer `arg` tree nodes   18 FRI special xs   19 query seed0|seed1 (64 B)
// 20 pow nonce (8 B)         21 positions (u64 each)        22 proof bytes              23 lde domain (N)
// 24 [constraints_ok u64]    25 FRI layer count (u64)       26 trace root | constraint root (64 B)   27 FRI roots (layers x 32)
long orc_prover_get(void* hv, int what, size_t arg, uint8_t* out, size_t cap) {
    auto h = (ProverHandle*)hv; Prover& p = *h->p;
    switch (what) {
        case 0: return put_cols(p.polys, out, cap);
        case 1: return put_cols(p.registers, out, cap);
        case 2: return put(p.trace_leaves.data(), p.trace_leaves.size() * 32, out, cap);
        case 3: return put(p.trace_tree.nodes.data(), p.trace_tree.nodes.size() * 32, out, cap);
        case 4: return put(p.ccoef.raw.data(), p.ccoef.raw.size() * 16, out, cap);
        case 5: return put(p.i_evaluations.data(), p.i_evaluations.size() * 16, out, cap);
        case 6: return put(p.f_evaluations.data(), p.f_evaluations.size() * 16, out, cap);
        case 7: return put(p.t_evaluations.data(), p.t_evaluations.size() * 16, out, cap);
        case 8: return put(p.constraint_poly.data(), p.constraint_poly.size() * 16, out, cap);
        case 9: return put(p.constraint_evaluations.data(), p.constraint_evaluations.size() * 16, out, cap);
        case 10: return put(p.constraint_tree.nodes.data(), p.constraint_tree.nodes.size() * 32, out, cap);
        case 11: return put(p.compcoef.raw.data(), p.compcoef.raw.size() * 16, out, cap);
        case 12: return put(p.trace_at_z1.data(), p.trace_at_z1.size() * 16, out, cap);
        case 13: return put(p.trace_at_z2.data(), p.trace_at_z2.size() * 16, out, cap);
        case 14: return put(p.composition_poly.data(), p.composition_poly.size() * 16, out, cap);
        case 15: return put(p.composed_evaluations.data(), p.composed_evaluations.size() * 16, out, cap);
        case 16: if (arg >= p.fri.values.size()) return -1; return put(p.fri.values[arg].data(), p.fri.values[arg].size() * 64, out, cap);
        case 17: if (arg >= p.fri.trees.size()) return -1; return put(p.fri.trees[arg].nodes.data(), p.fri.trees[arg].nodes.size() * 32, out, cap);
        case 18: return put(p.fri.special_xs.data(), p.fri.special_xs.size() * 16, out, cap);
        case 19: { uint8_t b[64]; memcpy(b, p.query_seed0.data(), 32); memcpy(b + 32, p.query_seed1.data(), 32); return put(b, 64, out, cap); }
        case 20: return put(&p.pow_nonce, 8, out, cap);
        case 21: { std::vector<uint64_t> v(p.positions.begin(), p.positions.end()); return put(v.data(), v.size() * 8, out, cap); }
        case 22: return put(h->proof_bytes.data(), h->proof_bytes.size(), out, cap);
        case 23: return put(p.lde_domain.data(), p.lde_domain.size() * 16, out, cap);
        case 24: { uint64_t v = p.constraints_ok ? 1 : 0; return put(&v, 8, out, cap); }
        case 25: { uint64_t v = p.fri.trees.size(); return put(&v, 8, out, cap); }
        case 26: { uint8_t b[64]; memset(b, 0, 64); if (!p.trace_tree.nodes.empty()) memcpy(b, p.trace_tree.root().data(), 32);
                   if (!p.constraint_tree.nodes.empty()) memcpy(b + 32, p.constraint_tree.root().data(), 32); return put(b, 64, out, cap); }
        case 27: { std::vector<uint8_t> b; for (auto& t : p.fri.trees) b.insert(b.end(), t.root().begin(), t.root().end()); return put(b.data(), b.size(), out, cap); }
    }
    return -1;
}

// standalone FRI commit phase over arbitrary evaluations (for the reference's fri prove/verify test, fri/mod.rs:39-95)
// returns 1 when fri::verify accepts; err receives the reference's error string otherwise
int orc_fri_prove_verify(const u128* evaluations, size_t domain_size, size_t claimed_degree, int drop_first_evaluation, char* err, size_t errcap) {
    try {
        vec ev(evaluations, evaluations + domain_size);
        vec domain = get_power_series(get_root_of_unity(domain_size), domain_size);
        ProofOptions o;
        FriReduction red = fri_reduce(ev, domain);
        auto positions = compute_query_positions(red.trees.back().root(), domain_size, o);
        FriProof proof = fri_build_proof(red, positions);
        vec sampled;
        for (size_t p : positions) sampled.push_back(ev[p]);
        if (drop_first_evaluation) sampled.erase(sampled.begin());
        VerifyResult r = fri_verify(proof, sampled, positions, claimed_degree, o);
        if (err && errcap) snprintf(err, errcap, "%s", r.error.c_str());
        return r.ok ? 1 : 0;
    } catch (const std::exception& e) { return fail(e); }
}

int orc_verify(const uint8_t* proof, size_t len, const uint8_t* program_hash, const u128* inputs, size_t nin, const u128* outputs, size_t nout, char* err, size_t errcap) {
    try {
        StarkProof p = deserialize_proof(proof, len);
        VerifyResult r = verify(program_hash, vec(inputs, inputs + nin), vec(outputs, outputs + nout), p);
        if (err && errcap) snprintf(err, errcap, "%s", r.error.c_str());
        return r.ok ? 1 : 0;
    } catch (const std::exception& e) { if (err && errcap) snprintf(err, errcap, "%s", e.what()); return fail(e); }
}

}  // extern "C"
