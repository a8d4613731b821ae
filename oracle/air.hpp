// ORACLE (test infrastructure, NOT product code) -- restatement of the reference's AIR: trace rows, op flags,
// decoder / stack transition constraints, boundary constraints and their random linear combination.
//
// Follows (all under /root/reference/src/stark/):
//   trace/trace_state.rs        TraceState :21-41, new :50, from_vec :73, op_code :165, update_from_trace :251, set_op_flags :281-350
//   constraints/utils.rs        is_binary :15, binary_not :20, are_equal :25, enforce_stack_copy :35, enforce_right_shift :44,
//                               enforce_left_shift :53, agg_constraint :73, extend_constants :87
//   constraints/decoder/mod.rs  degrees :31-47, Decoder::new :74, evaluate :129, evaluate_at :155, MASKS :219
//   constraints/decoder/op_bits.rs :10-79, decoder/sponge.rs :10-43, decoder/flow_ops.rs :10-165
//   constraints/stack/mod.rs    Stack::new :58, evaluate :86, evaluate_at :98, enforce_constraints :117-195
//   constraints/stack/{input,arithmetic,manipulation,comparison,conditional,hash}.rs
//   constraints/evaluator.rs    from_trace :35, from_proof :81, evaluate_transition :139, evaluate_transition_at :167,
//                               evaluate_boundaries :181-326, combine_transition_constraints :335, group_transition_constraints :385
//   utils/coefficients.rs       ConstraintCoefficients :62-77,108-185, CompositionCoefficients :80-104
// Reference quirks reproduced on purpose: ld_op_flags[2] uses cf_op_bits[1]; SWAP writes both
// constraints into slot 0; PUSH/ASSERT flag adjustments use the other bank's bit 0 and happen after BEGIN/NOOP flags.
#pragma once
#include "polynom.hpp"
#include "vm.hpp"
#include "prng.hpp"

namespace orc {

static const size_t MAX_CONSTRAINT_DEGREE = 8;                 // stark/mod.rs:25
static const size_t NUM_OP_CONSTRAINTS = 15, NUM_SPONGE_CONSTRAINTS = 4;
static const size_t NUM_STATIC_DECODER_CONSTRAINTS = NUM_OP_CONSTRAINTS + NUM_SPONGE_CONSTRAINTS + 1;   // decoder/mod.rs:53
static const size_t NUM_AUX_STACK_CONSTRAINTS = 2;            // stack/mod.rs:39

static inline u128 is_binary(u128 v) { return sub(mul(v, v), v); }
static inline u128 binary_not(u128 v) { return sub(1, v); }
static inline u128 are_equal(u128 a, u128 b) { return sub(a, b); }
static inline void agg(u128* r, size_t i, u128 flag, u128 value) { r[i] = add(r[i], mul(flag, value)); }

static inline void enforce_stack_copy(u128* r, size_t len, const u128* o, const u128* n, size_t from, u128 f) {
    for (size_t i = from; i < len; i++) agg(r, i, f, are_equal(o[i], n[i]));
}
static inline void enforce_right_shift(u128* r, size_t len, const u128* o, const u128* n, size_t num, u128 f) {
    for (size_t i = num; i < len; i++) agg(r, i, f, are_equal(o[i - num], n[i]));
}
static inline void enforce_left_shift(u128* r, size_t len, const u128* o, const u128* n, size_t from, size_t num, u128 f) {
    size_t start = from - num, rem = len - num;
    for (size_t i = start; i < rem; i++) agg(r, i, f, are_equal(o[i + num], n[i]));
    for (size_t i = rem; i < len; i++) agg(r, i, f, n[i]);
}

// ---- trace state ------------------------------------------------------------------------------------------
struct TraceState {
    u128 op_counter = 0, sponge[4] = {0, 0, 0, 0}, cf_bits[3] = {0, 0, 0}, ld_bits[5] = {0, 0, 0, 0, 0}, hd_bits[2] = {0, 0};
    vec ctx_stack, loop_stack, user_stack;
    size_t ctx_depth, loop_depth, stack_depth;
    u128 cf_flags[8], ld_flags[32], hd_flags[4], begin_flag = 0, noop_flag = 0;
    bool flags_set = false;

    TraceState(size_t ctx, size_t lp, size_t st) : ctx_depth(ctx), loop_depth(lp), stack_depth(st) {
        ctx_stack.assign(std::max(ctx, MIN_CONTEXT_DEPTH), 0);
        loop_stack.assign(std::max(lp, MIN_LOOP_DEPTH), 0);
        user_stack.assign(std::max(st, MIN_STACK_DEPTH), 0);
    }
    static TraceState from_vec(size_t ctx, size_t lp, size_t st, const vec& s) {
        TraceState t(ctx, lp, st);
        t.load_row([&](size_t j) { return s[j]; });
        return t;
    }
    template <class F> void load_row(F get) {
        op_counter = get(0);
        for (int i = 0; i < 4; i++) sponge[i] = get(1 + i);
        for (int i = 0; i < 3; i++) cf_bits[i] = get(5 + i);
        for (int i = 0; i < 5; i++) ld_bits[i] = get(8 + i);
        for (int i = 0; i < 2; i++) hd_bits[i] = get(13 + i);
        size_t c = 15;
        for (size_t i = 0; i < ctx_depth; i++) ctx_stack[i] = get(c + i);
        c += ctx_depth;
        for (size_t i = 0; i < loop_depth; i++) loop_stack[i] = get(c + i);
        c += loop_depth;
        for (size_t i = 0; i < stack_depth; i++) user_stack[i] = get(c + i);
        flags_set = false;
    }
    void update_from_trace(const std::vector<vec>& trace, size_t step) { load_row([&](size_t j) { return trace[j][step]; }); }
    size_t width() const { return 15 + ctx_depth + loop_depth + stack_depth; }
    vec to_vec() const {
        vec r{op_counter};
        r.insert(r.end(), sponge, sponge + 4); r.insert(r.end(), cf_bits, cf_bits + 3);
        r.insert(r.end(), ld_bits, ld_bits + 5); r.insert(r.end(), hd_bits, hd_bits + 2);
        r.insert(r.end(), ctx_stack.begin(), ctx_stack.begin() + ctx_depth);
        r.insert(r.end(), loop_stack.begin(), loop_stack.begin() + loop_depth);
        r.insert(r.end(), user_stack.begin(), user_stack.begin() + stack_depth);
        return r;
    }
    u128 op_code() const {                                      // trace_state.rs:165
        u128 r = ld_bits[0];
        r = add(r, mul(ld_bits[1], 2)); r = add(r, mul(ld_bits[2], 4)); r = add(r, mul(ld_bits[3], 8));
        r = add(r, mul(ld_bits[4], 16)); r = add(r, mul(hd_bits[0], 32)); r = add(r, mul(hd_bits[1], 64));
        return r;
    }
    void ensure_flags() { if (!flags_set) set_op_flags(); }
    void set_op_flags() {                                       // trace_state.rs:281-350
        u128 not0 = binary_not(cf_bits[0]), not1 = binary_not(cf_bits[1]);
        cf_flags[0] = mul(not0, not1); cf_flags[1] = mul(cf_bits[0], not1);
        cf_flags[2] = mul(not0, cf_bits[1]); cf_flags[3] = mul(cf_bits[0], cf_bits[1]);
        for (int i = 0; i < 4; i++) cf_flags[4 + i] = cf_flags[i];
        u128 not2 = binary_not(cf_bits[2]);
        for (int i = 0; i < 4; i++) cf_flags[i] = mul(cf_flags[i], not2);
        for (int i = 4; i < 8; i++) cf_flags[i] = mul(cf_flags[i], cf_bits[2]);

        not0 = binary_not(ld_bits[0]); not1 = binary_not(ld_bits[1]);
        ld_flags[0] = mul(not0, not1); ld_flags[1] = mul(ld_bits[0], not1);
        ld_flags[2] = mul(not0, cf_bits[1]);                    // sic: cf_op_bits[1] (trace_state.rs:301)
        ld_flags[3] = mul(ld_bits[0], ld_bits[1]);
        for (int i = 0; i < 4; i++) ld_flags[4 + i] = ld_flags[i];
        not2 = binary_not(ld_bits[2]);
        for (int i = 0; i < 4; i++) ld_flags[i] = mul(ld_flags[i], not2);
        for (int i = 4; i < 8; i++) ld_flags[i] = mul(ld_flags[i], ld_bits[2]);
        for (int i = 0; i < 8; i++) ld_flags[8 + i] = ld_flags[i];
        u128 not3 = binary_not(ld_bits[3]);
        for (int i = 0; i < 8; i++) ld_flags[i] = mul(ld_flags[i], not3);
        for (int i = 8; i < 16; i++) ld_flags[i] = mul(ld_flags[i], ld_bits[3]);
        for (int i = 0; i < 16; i++) ld_flags[16 + i] = ld_flags[i];
        u128 not4 = binary_not(ld_bits[4]);
        for (int i = 0; i < 16; i++) ld_flags[i] = mul(ld_flags[i], not4);
        for (int i = 16; i < 32; i++) ld_flags[i] = mul(ld_flags[i], ld_bits[4]);

        not0 = binary_not(hd_bits[0]); not1 = binary_not(hd_bits[1]);
        hd_flags[0] = mul(not0, not1); hd_flags[1] = mul(hd_bits[0], not1);
        hd_flags[2] = mul(not0, hd_bits[1]); hd_flags[3] = mul(hd_bits[0], hd_bits[1]);

        begin_flag = mul(ld_flags[ld_index(OP_BEGIN)], hd_flags[hd_index(OP_BEGIN)]);
        noop_flag = mul(ld_flags[ld_index(OP_NOOP)], hd_flags[hd_index(OP_NOOP)]);
        hd_flags[0] = mul(hd_flags[0], ld_bits[0]);             // PUSH adjustment (trace_state.rs:343)
        ld_flags[0] = mul(ld_flags[0], hd_bits[0]);             // ASSERT adjustment (trace_state.rs:346)
        flags_set = true;
    }
};

// ---- periodic constants ----------------------------------------------------------------------------------------
// interpolate each 16-entry cycle into a polynomial and evaluate it over 16*extension_factor points   (constraints/utils.rs:87)
static inline void extend_constants(const std::vector<vec>& constants, size_t extension_factor, std::vector<vec>& polys, std::vector<vec>& evaluations) {
    u128 root = get_root_of_unity(BASE_CYCLE_LENGTH);
    vec inv_tw = get_inv_twiddles(root, BASE_CYCLE_LENGTH);
    size_t domain_size = BASE_CYCLE_LENGTH * extension_factor;
    vec tw = get_twiddles(get_root_of_unity(domain_size), domain_size);
    polys.clear(); evaluations.clear();
    for (const vec& c : constants) {
        vec e = c;
        interpolate_fft_twiddles(e.data(), e.size(), inv_tw, true);
        polys.push_back(e);
        e.resize(domain_size, 0);
        eval_fft_twiddles(e.data(), e.size(), tw, true);
        evaluations.push_back(e);
    }
}

// ---- decoder constraints ------------------------------------------------------------------------------------------
static inline void enforce_op_bits(u128* result, TraceState& cur, TraceState& nxt, const u128* masks) {   // op_bits.rs:10
    cur.ensure_flags(); nxt.ensure_flags();
    size_t i = 0;
    u128 cf_bit_sum = 0;
    for (int k = 0; k < 3; k++) { result[i] = is_binary(cur.cf_bits[k]); cf_bit_sum = add(cf_bit_sum, cur.cf_bits[k]); i++; }
    u128 ld_bit_prod = 1;
    for (int k = 0; k < 5; k++) { result[i] = is_binary(cur.ld_bits[k]); ld_bit_prod = mul(ld_bit_prod, cur.ld_bits[k]); i++; }
    u128 hd_bit_prod = 1;
    for (int k = 0; k < 2; k++) { result[i] = is_binary(cur.hd_bits[k]); hd_bit_prod = mul(hd_bit_prod, cur.hd_bits[k]); i++; }

    u128 op_counter = cur.op_counter;
    u128 is_hacc = cur.cf_flags[F_HACC];
    u128 hacc_transition = mul(add(op_counter, 1), is_hacc);
    u128 rest_transition = mul(op_counter, binary_not(is_hacc));
    result[i] = are_equal(add(hacc_transition, rest_transition), nxt.op_counter); i++;
    result[i] = mul(op_counter, mul(binary_not(ld_bit_prod), binary_not(hd_bit_prod))); i++;
    result[i] = mul(cf_bit_sum, binary_not(mul(ld_bit_prod, hd_bit_prod))); i++;
    result[i] = mul(cur.cf_flags[F_VOID], binary_not(nxt.cf_flags[F_VOID])); i++;

    u128 prefix_mask = masks[1];
    agg(result, i, cur.cf_flags[F_BEGIN], prefix_mask); agg(result, i, cur.cf_flags[F_LOOP], prefix_mask);
    agg(result, i, cur.cf_flags[F_WRAP], prefix_mask);  agg(result, i, cur.cf_flags[F_BREAK], prefix_mask);
    u128 base_cycle_mask = masks[0];
    agg(result, i, cur.cf_flags[F_TEND], base_cycle_mask); agg(result, i, cur.cf_flags[F_FEND], base_cycle_mask);
    agg(result, i, cur.hd_flags[hd_index(OP_PUSH)], masks[2]);
}

static inline void enforce_hacc(u128* result, TraceState& cur, TraceState& nxt, const u128* ark, u128 op_flag) {   // decoder/sponge.rs:10
    cur.ensure_flags();
    u128 stack_top = nxt.user_stack[0];
    u128 push_flag = cur.hd_flags[hd_index(OP_PUSH)];
    u128 op_value = mul(stack_top, push_flag);
    u128 old_sponge[4], new_sponge[4];
    for (int i = 0; i < 4; i++) old_sponge[i] = add(cur.sponge[i], ark[i]);
    Rescue<4>::sbox(old_sponge);
    sponge_mds(old_sponge);
    old_sponge[0] = add(old_sponge[0], cur.op_code());
    old_sponge[1] = add(old_sponge[1], op_value);
    for (int i = 0; i < 4; i++) new_sponge[i] = nxt.sponge[i];
    sponge_inv_mds(new_sponge);
    Rescue<4>::sbox(new_sponge);
    for (int i = 0; i < 4; i++) new_sponge[i] = sub(new_sponge[i], ark[4 + i]);
    for (int i = 0; i < 4; i++) agg(result, i, op_flag, are_equal(old_sponge[i], new_sponge[i]));
}

// flow ops operate on result = evaluations[NUM_OP_CONSTRAINTS..]                                      flow_ops.rs:10-165
struct FlowCtx {
    u128* result; TraceState& cur; TraceState& nxt;
    size_t cl, ll;
    u128* ctx_result() { return result + SPONGE_WIDTH + 1; }
    u128* loop_result() { return result + SPONGE_WIDTH + 1 + cl; }
    FlowCtx(u128* r, TraceState& c, TraceState& n) : result(r), cur(c), nxt(n), cl(c.ctx_stack.size()), ll(c.loop_stack.size()) {}
    void sponge_cleared(u128 f) { for (int i = 0; i < 4; i++) agg(result, i, f, nxt.sponge[i]); }
    void begin(u128 f) {
        sponge_cleared(f);
        agg(ctx_result(), 0, f, are_equal(cur.sponge[0], nxt.ctx_stack[0]));
        enforce_right_shift(ctx_result(), cl, cur.ctx_stack.data(), nxt.ctx_stack.data(), 1, f);
        enforce_stack_copy(loop_result(), ll, cur.loop_stack.data(), nxt.loop_stack.data(), 0, f);
    }
    void tend(u128 f) {
        agg(result, 0, f, are_equal(cur.ctx_stack[0], nxt.sponge[0]));
        agg(result, 1, f, are_equal(cur.sponge[0], nxt.sponge[1]));
        agg(result, 3, f, nxt.sponge[3]);
        enforce_left_shift(ctx_result(), cl, cur.ctx_stack.data(), nxt.ctx_stack.data(), 1, 1, f);
        enforce_stack_copy(loop_result(), ll, cur.loop_stack.data(), nxt.loop_stack.data(), 0, f);
    }
    void fend(u128 f) {
        agg(result, 0, f, are_equal(cur.ctx_stack[0], nxt.sponge[0]));
        agg(result, 2, f, are_equal(cur.sponge[0], nxt.sponge[2]));
        agg(result, 3, f, nxt.sponge[3]);
        enforce_left_shift(ctx_result(), cl, cur.ctx_stack.data(), nxt.ctx_stack.data(), 1, 1, f);
        enforce_stack_copy(loop_result(), ll, cur.loop_stack.data(), nxt.loop_stack.data(), 0, f);
    }
    void loop(u128 f) {
        sponge_cleared(f);
        agg(ctx_result(), 0, f, are_equal(cur.sponge[0], nxt.ctx_stack[0]));
        enforce_right_shift(ctx_result(), cl, cur.ctx_stack.data(), nxt.ctx_stack.data(), 1, f);
        enforce_right_shift(loop_result(), ll, cur.loop_stack.data(), nxt.loop_stack.data(), 1, f);
    }
    void wrap(u128 f) {
        sponge_cleared(f);
        agg(result, SPONGE_WIDTH, f, are_equal(cur.sponge[0], cur.loop_stack[0]));
        enforce_stack_copy(ctx_result(), cl, cur.ctx_stack.data(), nxt.ctx_stack.data(), 0, f);
        enforce_stack_copy(loop_result(), ll, cur.loop_stack.data(), nxt.loop_stack.data(), 0, f);
    }
    void brk(u128 f) {
        for (int i = 0; i < 4; i++) agg(result, i, f, are_equal(cur.sponge[i], nxt.sponge[i]));
        agg(result, SPONGE_WIDTH, f, are_equal(cur.sponge[0], cur.loop_stack[0]));
        enforce_stack_copy(ctx_result(), cl, cur.ctx_stack.data(), nxt.ctx_stack.data(), 0, f);
        enforce_left_shift(loop_result(), ll, cur.loop_stack.data(), nxt.loop_stack.data(), 1, 1, f);
    }
    void vd(u128 f) {
        for (int i = 0; i < 4; i++) agg(result, i, f, are_equal(cur.sponge[i], nxt.sponge[i]));
        enforce_stack_copy(ctx_result(), cl, cur.ctx_stack.data(), nxt.ctx_stack.data(), 0, f);
        enforce_stack_copy(loop_result(), ll, cur.loop_stack.data(), nxt.loop_stack.data(), 0, f);
    }
};

static const uint8_t CYCLE_MASKS[3][16] = {                     // decoder/mod.rs:219
    {0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1},
    {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0},
    {0, 1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1},
};

struct DecoderAir {
    size_t ctx_depth, loop_depth, trace_length, cycle_length;
    std::vector<vec> ark_polys, ark_evals, mask_polys, mask_evals;   // [8][..], [3][..]
    std::vector<size_t> degrees;
    DecoderAir(size_t tl, size_t ext, size_t ctx, size_t lp) : ctx_depth(ctx), loop_depth(lp), trace_length(tl) {   // decoder/mod.rs:74
        static const size_t op_deg[15] = {2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 8, 8, 6, 4};
        degrees.assign(op_deg, op_deg + 15);
        for (size_t d : {6, 7, 6, 6}) degrees.push_back(d);
        degrees.push_back(4);
        degrees.resize(degrees.size() + std::max(ctx, MIN_CONTEXT_DEPTH) + std::max(lp, MIN_LOOP_DEPTH), 4);
        cycle_length = BASE_CYCLE_LENGTH * ext;
        std::vector<vec> ark(8, vec(16)), masks(3, vec(16));
        for (int r = 0; r < 8; r++) for (int c = 0; c < 16; c++) ark[r][c] = cst(SPONGE_ARK[r][c]);
        for (int r = 0; r < 3; r++) for (int c = 0; c < 16; c++) masks[r][c] = CYCLE_MASKS[r][c];
        extend_constants(ark, ext, ark_polys, ark_evals);
        extend_constants(masks, ext, mask_polys, mask_evals);
    }
    size_t constraint_count() const { return degrees.size(); }
    void run(TraceState& cur, TraceState& nxt, const u128* ark, const u128* masks, u128* result) {
        enforce_op_bits(result, cur, nxt, masks);
        cur.ensure_flags();
        FlowCtx fc(result + NUM_OP_CONSTRAINTS, cur, nxt);
        enforce_hacc(fc.result, cur, nxt, ark, cur.cf_flags[F_HACC]);
        fc.begin(cur.cf_flags[F_BEGIN]); fc.tend(cur.cf_flags[F_TEND]); fc.fend(cur.cf_flags[F_FEND]);
        fc.loop(cur.cf_flags[F_LOOP]); fc.wrap(cur.cf_flags[F_WRAP]); fc.brk(cur.cf_flags[F_BREAK]);
        fc.vd(cur.cf_flags[F_VOID]);
    }
    void evaluate(TraceState& cur, TraceState& nxt, size_t step, u128* result) {               // decoder/mod.rs:129
        u128 ark[8], masks[3];
        size_t s = step % cycle_length;
        for (int i = 0; i < 8; i++) ark[i] = ark_evals[i][s];
        for (int i = 0; i < 3; i++) masks[i] = mask_evals[i][s];
        run(cur, nxt, ark, masks, result);
    }
    void evaluate_at(TraceState& cur, TraceState& nxt, u128 x, u128* result) {                 // decoder/mod.rs:155
        u128 xc = exp(x, (u128)(trace_length / BASE_CYCLE_LENGTH));
        u128 ark[8], masks[3];
        for (int i = 0; i < 8; i++) ark[i] = poly_eval(ark_polys[i], xc);
        for (int i = 0; i < 3; i++) masks[i] = poly_eval(mask_polys[i], xc);
        run(cur, nxt, ark, masks, result);
    }
};

// ---- stack constraints ------------------------------------------------------------------------------------------------
static inline void enforce_stack_constraints(TraceState& cur, TraceState& nxt, const u128* ark, u128* result, size_t result_len) {   // stack/mod.rs:117
    cur.ensure_flags();
    u128* aux = result;
    u128* out = result + NUM_AUX_STACK_CONSTRAINTS;
    size_t out_len = result_len - NUM_AUX_STACK_CONSTRAINTS;
    const u128* o = cur.user_stack.data();
    const u128* n = nxt.user_stack.data();
    size_t L = cur.user_stack.size();
    vec ev(L, 0);
    u128* e = ev.data();
    const u128* ld = cur.ld_flags;
    u128 f;

    // assertions (comparison.rs:24-38)
    f = ld[ld_index(OP_ASSERT)];   enforce_left_shift(e, L, o, n, 1, 1, f); agg(aux, 0, f, are_equal(1, o[0]));
    f = ld[ld_index(OP_ASSERTEQ)]; enforce_left_shift(e, L, o, n, 2, 2, f); agg(aux, 0, f, are_equal(o[0], o[1]));
    // input (input.rs:6-22)
    enforce_right_shift(e, L, o, n, 1, ld[ld_index(OP_READ)]);
    enforce_right_shift(e, L, o, n, 2, ld[ld_index(OP_READ2)]);
    // manipulation (manipulation.rs:12-117)
    f = ld[ld_index(OP_DUP)];  agg(e, 0, f, are_equal(n[0], o[0])); enforce_right_shift(e, L, o, n, 1, f);
    f = ld[ld_index(OP_DUP2)]; agg(e, 0, f, are_equal(n[0], o[0])); agg(e, 1, f, are_equal(n[1], o[1])); enforce_right_shift(e, L, o, n, 2, f);
    f = ld[ld_index(OP_DUP4)]; for (int i = 0; i < 4; i++) agg(e, i, f, are_equal(n[i], o[i])); enforce_right_shift(e, L, o, n, 4, f);
    f = ld[ld_index(OP_PAD2)]; agg(e, 0, f, n[0]); agg(e, 1, f, n[1]); enforce_right_shift(e, L, o, n, 2, f);
    enforce_left_shift(e, L, o, n, 1, 1, ld[ld_index(OP_DROP)]);
    enforce_left_shift(e, L, o, n, 4, 4, ld[ld_index(OP_DROP4)]);
    f = ld[ld_index(OP_SWAP)];                                    // sic: both into slot 0 (manipulation.rs:63-64)
    agg(e, 0, f, are_equal(n[0], o[1])); agg(e, 0, f, are_equal(n[1], o[0])); enforce_stack_copy(e, L, o, n, 2, f);
    f = ld[ld_index(OP_SWAP2)];
    agg(e, 0, f, are_equal(n[0], o[2])); agg(e, 1, f, are_equal(n[1], o[3])); agg(e, 2, f, are_equal(n[2], o[0])); agg(e, 3, f, are_equal(n[3], o[1]));
    enforce_stack_copy(e, L, o, n, 4, f);
    f = ld[ld_index(OP_SWAP4)];
    for (int i = 0; i < 4; i++) agg(e, i, f, are_equal(n[i], o[4 + i]));
    for (int i = 0; i < 4; i++) agg(e, 4 + i, f, are_equal(n[4 + i], o[i]));
    enforce_stack_copy(e, L, o, n, 8, f);
    f = ld[ld_index(OP_ROLL4)];
    agg(e, 0, f, are_equal(n[0], o[3])); for (int i = 1; i < 4; i++) agg(e, i, f, are_equal(n[i], o[i - 1]));
    enforce_stack_copy(e, L, o, n, 4, f);
    f = ld[ld_index(OP_ROLL8)];
    agg(e, 0, f, are_equal(n[0], o[7])); for (int i = 1; i < 8; i++) agg(e, i, f, are_equal(n[i], o[i - 1]));
    enforce_stack_copy(e, L, o, n, 8, f);
    // arithmetic and boolean (arithmetic.rs:12-118)
    f = ld[ld_index(OP_ADD)]; agg(e, 0, f, are_equal(n[0], add(o[0], o[1]))); enforce_left_shift(e, L, o, n, 2, 1, f);
    f = ld[ld_index(OP_MUL)]; agg(e, 0, f, are_equal(n[0], mul(o[0], o[1]))); enforce_left_shift(e, L, o, n, 2, 1, f);
    f = ld[ld_index(OP_INV)]; agg(e, 0, f, are_equal(1, mul(n[0], o[0]))); enforce_stack_copy(e, L, o, n, 1, f);
    f = ld[ld_index(OP_NEG)]; agg(e, 0, f, add(n[0], o[0])); enforce_stack_copy(e, L, o, n, 1, f);
    f = ld[ld_index(OP_NOT)]; agg(e, 0, f, are_equal(n[0], binary_not(o[0]))); enforce_stack_copy(e, L, o, n, 1, f); agg(aux, 0, f, is_binary(o[0]));
    f = ld[ld_index(OP_AND)]; agg(e, 0, f, are_equal(n[0], mul(o[0], o[1]))); enforce_left_shift(e, L, o, n, 2, 1, f);
    agg(aux, 0, f, is_binary(o[0])); agg(aux, 1, f, is_binary(o[1]));
    f = ld[ld_index(OP_OR)]; agg(e, 0, f, are_equal(n[0], binary_not(mul(binary_not(o[0]), binary_not(o[1]))))); enforce_left_shift(e, L, o, n, 2, 1, f);
    agg(aux, 0, f, is_binary(o[0])); agg(aux, 1, f, is_binary(o[1]));
    // comparison (comparison.rs:45-133)
    {
        f = ld[ld_index(OP_EQ)];
        u128 diff = sub(o[1], o[2]);
        agg(e, 0, f, are_equal(n[0], binary_not(mul(diff, o[0]))));
        enforce_left_shift(e, L, o, n, 3, 2, f);
        agg(aux, 0, f, mul(n[0], diff));
    }
    {
        f = ld[ld_index(OP_BINACC)];
        u128 bit = n[0], pw = o[2];
        agg(e, 0, f, is_binary(bit)); agg(e, 1, f, n[1]);
        agg(e, 2, f, are_equal(n[2], mul(pw, 2)));
        agg(e, 3, f, are_equal(n[3], add(o[3], mul(bit, pw))));
        enforce_stack_copy(e, L, o, n, 4, f);
    }
    // conditional (conditional.rs:12-82)
    {
        f = ld[ld_index(OP_CHOOSE)];
        u128 c = o[2], nc = binary_not(c);
        agg(e, 0, f, are_equal(n[0], add(mul(c, o[0]), mul(nc, o[1]))));
        enforce_left_shift(e, L, o, n, 3, 2, f);
        agg(aux, 0, f, is_binary(c));
    }
    {
        f = ld[ld_index(OP_CHOOSE2)];
        u128 c = o[4], nc = binary_not(c);
        agg(e, 0, f, are_equal(n[0], add(mul(c, o[0]), mul(nc, o[2]))));
        agg(e, 1, f, are_equal(n[1], add(mul(c, o[1]), mul(nc, o[3]))));
        enforce_left_shift(e, L, o, n, 6, 4, f);
        agg(aux, 0, f, is_binary(c));
    }
    {
        f = ld[ld_index(OP_CSWAP2)];
        u128 c = o[4], nc = binary_not(c);
        agg(e, 0, f, are_equal(n[0], add(mul(c, o[2]), mul(nc, o[0]))));
        agg(e, 1, f, are_equal(n[1], add(mul(c, o[3]), mul(nc, o[1]))));
        agg(e, 2, f, are_equal(n[2], add(mul(c, o[0]), mul(nc, o[2]))));
        agg(e, 3, f, are_equal(n[3], add(mul(c, o[1]), mul(nc, o[3]))));
        enforce_left_shift(e, L, o, n, 6, 2, f);
        agg(aux, 0, f, is_binary(c));
    }
    // high-degree operations
    const u128* hd = cur.hd_flags;
    enforce_right_shift(e, L, o, n, 1, hd[hd_index(OP_PUSH)]);                     // input.rs:6
    {
        f = hd[hd_index(OP_CMP)];                                                // comparison.rs:64
        u128 x_bit = n[1], y_bit = n[2];
        agg(e, 0, f, is_binary(x_bit)); agg(e, 1, f, is_binary(y_bit));
        u128 not_set = n[3];
        u128 bit_gt = mul(x_bit, binary_not(y_bit)), bit_lt = mul(y_bit, binary_not(x_bit));
        agg(e, 2, f, are_equal(n[4], add(o[4], mul(bit_gt, not_set))));
        agg(e, 3, f, are_equal(n[5], add(o[5], mul(bit_lt, not_set))));
        u128 pw = o[0];
        u128 x_acc = add(o[7], mul(x_bit, pw)), y_acc = add(o[6], mul(y_bit, pw));
        agg(e, 4, f, are_equal(n[6], y_acc)); agg(e, 5, f, are_equal(n[7], x_acc));
        agg(e, 6, f, are_equal(not_set, mul(binary_not(o[5]), binary_not(o[4]))));
        agg(e, 7, f, are_equal(mul(n[0], 2), pw));
        enforce_stack_copy(e, L, o, n, 8, f);
    }
    {
        f = hd[hd_index(OP_RESCR)];                                              // hash.rs:9
        u128 os[6], ns[6];
        for (int i = 0; i < 6; i++) os[i] = add(o[i], ark[i]);
        Rescue<6>::sbox(os); hasher_mds(os);
        for (int i = 0; i < 6; i++) ns[i] = n[i];
        hasher_inv_mds(ns); Rescue<6>::sbox(ns);
        for (int i = 0; i < 6; i++) ns[i] = sub(ns[i], ark[6 + i]);
        for (int i = 0; i < 6; i++) agg(e, i, f, are_equal(ns[i], os[i]));
        enforce_stack_copy(e, L, o, n, 6, f);
    }
    // composite operations
    enforce_stack_copy(e, L, o, n, 0, cur.begin_flag);
    enforce_stack_copy(e, L, o, n, 0, cur.noop_flag);
    for (size_t i = 0; i < out_len; i++) out[i] = e[i];
}

struct StackAir {
    size_t trace_length, cycle_length;
    std::vector<vec> ark_polys, ark_evals;   // [12][..]
    std::vector<size_t> degrees;
    StackAir(size_t tl, size_t ext, size_t stack_depth) : trace_length(tl) {      // stack/mod.rs:58
        degrees.assign(2, 7);
        degrees.resize(stack_depth + NUM_AUX_STACK_CONSTRAINTS, 7);
        cycle_length = BASE_CYCLE_LENGTH * ext;
        std::vector<vec> ark(12, vec(16));
        for (int r = 0; r < 12; r++) for (int c = 0; c < 16; c++) ark[r][c] = cst(HASHER_ARK[r][c]);
        extend_constants(ark, ext, ark_polys, ark_evals);
    }
    void evaluate(TraceState& cur, TraceState& nxt, size_t step, u128* result) {   // stack/mod.rs:86
        u128 ark[12];
        size_t s = step % cycle_length;
        for (int i = 0; i < 12; i++) ark[i] = ark_evals[i][s];
        enforce_stack_constraints(cur, nxt, ark, result, degrees.size());
    }
    void evaluate_at(TraceState& cur, TraceState& nxt, u128 x, u128* result) {     // stack/mod.rs:98
        u128 xc = exp(x, (u128)(trace_length / BASE_CYCLE_LENGTH));
        u128 ark[12];
        for (int i = 0; i < 12; i++) ark[i] = poly_eval(ark_polys[i], xc);
        enforce_stack_constraints(cur, nxt, ark, result, degrees.size());
    }
};

// ---- coefficients (utils/coefficients.rs) --------------------------------------------------------------------------------
static const size_t NUM_OP_BITS = 10;
static const size_t NUM_BOUNDARY_CONSTRAINTS = 1 + SPONGE_WIDTH + NUM_OP_BITS + MAX_CONTEXT_DEPTH + MAX_LOOP_DEPTH + MAX_PUBLIC_INPUTS;   // 47
static const size_t NUM_TRANSITION_CONSTRAINTS = NUM_STATIC_DECODER_CONSTRAINTS + MAX_CONTEXT_DEPTH + MAX_LOOP_DEPTH + MAX_STACK_DEPTH + NUM_AUX_STACK_CONSTRAINTS;  // 78
static const size_t NUM_CONSTRAINTS = NUM_TRANSITION_CONSTRAINTS + 2 * NUM_BOUNDARY_CONSTRAINTS;   // 172

struct BoundaryCoefficients { u128 op_counter[2], sponge[8], op_bits[20], ctx_stack[32], loop_stack[16], user_stack[16]; };

static inline size_t build_boundary_coefficients(const u128* c, BoundaryCoefficients& b) {    // coefficients.rs:108
    size_t p = 0;
    for (int i = 0; i < 2; i++) b.op_counter[i] = c[p++];
    for (int i = 0; i < 8; i++) b.sponge[i] = c[p++];
    for (int i = 0; i < 20; i++) b.op_bits[i] = c[p++];
    for (int i = 0; i < 32; i++) b.ctx_stack[i] = c[p++];
    for (int i = 0; i < 16; i++) b.loop_stack[i] = c[p++];
    for (int i = 0; i < 16; i++) b.user_stack[i] = c[p++];
    return p;
}
static inline vec build_transition_coefficients(const u128* c, size_t ctx, size_t lp, size_t st) {   // coefficients.rs:140
    ctx = std::max(ctx, MIN_CONTEXT_DEPTH); lp = std::max(lp, MIN_LOOP_DEPTH); st = std::max(st, MIN_STACK_DEPTH);
    vec r;
    size_t s = 0;
    auto take = [&](size_t from, size_t cnt) { for (size_t i = 0; i < cnt; i++) r.push_back(c[from + i]); };
    take(s, NUM_STATIC_DECODER_CONSTRAINTS * 2); s += NUM_STATIC_DECODER_CONSTRAINTS * 2;
    take(s, ctx * 2); s += MAX_CONTEXT_DEPTH * 2;
    take(s, lp * 2); s += MAX_LOOP_DEPTH * 2;
    take(s, NUM_AUX_STACK_CONSTRAINTS * 2); s += NUM_AUX_STACK_CONSTRAINTS * 2;
    take(s, st * 2);
    return r;
}
struct ConstraintCoefficients {
    BoundaryCoefficients i_boundary, f_boundary;
    vec transition;
    vec raw;                                                    // the 344 draws, kept for the C-ABI parity tests
    ConstraintCoefficients() {}
    ConstraintCoefficients(const uint8_t seed[32], size_t ctx, size_t lp, size_t st) { init(prng_vector(seed, 2 * NUM_CONSTRAINTS), ctx, lp, st); }
    void init(const vec& coefficients, size_t ctx, size_t lp, size_t st) {      // coefficients.rs:63
        raw = coefficients;
        size_t i = build_boundary_coefficients(raw.data(), i_boundary);
        i += build_boundary_coefficients(raw.data() + i, f_boundary);
        transition = build_transition_coefficients(raw.data() + i, ctx, lp, st);
    }
};
struct CompositionCoefficients {                                // coefficients.rs:52,80
    u128 trace1[2 * MAX_REGISTER_COUNT], trace2[2 * MAX_REGISTER_COUNT], t1_degree, t2_degree, constraints;
    vec raw;
    CompositionCoefficients() {}
    explicit CompositionCoefficients(const uint8_t seed[32]) { init(prng_vector(seed, 1 + 4 * MAX_REGISTER_COUNT + 3)); }
    void init(const vec& c) {
        raw = c;
        size_t p = 1;
        for (size_t i = 0; i < 2 * MAX_REGISTER_COUNT; i++) trace1[i] = c[p++];
        for (size_t i = 0; i < 2 * MAX_REGISTER_COUNT; i++) trace2[i] = c[p++];
        t1_degree = c[p]; t2_degree = c[p + 1]; constraints = c[p + 2];
    }
};

// ---- evaluator -----------------------------------------------------------------------------------------------------------
struct Evaluator {
    DecoderAir decoder;
    StackAir stack;
    ConstraintCoefficients coefficients;
    size_t domain_size, extension_factor, t_constraint_num;
    std::vector<std::pair<u128, std::vector<size_t>>> t_degree_groups;
    vec program_hash, inputs, outputs;
    u128 op_count, b_degree_adj;

    Evaluator(size_t trace_length, size_t ext, size_t ctx, size_t lp, size_t st, size_t domain,
              const ConstraintCoefficients& cc, const vec& prog_hash, u128 opc, const vec& in, const vec& out)
        : decoder(trace_length, ext, ctx, lp), stack(trace_length, ext, st), coefficients(cc), domain_size(domain),
          extension_factor(ext), program_hash(prog_hash), inputs(in), outputs(out), op_count(opc) {
        std::vector<size_t> degrees = decoder.degrees;
        degrees.insert(degrees.end(), stack.degrees.begin(), stack.degrees.end());
        t_constraint_num = degrees.size();
        // group_transition_constraints (evaluator.rs:385)
        std::vector<std::vector<size_t>> groups(9);
        for (size_t i = 0; i < degrees.size(); i++) groups[degrees[i]].push_back(i);
        size_t target = (MAX_CONSTRAINT_DEGREE - 1) * trace_length + (trace_length - 1);        // evaluator.rs:426
        for (size_t d = 0; d < groups.size(); d++) {
            if (groups[d].empty()) continue;
            size_t constraint_degree = (trace_length - 1) * d;
            t_degree_groups.push_back({(u128)(target - constraint_degree), groups[d]});
        }
        size_t b_target = (MAX_CONSTRAINT_DEGREE - 1) * trace_length + 1;                        // evaluator.rs:417
        b_degree_adj = (u128)(b_target - (trace_length - 1));
    }
    size_t trace_length() const { return domain_size / extension_factor; }
    u128 get_x_at_last_step() const { return exp(get_root_of_unity(trace_length()), (u128)(trace_length() - 1)); }   // evaluator.rs:128
    bool should_evaluate_to_zero_at(size_t step) const { return (step & (extension_factor - 1)) == 0 && step != domain_size - extension_factor; }

    u128 combine_transition_constraints(const vec& ev, u128 x) const {                            // evaluator.rs:335
        const vec& cc = coefficients.transition;
        u128 result = 0;
        size_t i = 0;
        for (auto& g : t_degree_groups) {
            u128 result_adj = 0;
            for (size_t idx : g.second) {
                result = add(result, mul(ev[idx], cc[i * 2]));
                result_adj = add(result_adj, mul(ev[idx], cc[i * 2 + 1]));
                i++;
            }
            result = add(result, mul(result_adj, exp(x, g.first)));
        }
        return result;
    }
    // returns false through `ok` when a transition constraint does not vanish on a trace row (panic at evaluator.rs:155)
    u128 evaluate_transition(TraceState& cur, TraceState& nxt, u128 x, size_t step, bool* ok, vec* raw = nullptr) {   // evaluator.rs:139
        vec ev(t_constraint_num, 0);
        decoder.evaluate(cur, nxt, step, ev.data());
        stack.evaluate(cur, nxt, step, ev.data() + decoder.constraint_count());
        if (raw) *raw = ev;
        if (should_evaluate_to_zero_at(step)) {
            for (u128 v : ev) if (v != 0) { if (ok) *ok = false; }
            return 0;
        }
        return combine_transition_constraints(ev, x);
    }
    u128 evaluate_transition_at(TraceState& cur, TraceState& nxt, u128 x) {                        // evaluator.rs:167
        vec ev(t_constraint_num, 0);
        decoder.evaluate_at(cur, nxt, x, ev.data());
        stack.evaluate_at(cur, nxt, x, ev.data() + decoder.constraint_count());
        return combine_transition_constraints(ev, x);
    }
    void evaluate_boundaries(const TraceState& cur, u128 x, u128& i_out, u128& f_out) const {     // evaluator.rs:181
        u128 xp = exp(x, b_degree_adj);
        for (int pass = 0; pass < 2; pass++) {
            const BoundaryCoefficients& cc = pass == 0 ? coefficients.i_boundary : coefficients.f_boundary;
            u128 res = 0, adj = 0;
            auto term = [&](u128 val, u128 c0, u128 c1) { res = add(res, mul(val, c0)); adj = add(adj, mul(val, c1)); };
            u128 one_or_zero = pass == 0 ? 0 : 1;      // op bits must be all 0 (BEGIN/HACC) at the start, all 1 (VOID/NOOP) at the end
            term(pass == 0 ? cur.op_counter : sub(cur.op_counter, op_count), cc.op_counter[0], cc.op_counter[1]);
            if (pass == 0) { for (int i = 0; i < 4; i++) term(cur.sponge[i], cc.sponge[i * 2], cc.sponge[i * 2 + 1]); }
            else { for (size_t i = 0; i < program_hash.size(); i++) term(sub(cur.sponge[i], program_hash[i]), cc.sponge[i * 2], cc.sponge[i * 2 + 1]); }
            size_t k = 0;
            for (int i = 0; i < 3; i++, k += 2) term(sub(cur.cf_bits[i], one_or_zero), cc.op_bits[k], cc.op_bits[k + 1]);
            for (int i = 0; i < 5; i++, k += 2) term(sub(cur.ld_bits[i], one_or_zero), cc.op_bits[k], cc.op_bits[k + 1]);
            for (int i = 0; i < 2; i++, k += 2) term(sub(cur.hd_bits[i], one_or_zero), cc.op_bits[k], cc.op_bits[k + 1]);
            for (size_t i = 0; i < cur.ctx_stack.size(); i++) term(cur.ctx_stack[i], cc.ctx_stack[i * 2], cc.ctx_stack[i * 2 + 1]);
            for (size_t i = 0; i < cur.loop_stack.size(); i++) term(cur.loop_stack[i], cc.loop_stack[i * 2], cc.loop_stack[i * 2 + 1]);
            const vec& io = pass == 0 ? inputs : outputs;
            for (size_t i = 0; i < io.size(); i++) term(sub(cur.user_stack[i], io[i]), cc.user_stack[i * 2], cc.user_stack[i * 2 + 1]);
            res = add(res, mul(adj, xp));
            (pass == 0 ? i_out : f_out) = res;
        }
    }
};

}  // namespace orc
