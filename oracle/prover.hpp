// ORACLE (test infrastructure, NOT product code) -- single-threaded CPU restatement of the reference STARK prover.
//
// Follows (under /root/reference/src/stark/):
//   prover.rs                 prove :17-168 (9 steps), twiddles_from_domain :173, evaluations_to_leaves :180, build_composition_poly :189
//   trace/trace_table.rs      extend :143, build_merkle_tree :174, eval_polys_at :189, get_composition_poly :206, get_register_values_at :127
//   constraints/constraint_table.rs   evaluate :45, combine_polys :54-88
//   constraints/constraint_poly.rs    eval :28, merge_into :39
//   fri/prover.rs             reduce :11-53, build_proof :55-95;  fri/utils.rs get_augmented_positions :4, hash_values :16
//   utils/mod.rs              get_composition_degree :13, get_incremental_trace_degree :20, compute_query_positions :25, map_trace_to_constraint_positions :46
//   utils/proof_of_work.rs    find_pow_nonce :4-32
//   options.rs :16-91, proof.rs :11-77
// The proof wire format is bincode's default configuration (little-endian fixed-width integers, u64 length prefixes; the call
// site is /root/reference/src/main.rs:44); bincode itself is a third-party crate that is not under /root/reference.
#pragma once
#include "air.hpp"
#include "merkle.hpp"
#include <functional>

namespace orc {

struct ProofOptions {                                           // options.rs:16
    size_t extension_factor = 32, num_queries = 50;
    uint32_t grinding_factor = 20;
};

struct FriLayer { hash32 root; std::vector<quad> values; std::vector<std::vector<hash32>> nodes; uint8_t depth; };   // fri/mod.rs:25
struct FriProof { std::vector<FriLayer> layers; hash32 rem_root; vec rem_values; };                                  // fri/mod.rs:18

struct StarkProof {                                             // proof.rs:11-37
    hash32 trace_root;
    uint8_t domain_depth = 0, ctx_depth = 0, loop_depth = 0, stack_depth = 0;
    uint32_t op_count = 0;
    std::vector<std::vector<hash32>> trace_nodes;
    std::vector<vec> trace_evaluations;
    hash32 constraint_root;
    BatchMerkleProof constraint_proof;
    vec trace_at_z1, trace_at_z2;
    FriProof degree_proof;
    uint64_t pow_nonce = 0;
    ProofOptions options;

    size_t domain_size() const { return (size_t)1 << domain_depth; }
    size_t trace_length() const { return domain_size() / options.extension_factor; }
};

static inline size_t get_composition_degree(size_t trace_length) { return (MAX_CONSTRAINT_DEGREE - 1) * trace_length - 1; }       // utils/mod.rs:13
static inline size_t get_incremental_trace_degree(size_t trace_length) { return get_composition_degree(trace_length) - (trace_length - 2); }   // :20

static inline std::vector<size_t> compute_query_positions(const hash32& seed, size_t domain_size, const ProofOptions& o) {   // utils/mod.rs:25
    ChaCha20Rng g(seed.data());
    std::vector<size_t> result;
    for (int it = 0; it < 1000; it++) {
        size_t value = (size_t)uniform_u64(g, domain_size);
        if (value % o.extension_factor == 0) continue;
        if (std::find(result.begin(), result.end(), value) != result.end()) continue;
        result.push_back(value);
        if (result.size() >= o.num_queries) break;
    }
    if (result.size() < o.num_queries) throw std::runtime_error("could not generate enough query positions");
    return result;
}
static inline std::vector<size_t> map_trace_to_constraint_positions(const std::vector<size_t>& positions) {   // utils/mod.rs:46
    std::vector<size_t> r;
    for (size_t p : positions) { size_t cp = p / 2; if (std::find(r.begin(), r.end(), cp) == r.end()) r.push_back(cp); }
    return r;
}
static inline std::vector<size_t> get_augmented_positions(const std::vector<size_t>& positions, size_t column_length) {   // fri/utils.rs:4
    size_t row_length = column_length / 4;
    std::vector<size_t> r;
    for (size_t p : positions) { size_t ap = p % row_length; if (std::find(r.begin(), r.end(), ap) == r.end()) r.push_back(ap); }
    return r;
}

static inline bool pow_check(const hash32& digest, uint32_t grinding) {
    uint64_t w = 0;
    for (int i = 7; i >= 0; i--) w = (w << 8) | digest[i];
    uint32_t tz = w == 0 ? 64 : (uint32_t)__builtin_ctzll(w);
    return tz >= grinding;
}
static inline hash32 pow_hash(const hash32& seed, uint64_t nonce) {
    uint8_t buf[64];
    memset(buf, 0, 64);
    memcpy(buf, seed.data(), 32);
    for (int i = 0; i < 8; i++) buf[32 + i] = (uint8_t)(nonce >> (8 * i));
    return hash_bytes(buf, 64);
}
static inline std::pair<hash32, uint64_t> find_pow_nonce(const hash32& seed, const ProofOptions& o) {   // proof_of_work.rs:4
    uint64_t nonce = 0;
    for (;;) {
        nonce += 1;
        hash32 d = pow_hash(seed, nonce);
        if (pow_check(d, o.grinding_factor)) return {d, nonce};
    }
}

static inline std::vector<hash32> fri_hash_values(const std::vector<quad>& values) {   // fri/utils.rs:16
    std::vector<hash32> r(values.size());
    for (size_t i = 0; i < values.size(); i++) {
        uint8_t buf[64];
        for (int k = 0; k < 4; k++) to_bytes(values[i][k], buf + 16 * k);
        r[i] = hash_bytes(buf, 64);
    }
    return r;
}

struct FriReduction { std::vector<MerkleTree> trees; std::vector<std::vector<quad>> values; vec special_xs; };

static inline FriReduction fri_reduce(const vec& evaluations, const vec& domain) {   // fri/prover.rs:11 (MAX_REMAINDER_LENGTH = 256, fri/mod.rs:13)
    FriReduction out;
    std::vector<quad> p_values = quartic_transpose(evaluations, 1);
    MerkleTree p_tree(fri_hash_values(p_values));
    while (p_tree.values.size() * 4 > 256) {
        size_t depth = out.trees.size();
        size_t stride = (size_t)1 << (2 * depth);
        std::vector<quad> xs = quartic_transpose(domain, stride);
        std::vector<quad> polys = quartic_interpolate_batch(xs, p_values);
        u128 special_x = prng(p_tree.root().data());
        out.special_xs.push_back(special_x);
        vec column = quartic_evaluate_batch(polys, special_x);
        std::vector<quad> c_values = quartic_transpose(column, 1);
        MerkleTree c_tree(fri_hash_values(c_values));
        out.trees.push_back(std::move(p_tree));
        out.values.push_back(std::move(p_values));
        p_tree = std::move(c_tree);
        p_values = std::move(c_values);
    }
    out.trees.push_back(std::move(p_tree));
    out.values.push_back(std::move(p_values));
    return out;
}

static inline FriProof fri_build_proof(const FriReduction& red, const std::vector<size_t>& positions_in) {   // fri/prover.rs:55
    std::vector<size_t> positions = positions_in;
    size_t domain_size = red.trees[0].values.size() * 4;
    FriProof fp;
    for (size_t i = 0; i + 1 < red.trees.size(); i++) {
        positions = get_augmented_positions(positions, domain_size);
        BatchMerkleProof pr = red.trees[i].prove_batch(positions);
        FriLayer layer;
        layer.root = red.trees[i].root();
        for (size_t p : positions) layer.values.push_back(red.values[i][p]);
        layer.nodes = pr.nodes;
        layer.depth = pr.depth;
        fp.layers.push_back(layer);
        domain_size /= 4;
    }
    const auto& last_values = red.values.back();
    size_t n = last_values.size();
    fp.rem_values.assign(n * 4, 0);
    for (size_t i = 0; i < n; i++) for (int k = 0; k < 4; k++) fp.rem_values[i + n * k] = last_values[i][k];
    fp.rem_root = red.trees.back().root();
    return fp;
}

// ---- the prover, step by step, keeping every intermediate for parity checks -------------------------------------------------
struct Prover {
    // inputs
    std::vector<vec> trace;            // W columns x n (moved into polys by extend())
    size_t ctx_depth, loop_depth, stack_depth, n, B, N, W;
    vec inputs, outputs;
    ProofOptions options;
    // optional challenge overrides (used to feed identical challenges to the GPU path)
    // intermediates
    vec lde_domain, lde_twiddles;
    std::vector<vec> polys, registers; // coefficient form (W x n) and LDE (W x N)
    std::vector<hash32> trace_leaves;
    MerkleTree trace_tree, constraint_tree;
    ConstraintCoefficients ccoef;
    vec i_evaluations, f_evaluations, t_evaluations;   // over the 8n domain
    bool constraints_ok = true;
    vec constraint_poly, constraint_evaluations;
    u128 z = 0;
    CompositionCoefficients compcoef;
    vec trace_at_z1, trace_at_z2, composition_poly, composed_evaluations;
    FriReduction fri;
    hash32 query_seed0, query_seed1;
    uint64_t pow_nonce = 0;
    std::vector<size_t> positions;
    double phase_ms[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};

    Prover(std::vector<vec> trace_cols, size_t ctx, size_t lp, const vec& in, const vec& out, const ProofOptions& opt)
        : trace(std::move(trace_cols)), ctx_depth(ctx), loop_depth(lp), inputs(in), outputs(out), options(opt) {
        W = trace.size(); n = trace[0].size(); B = opt.extension_factor; N = n * B;
        size_t decoder_width = 15 + ctx + lp;                      // trace_state.rs:116
        if (W <= decoder_width) throw std::runtime_error("user stack must consist of at least one register");
        if (W >= MAX_REGISTER_COUNT) throw std::runtime_error("too many registers");
        if (B < 16 || (B & (B - 1))) throw std::runtime_error("invalid extension factor");
        stack_depth = W - decoder_width;                           // trace_table.rs:41
    }

    TraceState state_at(const std::vector<vec>& table, size_t step) const { TraceState s(ctx_depth, loop_depth, stack_depth); s.update_from_trace(table, step); return s; }
    TraceState last_state() const { return registers.empty() ? state_at(trace, n - 1) : state_at(registers, N - B); }   // trace_table.rs:68

    void step1_extend() {                                          // prover.rs:22-27, trace_table.rs:143
        u128 lde_root = get_root_of_unity(N);
        lde_domain = get_power_series(lde_root, N);
        lde_twiddles.assign(lde_domain.begin(), lde_domain.begin() + N / 2);
        permute(lde_twiddles);
        vec inv_tw = get_inv_twiddles(get_root_of_unity(n), n);
        polys = std::move(trace);
        registers.clear();
        for (vec& poly : polys) {
            interpolate_fft_twiddles(poly.data(), n, inv_tw, true);
            vec reg(N, 0);
            std::copy(poly.begin(), poly.end(), reg.begin());
            eval_fft_twiddles(reg.data(), N, lde_twiddles, true);
            registers.push_back(std::move(reg));
        }
    }
    void step2_trace_tree() {                                      // prover.rs:35, trace_table.rs:174
        trace_leaves.resize(N);
        std::vector<uint8_t> row(W * 16);
        for (size_t i = 0; i < N; i++) {
            for (size_t j = 0; j < W; j++) to_bytes(registers[j][i], row.data() + 16 * j);
            trace_leaves[i] = hash_bytes(row.data(), row.size());
        }
        trace_tree = MerkleTree(trace_leaves);
    }
    Evaluator make_evaluator() {
        TraceState last = last_state();
        vec prog_hash(last.sponge, last.sponge + PROGRAM_DIGEST_SIZE);
        return Evaluator(n, MAX_CONSTRAINT_DEGREE, ctx_depth, loop_depth, stack_depth, n * MAX_CONSTRAINT_DEGREE,
                         ccoef, prog_hash, last.op_counter, inputs, outputs);
    }
    // `coeff_override`: 344 transition/boundary coefficients supplied by the caller instead of prng_vector(trace_root)
    void step3_evaluate_constraints(const vec* coeff_override = nullptr) {   // prover.rs:43-64
        if (coeff_override) ccoef.init(*coeff_override, ctx_depth, loop_depth, stack_depth);
        else ccoef = ConstraintCoefficients(trace_tree.root().data(), ctx_depth, loop_depth, stack_depth);
        Evaluator ev = make_evaluator();
        size_t D = n * MAX_CONSTRAINT_DEGREE;
        i_evaluations.assign(D, 0); f_evaluations.assign(D, 0); t_evaluations.assign(D, 0);
        TraceState cur(ctx_depth, loop_depth, stack_depth), nxt(ctx_depth, loop_depth, stack_depth);
        size_t stride = B / MAX_CONSTRAINT_DEGREE;
        constraints_ok = true;
        for (size_t i = 0; i < N; i += stride) {
            cur.update_from_trace(registers, i);
            nxt.update_from_trace(registers, (i + B) % N);
            size_t step = i / stride;
            ev.evaluate_boundaries(cur, lde_domain[i], i_evaluations[step], f_evaluations[step]);   // constraint_table.rs:45
            t_evaluations[step] = ev.evaluate_transition(cur, nxt, lde_domain[i], step, &constraints_ok);
        }
    }
    void step4_combine() {                                         // prover.rs:73, constraint_table.rs:54
        size_t D = n * MAX_CONSTRAINT_DEGREE;
        vec inv_tw = get_inv_twiddles(get_root_of_unity(D), D);
        vec ie = i_evaluations, fe = f_evaluations, te = t_evaluations;
        interpolate_fft_twiddles(ie.data(), D, inv_tw, true);
        syn_div_in_place(ie, 1);
        constraint_poly = ie;
        interpolate_fft_twiddles(fe.data(), D, inv_tw, true);
        u128 x_last = exp(get_root_of_unity(n), (u128)(n - 1));
        syn_div_in_place(fe, x_last);
        for (size_t i = 0; i < D; i++) constraint_poly[i] = add(constraint_poly[i], fe[i]);
        interpolate_fft_twiddles(te.data(), D, inv_tw, true);
        syn_div_expanded_in_place(te, n, vec{x_last});
        for (size_t i = 0; i < D; i++) constraint_poly[i] = add(constraint_poly[i], te[i]);
    }
    void step5_constraint_tree() {                                 // prover.rs:82-86
        constraint_evaluations.assign(N, 0);
        std::copy(constraint_poly.begin(), constraint_poly.end(), constraint_evaluations.begin());
        eval_fft_twiddles(constraint_evaluations.data(), N, lde_twiddles, true);
        std::vector<hash32> leaves(N / 2);
        for (size_t j = 0; j < N / 2; j++) { to_bytes(constraint_evaluations[2 * j], leaves[j].data()); to_bytes(constraint_evaluations[2 * j + 1], leaves[j].data() + 16); }
        constraint_tree = MerkleTree(leaves);
    }
    // `z_override` / `cc_override` (517 draws incl. draw 0 = z): challenges supplied by the caller
    void step6_deep_composition(const vec* draws_override = nullptr) {   // prover.rs:94-101, 189-201
        if (draws_override) { z = (*draws_override)[0]; compcoef.init(*draws_override); }
        else { z = prng(constraint_tree.root().data()); compcoef = CompositionCoefficients(constraint_tree.root().data()); }
        // trace_table.rs:206-261
        u128 g = get_root_of_unity(n);
        u128 next_z = mul(z, g);
        trace_at_z1.clear(); trace_at_z2.clear();
        for (auto& p : polys) trace_at_z1.push_back(poly_eval(p, z));
        for (auto& p : polys) trace_at_z2.push_back(poly_eval(p, next_z));
        vec t1(n, 0), t2(n, 0);
        for (size_t i = 0; i < W; i++) {
            for (size_t k = 0; k < n; k++) t1[k] = add(t1[k], mul(polys[i][k], compcoef.trace1[i]));
            t1[0] = sub(t1[0], mul(trace_at_z1[i], compcoef.trace1[i]));
            for (size_t k = 0; k < n; k++) t2[k] = add(t2[k], mul(polys[i][k], compcoef.trace2[i]));
            t2[0] = sub(t2[0], mul(trace_at_z2[i], compcoef.trace2[i]));
        }
        syn_div_in_place(t1, z);
        syn_div_in_place(t2, next_z);
        for (size_t k = 0; k < n; k++) t1[k] = add(t1[k], t2[k]);
        size_t poly_size = 1; while (poly_size < get_composition_degree(n)) poly_size <<= 1;   // next_power_of_two
        composition_poly.assign(poly_size, 0);
        size_t inc = get_incremental_trace_degree(n);
        for (size_t k = 0; k < n; k++) composition_poly[k] = add(composition_poly[k], mul(t1[k], compcoef.t1_degree));
        for (size_t k = 0; k < n; k++) composition_poly[inc + k] = add(composition_poly[inc + k], mul(t1[k], compcoef.t2_degree));
        // constraint_poly.rs:39 merge_into
        vec cp = constraint_poly;
        u128 z_value = poly_eval(cp, z);
        cp[0] = sub(cp[0], z_value);
        syn_div_in_place(cp, z);
        for (size_t k = 0; k < cp.size(); k++) composition_poly[k] = add(composition_poly[k], mul(cp[k], compcoef.constraints));
        composed_evaluations.assign(N, 0);
        std::copy(composition_poly.begin(), composition_poly.end(), composed_evaluations.begin());
        eval_fft_twiddles(composed_evaluations.data(), N, lde_twiddles, true);
    }
    void step7_fri() { fri = fri_reduce(composed_evaluations, lde_domain); }       // prover.rs:111
    void step8_queries() {                                         // prover.rs:120-133
        std::vector<uint8_t> roots;
        for (auto& t : fri.trees) roots.insert(roots.end(), t.root().begin(), t.root().end());
        query_seed0 = hash_bytes(roots.data(), roots.size());
        auto pw = find_pow_nonce(query_seed0, options);
        query_seed1 = pw.first; pow_nonce = pw.second;
        positions = compute_query_positions(query_seed1, N, options);
    }
    StarkProof step9_build_proof() {                               // prover.rs:143-165
        StarkProof p;
        p.degree_proof = fri_build_proof(fri, positions);
        for (size_t pos : positions) { vec row; for (auto& r : registers) row.push_back(r[pos]); p.trace_evaluations.push_back(row); }
        BatchMerkleProof tp = trace_tree.prove_batch(positions);
        p.trace_root = trace_tree.root();
        p.trace_nodes = tp.nodes;
        p.domain_depth = tp.depth;
        p.ctx_depth = (uint8_t)ctx_depth; p.loop_depth = (uint8_t)loop_depth; p.stack_depth = (uint8_t)stack_depth;
        p.op_count = (uint32_t)last_state().op_counter;
        p.constraint_root = constraint_tree.root();
        p.constraint_proof = constraint_tree.prove_batch(map_trace_to_constraint_positions(positions));
        p.trace_at_z1 = trace_at_z1; p.trace_at_z2 = trace_at_z2;
        p.pow_nonce = pow_nonce;
        p.options = options;
        return p;
    }
    StarkProof prove(std::function<double()> now_ms = nullptr) {
        auto t = [&]() { return now_ms ? now_ms() : 0.0; };
        double t0 = t(); step1_extend();            phase_ms[0] = t() - t0;
        t0 = t(); step2_trace_tree();               phase_ms[1] = t() - t0;
        t0 = t(); step3_evaluate_constraints();     phase_ms[2] = t() - t0;
        if (!constraints_ok) throw std::runtime_error("transition constraints were not satisfied");
        t0 = t(); step4_combine();                  phase_ms[3] = t() - t0;
        t0 = t(); step5_constraint_tree();          phase_ms[4] = t() - t0;
        t0 = t(); step6_deep_composition();         phase_ms[5] = t() - t0;
        t0 = t(); step7_fri();                      phase_ms[6] = t() - t0;
        t0 = t(); step8_queries();                  phase_ms[7] = t() - t0;
        t0 = t(); StarkProof p = step9_build_proof(); phase_ms[8] = t() - t0;
        return p;
    }
};

// ---- bincode-style serialisation of StarkProof (field order of proof.rs:11-37, merkle.rs:14-18, fri/mod.rs:18-30, options.rs:16-27) ----
struct ByteWriter {
    std::vector<uint8_t> b;
    void u8(uint8_t v) { b.push_back(v); }
    void u32(uint32_t v) { for (int i = 0; i < 4; i++) b.push_back((uint8_t)(v >> (8 * i))); }
    void u64(uint64_t v) { for (int i = 0; i < 8; i++) b.push_back((uint8_t)(v >> (8 * i))); }
    void el(u128 v) { uint8_t t[16]; to_bytes(v, t); b.insert(b.end(), t, t + 16); }
    void h(const hash32& v) { b.insert(b.end(), v.begin(), v.end()); }       // [u8; 32] is a fixed-size array: no length prefix
    void hv(const std::vector<hash32>& v) { u64(v.size()); for (auto& x : v) h(x); }
    void hvv(const std::vector<std::vector<hash32>>& v) { u64(v.size()); for (auto& x : v) hv(x); }
    void ev(const vec& v) { u64(v.size()); for (u128 x : v) el(x); }
};
static inline std::vector<uint8_t> serialize_proof(const StarkProof& p) {
    ByteWriter w;
    w.h(p.trace_root);
    w.u8(p.domain_depth); w.u8(p.ctx_depth); w.u8(p.loop_depth); w.u8(p.stack_depth); w.u32(p.op_count);
    w.hvv(p.trace_nodes);
    w.u64(p.trace_evaluations.size()); for (auto& r : p.trace_evaluations) w.ev(r);
    w.h(p.constraint_root);
    w.hv(p.constraint_proof.values); w.hvv(p.constraint_proof.nodes); w.u8(p.constraint_proof.depth);
    w.ev(p.trace_at_z1); w.ev(p.trace_at_z2);
    w.u64(p.degree_proof.layers.size());
    for (auto& l : p.degree_proof.layers) {
        w.h(l.root);
        w.u64(l.values.size()); for (auto& q : l.values) for (int k = 0; k < 4; k++) w.el(q[k]);
        w.hvv(l.nodes); w.u8(l.depth);
    }
    w.h(p.degree_proof.rem_root); w.ev(p.degree_proof.rem_values);
    w.u64(p.pow_nonce);
    w.u8((uint8_t)__builtin_ctzll((unsigned long long)p.options.extension_factor)); w.u8((uint8_t)p.options.num_queries);
    w.u8((uint8_t)p.options.grinding_factor); w.u8(0 /* blake3, options.rs:107 */);
    return w.b;
}
struct ByteReader {
    const uint8_t* p; size_t n, o = 0;
    ByteReader(const uint8_t* d, size_t len) : p(d), n(len) {}
    void need(size_t k) { if (o + k > n) throw std::runtime_error("proof truncated"); }
    uint8_t u8() { need(1); return p[o++]; }
    uint32_t u32() { need(4); uint32_t v = 0; for (int i = 0; i < 4; i++) v |= (uint32_t)p[o + i] << (8 * i); o += 4; return v; }
    uint64_t u64() { need(8); uint64_t v = 0; for (int i = 0; i < 8; i++) v |= (uint64_t)p[o + i] << (8 * i); o += 8; return v; }
    u128 el() { need(16); u128 v = from_bytes(p + o); o += 16; return v; }
    hash32 h() { need(32); hash32 v; memcpy(v.data(), p + o, 32); o += 32; return v; }
    std::vector<hash32> hv() { size_t k = u64(); need(k * 32); std::vector<hash32> v(k); for (auto& x : v) x = h(); return v; }
    std::vector<std::vector<hash32>> hvv() { size_t k = u64(); need(k * 8); std::vector<std::vector<hash32>> v(k); for (auto& x : v) x = hv(); return v; }
    vec ev() { size_t k = u64(); need(k * 16); vec v(k); for (auto& x : v) x = el(); return v; }
};
static inline StarkProof deserialize_proof(const uint8_t* d, size_t len) {
    ByteReader r(d, len);
    StarkProof p;
    p.trace_root = r.h();
    p.domain_depth = r.u8(); p.ctx_depth = r.u8(); p.loop_depth = r.u8(); p.stack_depth = r.u8(); p.op_count = r.u32();
    p.trace_nodes = r.hvv();
    size_t k = r.u64(); r.need(k * 8); p.trace_evaluations.resize(k); for (auto& v : p.trace_evaluations) v = r.ev();
    p.constraint_root = r.h();
    p.constraint_proof.values = r.hv(); p.constraint_proof.nodes = r.hvv(); p.constraint_proof.depth = r.u8();
    p.trace_at_z1 = r.ev(); p.trace_at_z2 = r.ev();
    k = r.u64(); r.need(k * 32); p.degree_proof.layers.resize(k);
    for (auto& l : p.degree_proof.layers) {
        l.root = r.h();
        size_t m = r.u64(); r.need(m * 64); l.values.resize(m);
        for (auto& q : l.values) for (int c = 0; c < 4; c++) q[c] = r.el();
        l.nodes = r.hvv(); l.depth = r.u8();
    }
    p.degree_proof.rem_root = r.h(); p.degree_proof.rem_values = r.ev();
    p.pow_nonce = r.u64();
    p.options.extension_factor = (size_t)1 << r.u8(); p.options.num_queries = r.u8(); p.options.grinding_factor = r.u8();
    if (r.u8() != 0) throw std::runtime_error("unsupported hash function");
    if (r.o != len) throw std::runtime_error("trailing bytes after proof");
    return p;
}

}  // namespace orc
