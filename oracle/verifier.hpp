// ORACLE (test infrastructure, NOT product code) -- restatement of the reference verifier; it is the end-to-end
// acceptance check for proofs produced by the HIP prover path.
//
// Follows /root/reference/src/stark/verifier.rs (verify :11-75, evaluate_constraints :79-96, compose_registers :98-136,
// compose_constraints :138-162), src/stark/fri/verifier.rs (verify :11-89, verify_remainder :91-124, get_column_values :128,
// build_layer_merkle_proof :141), src/stark/utils/proof_of_work.rs:34-56 (verify_pow_nonce), src/stark/proof.rs:91-104,139-154.
// Error strings are the reference's own.
#pragma once
#include "prover.hpp"

namespace orc {

struct VerifyResult { bool ok; std::string error; };

static inline VerifyResult fri_verify_remainder(const vec& remainder, size_t max_degree_plus_1, u128 domain_root, size_t extension_factor) {   // fri/verifier.rs:91
    if (max_degree_plus_1 > remainder.size()) return {false, "remainder degree is greater than number of remainder values"};
    std::vector<size_t> positions;
    for (size_t i = 0; i < remainder.size(); i++) if (i % extension_factor != 0) positions.push_back(i);
    vec domain = get_power_series(domain_root, remainder.size());
    vec xs, ys;
    for (size_t i = 0; i < max_degree_plus_1; i++) { xs.push_back(domain[positions[i]]); ys.push_back(remainder[positions[i]]); }
    vec poly = poly_interpolate(xs, ys);
    for (size_t i = max_degree_plus_1; i < positions.size(); i++) {
        size_t p = positions[i];
        if (poly_eval(poly, domain[p]) != remainder[p])
            return {false, "remainder is not a valid degree " + std::to_string(max_degree_plus_1 - 1) + " polynomial"};
    }
    return {true, ""};
}

static inline VerifyResult fri_verify(const FriProof& proof, const vec& evaluations_in, const std::vector<size_t>& positions_in,
                                      size_t max_degree, const ProofOptions& options) {                                       // fri/verifier.rs:11
    size_t domain_size = ((size_t)1 << proof.layers[0].depth) * 4;
    u128 domain_root = get_root_of_unity(domain_size);
    u128 quartic_roots[4] = {1, exp(domain_root, (u128)(domain_size / 4)), exp(domain_root, (u128)(domain_size / 2)), exp(domain_root, (u128)(domain_size * 3 / 4))};
    size_t max_degree_plus_1 = max_degree + 1;
    std::vector<size_t> positions = positions_in;
    vec evaluations = evaluations_in;
    for (size_t depth = 0; depth < proof.layers.size(); depth++) {
        const FriLayer& layer = proof.layers[depth];
        std::vector<size_t> augmented = get_augmented_positions(positions, domain_size);
        // get_column_values
        size_t row_length = domain_size / 4;
        vec column_values;
        for (size_t position : positions) {
            size_t idx = std::find(augmented.begin(), augmented.end(), position % row_length) - augmented.begin();
            if (idx >= layer.values.size()) return {false, "evaluations did not match column value at depth " + std::to_string(depth)};
            column_values.push_back(layer.values[idx][position / row_length]);
        }
        if (evaluations != column_values) return {false, "evaluations did not match column value at depth " + std::to_string(depth)};
        BatchMerkleProof mp;
        mp.values = fri_hash_values(layer.values); mp.nodes = layer.nodes; mp.depth = layer.depth;
        if (!MerkleTree::verify_batch(layer.root, augmented, mp)) return {false, "verification of Merkle proof failed at layer " + std::to_string(depth)};
        std::vector<quad> xs;
        for (size_t i : augmented) {
            u128 xe = exp(domain_root, (u128)i);
            xs.push_back({mul(quartic_roots[0], xe), mul(quartic_roots[1], xe), mul(quartic_roots[2], xe), mul(quartic_roots[3], xe)});
        }
        std::vector<quad> row_polys = quartic_interpolate_batch(xs, layer.values);
        u128 special_x = prng(layer.root.data());
        evaluations = quartic_evaluate_batch(row_polys, special_x);
        domain_root = exp(domain_root, 4);
        max_degree_plus_1 /= 4;
        domain_size /= 4;
        positions = augmented;
    }
    for (size_t i = 0; i < positions.size() && i < evaluations.size(); i++)
        if (proof.rem_values[positions[i]] != evaluations[i]) return {false, "remainder values are inconsistent with values of the last column"};
    return fri_verify_remainder(proof.rem_values, max_degree_plus_1, domain_root, options.extension_factor);
}

static inline VerifyResult verify(const uint8_t program_hash[32], const vec& inputs, const vec& outputs, const StarkProof& proof) {   // verifier.rs:11
    const ProofOptions& options = proof.options;
    // 1 -- proof of work and query positions
    std::vector<uint8_t> fri_roots;
    for (auto& l : proof.degree_proof.layers) fri_roots.insert(fri_roots.end(), l.root.begin(), l.root.end());
    fri_roots.insert(fri_roots.end(), proof.degree_proof.rem_root.begin(), proof.degree_proof.rem_root.end());
    hash32 seed = hash_bytes(fri_roots.data(), fri_roots.size());
    hash32 seed1 = pow_hash(seed, proof.pow_nonce);                 // proof_of_work.rs:34
    if (!pow_check(seed1, options.grinding_factor)) return {false, "seed proof-of-work verification failed"};
    std::vector<size_t> t_positions = compute_query_positions(seed1, proof.domain_size(), options);
    std::vector<size_t> c_positions = map_trace_to_constraint_positions(t_positions);
    // 2 -- minimum operation count
    if (proof.op_count < MIN_TRACE_LENGTH) return {false, "Verification of minimum operation count failed"};
    // 3 -- Merkle proofs
    BatchMerkleProof tp;
    tp.nodes = proof.trace_nodes; tp.depth = proof.domain_depth;
    for (auto& row : proof.trace_evaluations) {                     // proof.rs:91
        std::vector<uint8_t> b(row.size() * 16);
        for (size_t j = 0; j < row.size(); j++) to_bytes(row[j], b.data() + 16 * j);
        tp.values.push_back(hash_bytes(b.data(), b.size()));
    }
    if (!MerkleTree::verify_batch(proof.trace_root, t_positions, tp)) return {false, "verification of trace Merkle proof failed"};
    if (!MerkleTree::verify_batch(proof.constraint_root, c_positions, proof.constraint_proof)) return {false, "verification of constraint Merkle proof failed"};
    // 4 -- constraint evaluations at the DEEP point z
    u128 z = prng(proof.constraint_root.data());
    size_t ctx = proof.ctx_depth, lp = proof.loop_depth, st = proof.stack_depth;
    size_t width = 15 + ctx + lp + st;
    if (proof.trace_at_z1.size() != width || proof.trace_at_z2.size() != width) return {false, "invalid deep values"};
    size_t trace_length = proof.trace_length();
    ConstraintCoefficients ccoef(proof.trace_root.data(), ctx, lp, st);
    vec prog_hash{from_bytes(program_hash), from_bytes(program_hash + 16)};      // evaluator.rs:432
    Evaluator ev(trace_length, options.extension_factor, ctx, lp, st, proof.domain_size(), ccoef, prog_hash, (u128)proof.op_count, inputs, outputs);
    TraceState s1 = TraceState::from_vec(ctx, lp, st, proof.trace_at_z1);
    TraceState s2 = TraceState::from_vec(ctx, lp, st, proof.trace_at_z2);
    u128 i_value, f_value;
    ev.evaluate_boundaries(s1, z, i_value, f_value);                // verifier.rs:79
    u128 t_value = ev.evaluate_transition_at(s1, s2, z);
    u128 zz = sub(z, 1);
    u128 c_at_z = div(i_value, zz);
    zz = sub(z, ev.get_x_at_last_step());
    c_at_z = add(c_at_z, div(f_value, zz));
    zz = div(sub(exp(z, (u128)trace_length), 1), zz);
    c_at_z = add(c_at_z, div(t_value, zz));
    // 5 -- composition polynomial evaluations
    CompositionCoefficients cc(proof.constraint_root.data());
    u128 lde_root = get_root_of_unity(proof.domain_size());
    u128 next_z = mul(z, get_root_of_unity(trace_length));
    u128 incremental_degree = (u128)get_incremental_trace_degree(trace_length);
    if (proof.trace_evaluations.size() != t_positions.size()) return {false, "invalid number of trace evaluations"};
    vec evaluations;
    for (size_t q = 0; q < t_positions.size(); q++) {               // verifier.rs:98
        const vec& regs = proof.trace_evaluations[q];
        u128 x = exp(lde_root, (u128)t_positions[q]);
        u128 comp = 0;
        for (size_t i = 0; i < regs.size(); i++) {
            u128 t1 = div(sub(regs[i], proof.trace_at_z1[i]), sub(x, z));
            comp = add(comp, mul(t1, cc.trace1[i]));
            u128 t2 = div(sub(regs[i], proof.trace_at_z2[i]), sub(x, next_z));
            comp = add(comp, mul(t2, cc.trace2[i]));
        }
        u128 xp = exp(x, incremental_degree);
        u128 adj = mul(mul(comp, xp), cc.t2_degree);
        comp = add(mul(comp, cc.t1_degree), adj);
        // verifier.rs:138 -- constraint part
        size_t position = t_positions[q];
        size_t leaf_idx = std::find(c_positions.begin(), c_positions.end(), position / 2) - c_positions.begin();
        if (leaf_idx >= proof.constraint_proof.values.size()) return {false, "invalid constraint proof"};
        u128 c_eval = from_bytes(proof.constraint_proof.values[leaf_idx].data() + (position % 2) * 16);
        u128 c_comp = mul(div(sub(c_eval, c_at_z), sub(x, z)), cc.constraints);
        evaluations.push_back(add(comp, c_comp));
    }
    // 6 -- low-degree proof
    VerifyResult r = fri_verify(proof.degree_proof, evaluations, t_positions, get_composition_degree(trace_length), options);
    if (!r.ok) return {false, "verification of low-degree proof failed: " + r.error};
    return {true, ""};
}

}  // namespace orc
