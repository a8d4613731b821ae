// Host-side proof assembly: which Merkle leaves / nodes a batch opening needs, and the StarkProof wire format.
//
// Mirrors /root/reference/src/crypto/merkle.rs:64-124 (prove_batch), :296-312 (map_indexes / normalize_indexes),
// src/stark/fri/utils.rs:4-14 (get_augmented_positions), src/stark/utils/mod.rs:46-53 (map_trace_to_constraint_positions),
// src/stark/fri/prover.rs:55-95 (build_proof) and the field order of src/stark/proof.rs:11-37, src/stark/fri/mod.rs:18-30,
// src/crypto/merkle.rs:14-18, src/stark/options.rs:16-27 under bincode's default encoding (src/main.rs:44): little-endian
// fixed-width integers, u64 length prefix for Vec, no prefix for fixed-size arrays, u128 as 16 little-endian bytes.
#pragma once
#include <stdint.h>
#include <algorithm>
#include <map>
#include <set>
#include <vector>

namespace dsth {

struct NodeRef { bool is_leaf; uint64_t index; };

struct BatchPlan {
    std::vector<uint64_t> values;                 // leaf index whose value goes to values[i]
    std::vector<std::vector<NodeRef>> nodes;      // per normalised index pair, in the reference's order
    uint8_t depth = 0;
};

inline BatchPlan plan_batch(const std::vector<uint64_t>& indexes_in, uint64_t num_leaves) {
    BatchPlan plan;
    std::map<uint64_t, size_t> index_map;
    for (size_t i = 0; i < indexes_in.size(); i++) index_map[indexes_in[i]] = i;
    std::set<uint64_t> norm;
    for (uint64_t i : indexes_in) norm.insert(i - (i & 1));
    plan.values = indexes_in;
    std::vector<uint64_t> next;
    for (uint64_t index : norm) {
        bool has1 = index_map.count(index) != 0, has2 = index_map.count(index + 1) != 0;
        if (has1 && has2) plan.nodes.push_back({});
        else if (has1) plan.nodes.push_back({NodeRef{true, index + 1}});
        else plan.nodes.push_back({NodeRef{true, index}});
        next.push_back((index + num_leaves) >> 1);
    }
    uint8_t depth = 0;
    while (((uint64_t)1 << depth) < num_leaves) depth++;
    plan.depth = depth;
    for (int d = 1; d < depth; d++) {
        std::vector<uint64_t> cur = next;
        next.clear();
        size_t i = 0;
        while (i < cur.size()) {
            uint64_t sib = cur[i] ^ 1;
            if (i + 1 < cur.size() && cur[i + 1] == sib) i += 1;
            else plan.nodes[i].push_back(NodeRef{false, sib});
            next.push_back(sib >> 1);
            i += 1;
        }
    }
    return plan;
}

inline std::vector<uint64_t> augmented_positions(const std::vector<uint64_t>& positions, uint64_t column_length) {
    uint64_t row_length = column_length / 4;
    std::vector<uint64_t> r;
    for (uint64_t p : positions) {
        uint64_t ap = p % row_length;
        if (std::find(r.begin(), r.end(), ap) == r.end()) r.push_back(ap);
    }
    return r;
}
inline std::vector<uint64_t> constraint_positions(const std::vector<uint64_t>& positions) {
    std::vector<uint64_t> r;
    for (uint64_t p : positions) {
        uint64_t cp = p / 2;
        if (std::find(r.begin(), r.end(), cp) == r.end()) r.push_back(cp);
    }
    return r;
}

struct Writer {
    std::vector<uint8_t> b;
    void u8(uint8_t v) { b.push_back(v); }
    void u32(uint32_t v) { for (int i = 0; i < 4; i++) b.push_back((uint8_t)(v >> (8 * i))); }
    void u64(uint64_t v) { for (int i = 0; i < 8; i++) b.push_back((uint8_t)(v >> (8 * i))); }
    void raw(const void* p, size_t n) { const uint8_t* q = (const uint8_t*)p; b.insert(b.end(), q, q + n); }
};

}  // namespace dsth
