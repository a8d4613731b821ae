// ORACLE (test infrastructure, NOT product code) -- Merkle tree restatement.
// Follows /root/reference/src/crypto/merkle.rs: MerkleTree::new :25, root :37, prove :47,
// prove_batch :64-124, verify :127, verify_batch :154-263, build_merkle_nodes :269-294,
// map_indexes :296, normalize_indexes :306. The hash is BLAKE3 (the only serialisable HashFunction,
// src/stark/options.rs:97-120).
#pragma once
#include "blake3.hpp"
#include <vector>
#include <array>
#include <map>
#include <set>
#include <algorithm>

namespace orc {

typedef std::array<uint8_t, 32> hash32;

static inline hash32 hash_bytes(const uint8_t* p, size_t len) { hash32 h; blake3(p, len, h.data()); return h; }
static inline hash32 hash_2x1(const hash32& a, const hash32& b) {
    uint8_t buf[64];
    memcpy(buf, a.data(), 32); memcpy(buf + 32, b.data(), 32);
    return hash_bytes(buf, 64);
}

static inline std::vector<hash32> build_merkle_nodes(const std::vector<hash32>& leaves) {   // merkle.rs:269
    size_t n = leaves.size() / 2;
    std::vector<hash32> nodes(2 * n);
    nodes[0].fill(0);
    for (size_t i = 0; i < n; i++) nodes[n + i] = hash_2x1(leaves[2 * i], leaves[2 * i + 1]);
    for (size_t i = n - 1; i >= 1; i--) nodes[i] = hash_2x1(nodes[2 * i], nodes[2 * i + 1]);
    return nodes;
}

struct BatchMerkleProof {                                       // merkle.rs:14
    std::vector<hash32> values;
    std::vector<std::vector<hash32>> nodes;
    uint8_t depth = 0;
};

static inline std::vector<size_t> normalize_indexes(const std::vector<size_t>& idx) {   // merkle.rs:306
    std::set<size_t> s;
    for (size_t i : idx) s.insert(i - (i & 1));
    return std::vector<size_t>(s.begin(), s.end());
}

struct MerkleTree {
    std::vector<hash32> nodes, values;
    MerkleTree() {}
    explicit MerkleTree(std::vector<hash32> leaves) {           // merkle.rs:25
        assert(leaves.size() >= 2 && (leaves.size() & (leaves.size() - 1)) == 0);
        nodes = build_merkle_nodes(leaves);
        values = std::move(leaves);
    }
    const hash32& root() const { return nodes[1]; }             // merkle.rs:37

    std::vector<hash32> prove(size_t index) const {             // merkle.rs:47
        std::vector<hash32> proof{values[index], values[index ^ 1]};
        size_t i = (index + nodes.size()) >> 1;
        while (i > 1) { proof.push_back(nodes[i ^ 1]); i >>= 1; }
        return proof;
    }

    BatchMerkleProof prove_batch(const std::vector<size_t>& indexes_in) const {   // merkle.rs:64
        size_t n = values.size();
        std::map<size_t, size_t> index_map;
        for (size_t i = 0; i < indexes_in.size(); i++) { assert(indexes_in[i] <= n); index_map[indexes_in[i]] = i; }
        assert(index_map.size() == indexes_in.size());
        std::vector<size_t> indexes = normalize_indexes(indexes_in);
        BatchMerkleProof pr;
        pr.values.assign(index_map.size(), hash32{});
        std::vector<size_t> next;
        for (size_t index : indexes) {
            const hash32& v1 = values[index];
            const hash32& v2 = values[index + 1];
            auto i1 = index_map.find(index), i2 = index_map.find(index + 1);
            if (i1 != index_map.end()) {
                if (i2 != index_map.end()) { pr.values[i1->second] = v1; pr.values[i2->second] = v2; pr.nodes.push_back({}); }
                else { pr.values[i1->second] = v1; pr.nodes.push_back({v2}); }
            } else { pr.values[i2->second] = v2; pr.nodes.push_back({v1}); }
            next.push_back((index + n) >> 1);
        }
        uint8_t depth = (uint8_t)__builtin_ctzll((unsigned long long)n);
        for (int d = 1; d < depth; d++) {
            std::vector<size_t> cur = next;
            next.clear();
            size_t i = 0;
            while (i < cur.size()) {
                size_t sib = cur[i] ^ 1;
                if (i + 1 < cur.size() && cur[i + 1] == sib) i += 1;
                else pr.nodes[i].push_back(nodes[sib]);
                next.push_back(sib >> 1);
                i += 1;
            }
        }
        pr.depth = depth;
        return pr;
    }

    static bool verify(const hash32& root, size_t index, const std::vector<hash32>& proof) {   // merkle.rs:127
        size_t r = index & 1;
        hash32 v = hash_2x1(proof[r], proof[1 - r]);
        size_t idx = (index + ((size_t)1 << (proof.size() - 1))) >> 1;
        for (size_t i = 2; i < proof.size(); i++) {
            v = (idx & 1) == 0 ? hash_2x1(v, proof[i]) : hash_2x1(proof[i], v);
            idx >>= 1;
        }
        return v == root;
    }

    static bool verify_batch(const hash32& root, const std::vector<size_t>& indexes_in, const BatchMerkleProof& proof) {  // merkle.rs:154
        std::map<size_t, hash32> v;
        size_t offset = (size_t)1 << proof.depth;
        std::map<size_t, size_t> index_map;
        for (size_t i = 0; i < indexes_in.size(); i++) {
            if (indexes_in[i] > offset - 1) return false;   // the reference asserts here (merkle.rs:300)
            index_map[indexes_in[i]] = i;
        }
        if (index_map.size() != indexes_in.size()) return false;
        std::vector<size_t> indexes = normalize_indexes(indexes_in);
        if (indexes.size() != proof.nodes.size()) return false;
        std::vector<size_t> next, ptrs;
        for (size_t i = 0; i < indexes.size(); i++) {
            size_t index = indexes[i];
            hash32 a, b;
            auto i1 = index_map.find(index), i2 = index_map.find(index + 1);
            if (i1 != index_map.end()) {
                if (proof.values.size() <= i1->second) return false;
                a = proof.values[i1->second];
                if (i2 != index_map.end()) {
                    if (proof.values.size() <= i2->second) return false;
                    b = proof.values[i2->second];
                    ptrs.push_back(0);
                } else {
                    if (proof.nodes[i].size() < 1) return false;
                    b = proof.nodes[i][0];
                    ptrs.push_back(1);
                }
            } else {
                if (proof.nodes[i].size() < 1) return false;
                a = proof.nodes[i][0];
                if (i2 == index_map.end()) return false;
                if (proof.values.size() <= i2->second) return false;
                b = proof.values[i2->second];
                ptrs.push_back(1);
            }
            size_t parent_index = (offset + index) >> 1;
            v[parent_index] = hash_2x1(a, b);
            next.push_back(parent_index);
        }
        for (int d = 1; d < proof.depth; d++) {
            std::vector<size_t> cur = next;
            next.clear();
            size_t i = 0;
            while (i < cur.size()) {
                size_t node_index = cur[i], sib_index = node_index ^ 1;
                hash32 sib;
                if (i + 1 < cur.size() && cur[i + 1] == sib_index) {
                    auto s = v.find(sib_index);
                    if (s == v.end()) return false;
                    sib = s->second;
                    i += 1;
                } else {
                    size_t p = ptrs[i];
                    if (proof.nodes[i].size() <= p) return false;
                    sib = proof.nodes[i][p];
                    ptrs[i] += 1;
                }
                auto nd = v.find(node_index);
                if (nd == v.end()) return false;
                hash32 parent = (node_index & 1) ? hash_2x1(sib, nd->second) : hash_2x1(nd->second, sib);
                size_t parent_index = node_index >> 1;
                v[parent_index] = parent;
                next.push_back(parent_index);
                i += 1;
            }
        }
        auto r = v.find(1);
        return r != v.end() && r->second == root;
    }
};

}  // namespace orc
