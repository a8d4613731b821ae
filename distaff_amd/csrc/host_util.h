// Host-side pieces of the prover that the reference takes from third-party crates or runs between the heavy phases:
//   * BLAKE3 of short messages (query seed = hash of the FRI roots, prover.rs:120-127; PoW digest, proof_of_work.rs:22-31)
//   * StdRng (ChaCha20) + Uniform sampling of rand 0.7.3 -- field::prng / prng_vector (src/math/field.rs:264-275) and
//     compute_query_positions (src/stark/utils/mod.rs:25-44).  `rand` is not part of the reference tree; the algorithm below is
//     a restatement of the crate's published behaviour (ChaCha20 block function per RFC 7539 with a 64-bit counter, words
//     consumed little-endian in order, u128 = two u64 low half first, widening-multiply rejection sampling).
//   * the coefficient layouts of utils/coefficients.rs and the helpers of utils/mod.rs.
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>
#include "blake3_dev.h"
#include "fe.h"

namespace dsth {

typedef unsigned __int128 u128;

inline u128 fe_to_u128(const fe& a) { return (u128)a.v[0] | ((u128)a.v[1] << 32) | ((u128)a.v[2] << 64) | ((u128)a.v[3] << 96); }
inline fe fe_from_u128(u128 x) { return fe_make((uint32_t)x, (uint32_t)(x >> 32), (uint32_t)(x >> 64), (uint32_t)(x >> 96)); }
inline fe fe_from_bytes(const uint8_t* b) { fe r; memcpy(r.v, b, 16); return r; }       // little-endian host
inline void fe_to_bytes(const fe& a, uint8_t* b) { memcpy(b, a.v, 16); }

// ---- BLAKE3, messages of at most 2048 bytes ------------------------------------------------------------------------------------
inline void b3_chunk(const uint8_t* p, size_t len, uint32_t chunk_index, bool root, uint32_t cv[8]) {
    b3_iv(cv);
    size_t blocks = len == 0 ? 1 : (len + 63) / 64;
    for (size_t b = 0; b < blocks; b++) {
        uint8_t buf[64];
        memset(buf, 0, 64);
        size_t bl = len - b * 64 < 64 ? len - b * 64 : 64;
        memcpy(buf, p + b * 64, bl);
        uint32_t m[16];
        memcpy(m, buf, 64);
        uint32_t flags = (b == 0 ? B3_CHUNK_START : 0u) | (b == blocks - 1 ? (B3_CHUNK_END | (root ? B3_ROOT : 0u)) : 0u);
        b3_compress(cv, m, chunk_index, 0, (uint32_t)bl, flags);
    }
}
// chaining value of the subtree over `len` > 1024 bytes starting at chunk `chunk0`, or of a single chunk (never the root)
inline void b3_subtree(const uint8_t* in, size_t len, uint64_t chunk0, uint32_t cv[8]) {
    if (len <= 1024) { b3_chunk(in, len, chunk0, false, cv); return; }
    size_t left = 1024;                                  // largest power-of-two number of chunks that leaves >= 1 byte on the right
    while (left * 2 < len) left *= 2;
    uint32_t m[16];
    b3_subtree(in, left, chunk0, m);
    b3_subtree(in + left, len - left, chunk0 + left / 1024, m + 8);
    b3_iv(cv);
    b3_compress(cv, m, 0, 0, 64, B3_PARENT);
}
// BLAKE3 of any length (the prover itself only hashes rows of at most two chunks and 64-byte node pairs)
inline bool blake3_short(const uint8_t* in, size_t len, uint8_t out[32]) {
    uint32_t cv[8];
    if (len <= 1024) b3_chunk(in, len, 0, true, cv);
    else {
        size_t left = 1024;
        while (left * 2 < len) left *= 2;
        uint32_t m[16];
        b3_subtree(in, left, 0, m);
        b3_subtree(in + left, len - left, left / 1024, m + 8);
        b3_iv(cv);
        b3_compress(cv, m, 0, 0, 64, B3_PARENT | B3_ROOT);
    }
    memcpy(out, cv, 32);
    return true;
}

// ---- ChaCha20 StdRng + Uniform --------------------------------------------------------------------------------------------------
struct StdRng {
    uint32_t key[8], buf[16];
    uint64_t counter = 0;
    int idx = 16;
    explicit StdRng(const uint8_t seed[32]) { memcpy(key, seed, 32); }
    static inline uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
    void refill() {
        uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
        for (int i = 0; i < 8; i++) in[4 + i] = key[i];
        in[12] = (uint32_t)counter; in[13] = (uint32_t)(counter >> 32); in[14] = 0; in[15] = 0;
        uint32_t x[16];
        memcpy(x, in, 64);
#define DST_QR(a, b, c, d) x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16); x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12); \
                           x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8);  x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7);
        for (int r = 0; r < 10; r++) {
            DST_QR(0, 4, 8, 12) DST_QR(1, 5, 9, 13) DST_QR(2, 6, 10, 14) DST_QR(3, 7, 11, 15)
            DST_QR(0, 5, 10, 15) DST_QR(1, 6, 11, 12) DST_QR(2, 7, 8, 13) DST_QR(3, 4, 9, 14)
        }
#undef DST_QR
        for (int i = 0; i < 16; i++) buf[i] = x[i] + in[i];
        counter++; idx = 0;
    }
    uint32_t next_u32() { if (idx >= 16) refill(); return buf[idx++]; }
    uint64_t next_u64() { uint64_t lo = next_u32(), hi = next_u32(); return lo | (hi << 32); }
    u128 next_u128() { u128 lo = next_u64(), hi = next_u64(); return lo | (hi << 64); }
};

inline void mul_wide(u128 a, u128 b, u128& hi, u128& lo) {
    uint64_t a0 = (uint64_t)a, a1 = (uint64_t)(a >> 64), b0 = (uint64_t)b, b1 = (uint64_t)(b >> 64);
    u128 p00 = (u128)a0 * b0, p01 = (u128)a0 * b1, p10 = (u128)a1 * b0, p11 = (u128)a1 * b1;
    u128 mid = (p00 >> 64) + (uint64_t)p01 + (uint64_t)p10;
    lo = (u128)(uint64_t)p00 | (mid << 64);
    hi = p11 + (p01 >> 64) + (p10 >> 64) + (mid >> 64);
}
static const u128 FIELD_P = (((u128)0xFFFFFFFFFFFFFFFFull) << 64) | 0xFFFFD30000000001ull;

inline u128 uniform_field(StdRng& g) {               // Uniform::from(0..M).sample(rng)
    const u128 range = FIELD_P, max = ~(u128)0;
    const u128 zone = max - ((max - range + 1) % range);
    for (;;) {
        u128 hi, lo;
        mul_wide(g.next_u128(), range, hi, lo);
        if (lo <= zone) return hi;
    }
}
inline uint64_t uniform_usize(StdRng& g, uint64_t range) {   // Uniform::from(0..range).sample(rng), 64-bit usize
    const uint64_t max = ~(uint64_t)0, zone = max - ((max - range + 1) % range);
    for (;;) {
        u128 w = (u128)g.next_u64() * range;
        if ((uint64_t)w <= zone) return (uint64_t)(w >> 64);
    }
}
inline void prng_vector(const uint8_t seed[32], size_t count, fe* out) {        // field.rs:271
    StdRng g(seed);
    for (size_t i = 0; i < count; i++) out[i] = fe_from_u128(uniform_field(g));
}
inline fe prng(const uint8_t seed[32]) { fe r; prng_vector(seed, 1, &r); return r; }   // field.rs:264

inline int query_positions(const uint8_t seed[32], uint64_t domain_size, uint32_t blowup, uint32_t num_queries, std::vector<uint64_t>& out) {  // utils/mod.rs:25
    StdRng g(seed);
    out.clear();
    for (int it = 0; it < 1000; it++) {
        uint64_t v = uniform_usize(g, domain_size);
        if (v % blowup == 0) continue;
        bool dup = false;
        for (uint64_t p : out) if (p == v) { dup = true; break; }
        if (dup) continue;
        out.push_back(v);
        if (out.size() >= num_queries) break;
    }
    return out.size() >= num_queries ? 0 : -1;
}

}  // namespace dsth
