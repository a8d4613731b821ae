// ORACLE (test infrastructure, NOT product code) -- Fiat-Shamir randomness of the reference.
//
// The reference draws field elements / positions with the third-party `rand` crate (^0.7.3, Cargo.toml:20):
//   field::prng / prng_vector   /root/reference/src/math/field.rs:264-275   StdRng::from_seed + Uniform<u128>(0..M)
//   compute_query_positions     /root/reference/src/stark/utils/mod.rs:25-44 StdRng::from_seed + Uniform<usize>(0..N)
// `rand` is NOT under /root/reference.  This file restates the published algorithms of rand 0.7.3 / rand_chacha 0.2:
//   * StdRng = ChaCha20 (20 rounds), key = the 32-byte seed, 64-bit block counter starting at 0, 64-bit stream id 0,
//     output consumed as little-endian u32 words in block order; next_u64 = two consecutive words (low word first);
//     a u128 sample = two next_u64 calls, low half first.
//   * Uniform<T>::sample (UniformInt, "widening multiply" rejection): range = high - low,
//     zone = MAX - ((MAX - range + 1) % range); loop { v = gen(); (hi, lo) = widening_mul(v, range);
//     if lo <= zone { return low + hi } }.
// PARITY STATUS -- "parity unpinned" for the two rules named below: the ChaCha20 block function is pinned against OpenSSL's chacha20 keystream (tests/golden);
// the word-consumption order and the Uniform rule are restated from the crates' published sources and are
// UNPINNED (no Rust toolchain and no reference fixture with concrete draws exists -- SURVEY.md section 8c).
#pragma once
#include "field.hpp"
#include <cstring>

namespace orc {

struct ChaCha20Rng {
    uint32_t key[8];
    uint64_t counter = 0;
    uint32_t buf[16];
    int idx = 16;

    explicit ChaCha20Rng(const uint8_t seed[32]) {
        for (int i = 0; i < 8; i++)
            key[i] = (uint32_t)seed[4 * i] | ((uint32_t)seed[4 * i + 1] << 8) | ((uint32_t)seed[4 * i + 2] << 16) | ((uint32_t)seed[4 * i + 3] << 24);
    }
    static inline uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
    static inline void qr(uint32_t* s, int a, int b, int c, int d) {
        s[a] += s[b]; s[d] = rotl(s[d] ^ s[a], 16);
        s[c] += s[d]; s[b] = rotl(s[b] ^ s[c], 12);
        s[a] += s[b]; s[d] = rotl(s[d] ^ s[a], 8);
        s[c] += s[d]; s[b] = rotl(s[b] ^ s[c], 7);
    }
    void refill() {
        uint32_t init[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
        for (int i = 0; i < 8; i++) init[4 + i] = key[i];
        init[12] = (uint32_t)counter; init[13] = (uint32_t)(counter >> 32); init[14] = 0; init[15] = 0;
        uint32_t s[16];
        memcpy(s, init, sizeof(s));
        for (int r = 0; r < 10; r++) {
            qr(s, 0, 4, 8, 12); qr(s, 1, 5, 9, 13); qr(s, 2, 6, 10, 14); qr(s, 3, 7, 11, 15);
            qr(s, 0, 5, 10, 15); qr(s, 1, 6, 11, 12); qr(s, 2, 7, 8, 13); qr(s, 3, 4, 9, 14);
        }
        for (int i = 0; i < 16; i++) buf[i] = s[i] + init[i];
        counter += 1;
        idx = 0;
    }
    uint32_t next_u32() { if (idx >= 16) refill(); return buf[idx++]; }
    uint64_t next_u64() { uint64_t lo = next_u32(); uint64_t hi = next_u32(); return lo | (hi << 32); }
    u128 next_u128() { u128 lo = next_u64(); u128 hi = next_u64(); return lo | (hi << 64); }
};

// Uniform<u128>::sample over [0, range)
static inline u128 uniform_u128(ChaCha20Rng& g, u128 range) {
    u128 max = ~(u128)0;
    u128 zone = max - ((max - range + 1) % range);
    for (;;) {
        u128 v = g.next_u128();
        u128 hi, lo;
        mul_wide(v, range, hi, lo);
        if (lo <= zone) return hi;
    }
}
// Uniform<usize>::sample over [0, range) on a 64-bit target
static inline uint64_t uniform_u64(ChaCha20Rng& g, uint64_t range) {
    uint64_t max = ~(uint64_t)0;
    uint64_t zone = max - ((max - range + 1) % range);
    for (;;) {
        uint64_t v = g.next_u64();
        u128 w = (u128)v * range;
        if ((uint64_t)w <= zone) return (uint64_t)(w >> 64);
    }
}

static inline u128 prng(const uint8_t seed[32]) {                       // field.rs:264
    ChaCha20Rng g(seed);
    return uniform_u128(g, P);
}
static inline std::vector<u128> prng_vector(const uint8_t seed[32], size_t length) {   // field.rs:271
    ChaCha20Rng g(seed);
    std::vector<u128> r(length);
    for (size_t i = 0; i < length; i++) r[i] = uniform_u128(g, P);
    return r;
}

}  // namespace orc
