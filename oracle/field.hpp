// ORACLE (test infrastructure, NOT product code) -- CPU restatement of the reference's field layer.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
//
// Follows /root/reference/src/math/field.rs:
//   M (modulus)            field.rs:11      G (2^40-th root of unity)  field.rs:14
//   add/sub/mul            field.rs:27,33,38
//   inv / inv_many_fill    field.rs:83,173  (zeros map to zero, field.rs:84,177,184)
//   exp                    field.rs:201     get_root_of_unity field.rs:228
//   get_power_series       field.rs:237
// Field results are unique canonical residues in [0, p), so any correct algorithm is bit-identical
// to the reference; the multiplication below reduces the 256-bit product with 2^128 = 45*2^40 - 1 (mod p)
// instead of the reference's two 128x64 partial products.
#pragma once
#include <cstdint>
#include <cstddef>
#include <vector>
#include <cassert>

namespace orc {

typedef unsigned __int128 u128;

static inline constexpr u128 make_u128(uint64_t hi, uint64_t lo) { return ((u128)hi << 64) | lo; }

// p = 2^128 - 45 * 2^40 + 1 = 340282366920938463463374557953744961537
static constexpr u128 P = make_u128(0xFFFFFFFFFFFFFFFFull, 0xFFFFD30000000001ull);
// G = 23953097886125630542083529559205016746 (root of unity of order 2^40)
static constexpr u128 G = make_u128(0x120532E7B364080Aull, 0x86B8723E1920F4AAull);
static constexpr u128 C128 = (u128)45 * ((u128)1 << 40) - 1;  // 2^128 mod p

static inline u128 add(u128 a, u128 b) {          // field.rs:27
    u128 z = P - b;
    return a < z ? P - z + a : a - z;
}
static inline u128 sub(u128 a, u128 b) {          // field.rs:33
    return a < b ? P - b + a : a - b;
}
static inline u128 neg(u128 a) { return sub(0, a); }   // field.rs:222

// 128x128 -> 256 (hi, lo)
static inline void mul_wide(u128 a, u128 b, u128& hi, u128& lo) {
    uint64_t a0 = (uint64_t)a, a1 = (uint64_t)(a >> 64);
    uint64_t b0 = (uint64_t)b, b1 = (uint64_t)(b >> 64);
    u128 p00 = (u128)a0 * b0, p01 = (u128)a0 * b1, p10 = (u128)a1 * b0, p11 = (u128)a1 * b1;
    u128 mid = (p00 >> 64) + (uint64_t)p01 + (uint64_t)p10;
    lo = (u128)(uint64_t)p00 | (mid << 64);
    hi = p11 + (p01 >> 64) + (p10 >> 64) + (mid >> 64);
}

static inline u128 reduce256(u128 hi, u128 lo) {
    // x = hi * 2^128 + lo  ==  lo + hi * C128  (mod p); iterate until the high half vanishes
    while (hi != 0) {
        u128 h2, l2;
        mul_wide(hi, C128, h2, l2);
        u128 s = lo + l2;
        hi = h2 + (s < lo ? 1 : 0);
        lo = s;
    }
    if (lo >= P) lo -= P;
    return lo;
}

static inline u128 mul(u128 a, u128 b) {          // field.rs:38
    u128 hi, lo;
    mul_wide(a, b, hi, lo);
    return reduce256(hi, lo);
}

static inline u128 exp(u128 b, u128 e) {          // field.rs:201 (0^e = 0, b^0 = 1 for b != 0)
    if (b == 0) return 0;
    u128 r = 1;
    while (e > 0) {
        if (e & 1) r = mul(r, b);
        e >>= 1;
        b = mul(b, b);
    }
    return r;
}

static inline u128 inv(u128 x) {                  // field.rs:83 (inv(0) = 0)
    if (x == 0) return 0;
    return exp(x, P - 2);
}
static inline u128 div(u128 a, u128 b) { return mul(a, inv(b)); }   // field.rs:195

static inline void inv_many_fill(const u128* values, u128* result, size_t n) {   // field.rs:173
    u128 last = 1;
    for (size_t i = 0; i < n; i++) {
        result[i] = last;
        if (values[i] != 0) last = mul(last, values[i]);
    }
    last = inv(last);
    for (size_t i = n; i-- > 0;) {
        if (values[i] == 0) result[i] = 0;
        else {
            result[i] = mul(last, result[i]);
            last = mul(last, values[i]);
        }
    }
}

static inline u128 get_root_of_unity(size_t order) {      // field.rs:228
    assert(order != 0 && (order & (order - 1)) == 0);
    int tz = __builtin_ctzll((unsigned long long)order);
    assert(tz <= 40);
    u128 p = (u128)1 << (40 - tz);
    return exp(G, p);
}

static inline std::vector<u128> get_power_series(u128 b, size_t length) {   // field.rs:237
    std::vector<u128> r(length);
    if (length == 0) return r;
    r[0] = 1;
    for (size_t i = 1; i < length; i++) r[i] = mul(r[i - 1], b);
    return r;
}

static inline u128 from_bytes(const uint8_t* b) {         // field.rs:279 (little-endian)
    u128 v = 0;
    for (int i = 15; i >= 0; i--) v = (v << 8) | b[i];
    return v;
}
static inline void to_bytes(u128 v, uint8_t* b) {
    for (int i = 0; i < 16; i++) { b[i] = (uint8_t)v; v >>= 8; }
}

}  // namespace orc
