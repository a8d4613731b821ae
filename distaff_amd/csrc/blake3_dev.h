// BLAKE3 compression for gfx950 (32-bit add / xor / rotate on the VALU, message schedule resolved at compile time).
// Restates the published BLAKE3 algorithm (the reference calls the third-party `blake3` crate at
// /root/reference/src/crypto/hash.rs:205-209, unkeyed, 32-byte output).  Only the shapes on the prover path are provided:
//   * one-block messages of exactly 64 bytes (Merkle node pairs, FRI rows, proof-of-work inputs): merkle.rs:278-291,
//     fri/utils.rs:16-22, utils/proof_of_work.rs:10-26;
//   * trace rows of W*16 bytes, W < 128, i.e. one or two chunks (trace_table.rs:174-185).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define B3_HD __host__ __device__ __forceinline__
#else
#define B3_HD inline
#endif

#define B3_IV0 0x6A09E667u
#define B3_IV1 0xBB67AE85u
#define B3_IV2 0x3C6EF372u
#define B3_IV3 0xA54FF53Au
#define B3_IV4 0x510E527Fu
#define B3_IV5 0x9B05688Cu
#define B3_IV6 0x1F83D9ABu
#define B3_IV7 0x5BE0CD19u
#define B3_CHUNK_START 1u
#define B3_CHUNK_END 2u
#define B3_PARENT 4u
#define B3_ROOT 8u

B3_HD uint32_t b3_rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

#define B3_G(a, b, c, d, mx, my) \
    a = a + b + (mx); d = b3_rotr(d ^ a, 16); c = c + d; b = b3_rotr(b ^ c, 12); \
    a = a + b + (my); d = b3_rotr(d ^ a, 8);  c = c + d; b = b3_rotr(b ^ c, 7);

#define B3_ROUND(m0, m1, m2, m3, m4, m5, m6, m7, m8, m9, m10, m11, m12, m13, m14, m15) \
    B3_G(s0, s4, s8, s12, m0, m1) B3_G(s1, s5, s9, s13, m2, m3) B3_G(s2, s6, s10, s14, m4, m5) B3_G(s3, s7, s11, s15, m6, m7) \
    B3_G(s0, s5, s10, s15, m8, m9) B3_G(s1, s6, s11, s12, m10, m11) B3_G(s2, s7, s8, s13, m12, m13) B3_G(s3, s4, s9, s14, m14, m15)

// cv: 8 words in/out; m: 16 message words
B3_HD void b3_compress(uint32_t* cv, const uint32_t* m, uint32_t counter_lo, uint32_t counter_hi, uint32_t block_len, uint32_t flags) {
    uint32_t s0 = cv[0], s1 = cv[1], s2 = cv[2], s3 = cv[3], s4 = cv[4], s5 = cv[5], s6 = cv[6], s7 = cv[7];
    uint32_t s8 = B3_IV0, s9 = B3_IV1, s10 = B3_IV2, s11 = B3_IV3, s12 = counter_lo, s13 = counter_hi, s14 = block_len, s15 = flags;
    // the message permutation (2,6,3,10,7,0,4,13,1,11,12,5,9,14,15,8) applied r times, written out per round
    B3_ROUND(m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], m[8], m[9], m[10], m[11], m[12], m[13], m[14], m[15])
    B3_ROUND(m[2], m[6], m[3], m[10], m[7], m[0], m[4], m[13], m[1], m[11], m[12], m[5], m[9], m[14], m[15], m[8])
    B3_ROUND(m[3], m[4], m[10], m[12], m[13], m[2], m[7], m[14], m[6], m[5], m[9], m[0], m[11], m[15], m[8], m[1])
    B3_ROUND(m[10], m[7], m[12], m[9], m[14], m[3], m[13], m[15], m[4], m[0], m[11], m[2], m[5], m[8], m[1], m[6])
    B3_ROUND(m[12], m[13], m[9], m[11], m[15], m[10], m[14], m[8], m[7], m[2], m[5], m[3], m[0], m[1], m[6], m[4])
    B3_ROUND(m[9], m[14], m[11], m[5], m[8], m[12], m[15], m[1], m[13], m[3], m[0], m[10], m[2], m[6], m[4], m[7])
    B3_ROUND(m[11], m[15], m[5], m[0], m[1], m[9], m[8], m[6], m[14], m[10], m[2], m[12], m[3], m[4], m[7], m[13])
    cv[0] = s0 ^ s8; cv[1] = s1 ^ s9; cv[2] = s2 ^ s10; cv[3] = s3 ^ s11;
    cv[4] = s4 ^ s12; cv[5] = s5 ^ s13; cv[6] = s6 ^ s14; cv[7] = s7 ^ s15;
}

B3_HD void b3_iv(uint32_t* cv) {
    cv[0] = B3_IV0; cv[1] = B3_IV1; cv[2] = B3_IV2; cv[3] = B3_IV3; cv[4] = B3_IV4; cv[5] = B3_IV5; cv[6] = B3_IV6; cv[7] = B3_IV7;
}

// digest of a message of exactly 64 bytes
B3_HD void b3_hash64(const uint32_t* m16, uint32_t* out8) {
    b3_iv(out8);
    b3_compress(out8, m16, 0, 0, 64, B3_CHUNK_START | B3_CHUNK_END | B3_ROOT);
}
