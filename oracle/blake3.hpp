// ORACLE (test infrastructure, NOT product code) -- BLAKE3 (unkeyed hash, 32-byte output).
//
// The reference calls the third-party `blake3` crate (^0.3.5, Cargo.toml:21) at
// /root/reference/src/crypto/hash.rs:205-209 (`blake3::hash(&values)`); the crate's sources are NOT
// under /root/reference, so this file restates the published BLAKE3 algorithm (BLAKE3 spec, sections
// 2.1-2.6: IV, message permutation, G function, 7 rounds, chunk chaining values, binary tree of
// parents, ROOT flag). It is pinned in this container against the official C implementation exported
// by /opt/rocm/lib/llvm/lib/libclang-cpp.so (llvm_blake3_hasher_*), see tests/golden/make_blake3_golden.py;
// the generated digests are committed under tests/golden/.
#pragma once
#include <cstdint>
#include <cstddef>
#include <cstring>

namespace orc {

static const uint32_t B3_IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au,
                                  0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
static const uint8_t B3_PERM[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};
enum { B3_CHUNK_START = 1, B3_CHUNK_END = 2, B3_PARENT = 4, B3_ROOT = 8 };

static inline uint32_t b3_rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
static inline void b3_g(uint32_t* s, int a, int b, int c, int d, uint32_t mx, uint32_t my) {
    s[a] = s[a] + s[b] + mx; s[d] = b3_rotr(s[d] ^ s[a], 16);
    s[c] = s[c] + s[d];      s[b] = b3_rotr(s[b] ^ s[c], 12);
    s[a] = s[a] + s[b] + my; s[d] = b3_rotr(s[d] ^ s[a], 8);
    s[c] = s[c] + s[d];      s[b] = b3_rotr(s[b] ^ s[c], 7);
}

// compression function; `out` receives the 8-word chaining value
static inline void b3_compress(const uint32_t cv[8], const uint32_t block[16], uint64_t counter,
                               uint32_t block_len, uint32_t flags, uint32_t out[8]) {
    uint32_t s[16], m[16], t[16];
    for (int i = 0; i < 8; i++) s[i] = cv[i];
    for (int i = 0; i < 4; i++) s[8 + i] = B3_IV[i];
    s[12] = (uint32_t)counter; s[13] = (uint32_t)(counter >> 32); s[14] = block_len; s[15] = flags;
    for (int i = 0; i < 16; i++) m[i] = block[i];
    for (int r = 0; r < 7; r++) {
        b3_g(s, 0, 4, 8, 12, m[0], m[1]);   b3_g(s, 1, 5, 9, 13, m[2], m[3]);
        b3_g(s, 2, 6, 10, 14, m[4], m[5]);  b3_g(s, 3, 7, 11, 15, m[6], m[7]);
        b3_g(s, 0, 5, 10, 15, m[8], m[9]);  b3_g(s, 1, 6, 11, 12, m[10], m[11]);
        b3_g(s, 2, 7, 8, 13, m[12], m[13]); b3_g(s, 3, 4, 9, 14, m[14], m[15]);
        for (int i = 0; i < 16; i++) t[i] = m[B3_PERM[i]];
        for (int i = 0; i < 16; i++) m[i] = t[i];
    }
    for (int i = 0; i < 8; i++) out[i] = s[i] ^ s[i + 8];
}

static inline void b3_load_block(const uint8_t* p, size_t len, uint32_t w[16]) {
    uint8_t buf[64];
    memset(buf, 0, 64);
    memcpy(buf, p, len);
    for (int i = 0; i < 16; i++)
        w[i] = (uint32_t)buf[4 * i] | ((uint32_t)buf[4 * i + 1] << 8) | ((uint32_t)buf[4 * i + 2] << 16) | ((uint32_t)buf[4 * i + 3] << 24);
}

// chaining value of one chunk (<= 1024 bytes); `root` applies the ROOT flag to the last block
static inline void b3_chunk_cv(const uint8_t* p, size_t len, uint64_t chunk_index, bool root, uint32_t out[8]) {
    uint32_t cv[8];
    for (int i = 0; i < 8; i++) cv[i] = B3_IV[i];
    size_t nblocks = len == 0 ? 1 : (len + 63) / 64;
    for (size_t b = 0; b < nblocks; b++) {
        size_t off = b * 64;
        size_t bl = (len - off) < 64 ? (len - off) : 64;
        uint32_t w[16];
        b3_load_block(p + off, bl, w);
        uint32_t flags = 0;
        if (b == 0) flags |= B3_CHUNK_START;
        if (b == nblocks - 1) { flags |= B3_CHUNK_END; if (root) flags |= B3_ROOT; }
        uint32_t nx[8];
        b3_compress(cv, w, chunk_index, (uint32_t)bl, flags, nx);
        for (int i = 0; i < 8; i++) cv[i] = nx[i];
    }
    for (int i = 0; i < 8; i++) out[i] = cv[i];
}

// chaining value of a subtree covering `len` bytes starting at chunk `chunk_index`
static inline void b3_subtree_cv(const uint8_t* p, size_t len, uint64_t chunk_index, bool root, uint32_t out[8]) {
    if (len <= 1024) { b3_chunk_cv(p, len, chunk_index, root, out); return; }
    // left subtree: the largest power-of-two number of chunks that leaves at least one byte on the right
    size_t chunks = (len - 1) / 1024;          // full chunks strictly before the last byte
    size_t left_chunks = 1;
    while (left_chunks * 2 <= chunks) left_chunks *= 2;
    size_t left_len = left_chunks * 1024;
    uint32_t block[16];
    b3_subtree_cv(p, left_len, chunk_index, false, block);
    b3_subtree_cv(p + left_len, len - left_len, chunk_index + left_chunks, false, block + 8);
    b3_compress(B3_IV, block, 0, 64, B3_PARENT | (root ? B3_ROOT : 0), out);
}

static inline void blake3(const uint8_t* in, size_t len, uint8_t out[32]) {   // hash.rs:205
    uint32_t cv[8];
    b3_subtree_cv(in, len, 0, true, cv);
    for (int i = 0; i < 8; i++) {
        out[4 * i] = (uint8_t)cv[i]; out[4 * i + 1] = (uint8_t)(cv[i] >> 8);
        out[4 * i + 2] = (uint8_t)(cv[i] >> 16); out[4 * i + 3] = (uint8_t)(cv[i] >> 24);
    }
}

}  // namespace orc
