// Host-side generator of Distaff execution traces for straight-line (Span / Group) programs -- the input side of the prover
// path.  It mirrors what processor::execute (/root/reference/src/processor/mod.rs:23-143) records for such
// programs: the decoder registers (processor/decoder/mod.rs: op counter, Rescue hash accumulator, op bits written one row
// behind, context stack) and the user stack registers (processor/stack/mod.rs), including the VOID padding of
// finalize_trace (decoder/mod.rs:253-270, stack/mod.rs:132-141).  The Fibonacci example (src/examples/fibonacci.rs:32-47) is
// built on top of it.  Field arithmetic on the host uses 64x64->128 multiplies (unsigned __int128).
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>
#include "host_util.h"
#include "rescue_constants.h"

namespace dsth {

// ---- host field (u128) -------------------------------------------------------------------------------------------------------
// p = 2^128 - C with C = 45 * 2^40 - 1 (46 bits): a 256-bit product hi:lo is folded as lo + hi * C, twice.
static const uint64_t HF_C = ((uint64_t)45 << 40) - 1;      // 2^128 mod p
inline u128 hf_add(u128 a, u128 b) { u128 z = FIELD_P - b; return a < z ? FIELD_P - z + a : a - z; }
inline u128 hf_sub(u128 a, u128 b) { return a < b ? FIELD_P - b + a : a - b; }
// hf_fold_lazy: any hi:lo -> a value < 2^128 congruent to it (not necessarily < p); hf_fold: the canonical value
inline u128 hf_fold_lazy(u128 hi, u128 lo) {
    u128 t0 = (u128)(uint64_t)hi * HF_C, t1 = (u128)(uint64_t)(hi >> 64) * HF_C;       // hi * C = t0 + t1 * 2^64 < 2^174
    u128 s = lo + t0;
    uint64_t top = (uint64_t)(t1 >> 64) + (s < lo);
    u128 s2 = s + (t1 << 64);
    top += s2 < s;                                                                       // < 2^47: what went past 2^128
    u128 f = (u128)top * HF_C;                                                           // < 2^93
    u128 r = s2 + f;
    if (r < s2) r += HF_C;                                                               // wrapped once more: r < 2^93 before the add
    return r;
}
inline u128 hf_fold(u128 hi, u128 lo) { u128 r = hf_fold_lazy(hi, lo); return r >= FIELD_P ? r - FIELD_P : r; }
template <bool LAZY = false>
inline u128 hf_mul(u128 a, u128 b) {                                                     // LAZY: operands and result anywhere below 2^128
    uint64_t a0 = (uint64_t)a, a1 = (uint64_t)(a >> 64), b0 = (uint64_t)b, b1 = (uint64_t)(b >> 64);
    u128 p00 = (u128)a0 * b0, p01 = (u128)a0 * b1, p10 = (u128)a1 * b0, p11 = (u128)a1 * b1;
    u128 mid = (p00 >> 64) + (uint64_t)p01 + (uint64_t)p10;
    u128 hi = p11 + (p01 >> 64) + (p10 >> 64) + (mid >> 64), lo = (u128)(uint64_t)p00 | (mid << 64);
    return LAZY ? hf_fold_lazy(hi, lo) : hf_fold(hi, lo);
}
template <bool LAZY = false>
inline u128 hf_sqr(u128 a) {                                                             // three multiplies instead of four
    uint64_t a0 = (uint64_t)a, a1 = (uint64_t)(a >> 64);
    u128 p00 = (u128)a0 * a0, p01 = (u128)a0 * a1, p11 = (u128)a1 * a1;
    u128 mid = (p00 >> 64) + 2 * (u128)(uint64_t)p01;
    u128 hi = p11 + 2 * (p01 >> 64) + (mid >> 64), lo = (u128)(uint64_t)p00 | (mid << 64);
    return LAZY ? hf_fold_lazy(hi, lo) : hf_fold(hi, lo);
}
inline u128 hf_pow(u128 b, u128 e) { if (!b) return 0; u128 r = 1; while (e) { if (e & 1) r = hf_mul(r, b); e >>= 1; b = hf_sqr(b); } return r; }
inline u128 limbs(const uint32_t v[4]) { return (u128)v[0] | ((u128)v[1] << 32) | ((u128)v[2] << 64) | ((u128)v[3] << 96); }
static const u128 HF_INV_ALPHA = (((u128)0xAAAAAAAAAAAAAAAAull) << 64) | 0xAAAA8CAAAAAAAAABull;     // utils/sponge.rs:70

// x -> x^INV_ALPHA (the inverse of cubing) on the four state elements at once.  The steps of a trace form one dependency chain through
// this function, so its latency is the cost of the generator: the four lanes advance in lock-step (four independent multiplies in flight)
// along an addition chain for the exponent's bit pattern -- INV_ALPHA = (10)^40 10001100 (10)^16 10101011 in binary (128 bits): with
// a_k = x^((10)^k), a_2k = a_k^(4^k) * a_k gives a_16 and a_40 in 6 multiplies; 127 squarings + 12 multiplies per lane against the
// 127 + 64 of square-and-multiply.  (Round 5's chain ended in (10)^20 10101011 -- a 136-bit exponent that happens to be congruent to
// INV_ALPHA modulo p - 1, so its results were the same; tests/test_host_logic.py now asserts the exponent the chain implements.)
struct Lanes { u128 v[4]; };
#if defined(HF_COUNT_OPS)          // tests/hostfield: the chain's operation counts pin the exponent it implements (127 squarings: below 2^128)
static int hf_count_sqr = 0, hf_count_mul = 0;
#define HF_COUNT(c, k) ((c) += (k))
#else
#define HF_COUNT(c, k) ((void)0)
#endif
inline void l_sqr(Lanes& r, int n) { HF_COUNT(hf_count_sqr, n); for (int k = 0; k < n; k++) for (int l = 0; l < 4; l++) r.v[l] = hf_sqr<true>(r.v[l]); }
inline void l_mul(Lanes& r, const Lanes& b) { HF_COUNT(hf_count_mul, 1); for (int l = 0; l < 4; l++) r.v[l] = hf_mul<true>(r.v[l], b.v[l]); }
inline void inv_alpha4(u128 s[4]) {
    Lanes x, a1, a2, a4, a8, a16, r;
    for (int l = 0; l < 4; l++) x.v[l] = s[l];
    a1 = x;    l_sqr(a1, 1);                             // x^(10b)
    a2 = a1;   l_sqr(a2, 2);   l_mul(a2, a1);            // (10)^2
    a4 = a2;   l_sqr(a4, 4);   l_mul(a4, a2);            // (10)^4
    a8 = a4;   l_sqr(a8, 8);   l_mul(a8, a4);            // (10)^8
    a16 = a8;  l_sqr(a16, 16); l_mul(a16, a8);           // (10)^16
    r = a16;   l_sqr(r, 32);   l_mul(r, a16);            // (10)^32
    l_sqr(r, 16); l_mul(r, a8);                          // (10)^40
    l_sqr(r, 1); l_mul(r, x);                            // 1
    l_sqr(r, 4); l_mul(r, x);                            // 0001
    l_sqr(r, 1); l_mul(r, x);                            // 1
    l_sqr(r, 2);                                         // 00
    l_sqr(r, 32); l_mul(r, a16);                         // (10)^16
    l_sqr(r, 8); l_mul(r, a4); l_mul(r, x);              // 10101011 = (10)^4 + 1
    for (int l = 0; l < 4; l++) s[l] = r.v[l] >= FIELD_P ? r.v[l] - FIELD_P : r.v[l];          // the chain ran on values below 2^128: canonical now
}

// one round of the program-hash accumulator (src/utils/sponge.rs:13-30)
struct SpongeTables {                                     // the constants as u128, converted once
    u128 ark[8][16], mds[16];
    SpongeTables() {
        for (int i = 0; i < 8; i++) for (int j = 0; j < 16; j++) ark[i][j] = limbs(SPONGE_ARK[i][j]);
        for (int i = 0; i < 16; i++) mds[i] = limbs(SPONGE_MDS[i]);
    }
};
inline const SpongeTables& sponge_tables() { static const SpongeTables t; return t; }
inline void sponge_mds(const SpongeTables& T, const u128 s[4], u128 r[4]) {
    for (int i = 0; i < 4; i++) { u128 acc = 0; for (int j = 0; j < 4; j++) acc = hf_add(acc, hf_mul(T.mds[i * 4 + j], s[j])); r[i] = acc; }
}
inline void sponge_round(u128 s[4], u128 op_code, u128 op_value, size_t step) {
    const SpongeTables& T = sponge_tables();
    size_t idx = step % 16;
    for (int i = 0; i < 4; i++) { u128 t = hf_add(s[i], T.ark[i][idx]); s[i] = hf_mul(hf_sqr(t), t); }
    u128 r[4];
    sponge_mds(T, s, r);
    r[0] = hf_add(r[0], op_code); r[1] = hf_add(r[1], op_value);
    for (int i = 0; i < 4; i++) s[i] = hf_add(r[i], T.ark[4 + i][idx]);
    inv_alpha4(s);
    sponge_mds(T, s, r);
    for (int i = 0; i < 4; i++) s[i] = r[i];
}

enum : uint8_t { VOP_BEGIN = 0x00, VOP_NOOP = 0x7F, VOP_ADD = 0x68, VOP_MUL = 0x69, VOP_DROP = 0x63, VOP_DUP2 = 0x73, VOP_SWAP = 0x78, VOP_PUSH = 0x1F };
enum : uint8_t { VFLOW_HACC = 0, VFLOW_BEGIN = 1, VFLOW_TEND = 2, VFLOW_VOID = 7 };

// Writes rows straight into caller-provided column storage of `n` rows.  Decoder layout (src/lib.rs:108-128):
// [op_counter | sponge x4 | cf bits x3 | ld bits x5 | hd bits x2 | ctx stack x ctx_depth | user stack x stack_depth].
struct TraceWriter {
    u128* cols; size_t n; size_t ctx_depth, stack_depth;
    size_t step = 0, depth;
    u128 sponge[4] = {0, 0, 0, 0};
    std::vector<u128> ctx;           // current context stack (index 0 = top), one extra slot for the outermost (always 0) context
    size_t ctx_len = 1;
    bool overflow = false;
    TraceWriter(u128* c, size_t rows, size_t ctxd, size_t stackd, const u128* inputs, size_t nin)
        : cols(c), n(rows), ctx_depth(ctxd), stack_depth(stackd), depth(nin), ctx(ctxd + 1, 0) {
        memset(cols, 0, sizeof(u128) * rows * (15 + ctxd + stackd));
        for (size_t i = 0; i < nin; i++) at(15 + ctxd + i, 0) = inputs[i];
    }
    u128& at(size_t col, size_t row) { return cols[col * n + row]; }
    bool advance(bool user_op) {
        if (step + 1 >= n) { overflow = true; return false; }
        step++;
        at(0, step) = at(0, step - 1) + (user_op ? 1 : 0);
        return true;
    }
    void set_bits(uint8_t flow, uint8_t user) {             // decoder/mod.rs:303: written for the previous row
        size_t s = step - 1;
        for (int i = 0; i < 3; i++) at(5 + i, s) = (flow >> i) & 1;
        for (int i = 0; i < 7; i++) at(8 + i, s) = (user >> i) & 1;
    }
    void write_sponge() { for (int i = 0; i < 4; i++) at(1 + i, step) = sponge[i]; }
    void write_ctx() { for (size_t i = 0; i < ctx_depth; i++) at(15 + i, step) = ctx[i]; }
    u128& st(size_t i, size_t row) { return at(15 + ctx_depth + i, row); }
    void stack_copy(size_t from) { for (size_t i = from; i < depth; i++) st(i, step) = st(i, step - 1); }
    void stack_shift_left(size_t start, size_t cnt) {       // stack/mod.rs:614
        for (size_t i = start; i < depth; i++) st(i - cnt, step) = st(i, step - 1);
        for (size_t i = depth - cnt; i < depth; i++) st(i, step) = 0;
        depth -= cnt;
    }
    void stack_shift_right(size_t cnt) {                    // stack/mod.rs:631
        depth += cnt;
        if (depth > stack_depth) { overflow = true; depth -= cnt; return; }
        for (size_t i = 0; i < depth - cnt; i++) st(i + cnt, step) = st(i, step - 1);
    }
    void stack_exec(uint8_t op, u128 value) {
        switch (op) {
            case VOP_BEGIN: case VOP_NOOP: stack_copy(0); break;
            case VOP_PUSH: stack_shift_right(1); st(0, step) = value; break;
            case VOP_DUP2: stack_shift_right(2); st(0, step) = st(0, step - 1); st(1, step) = st(1, step - 1); break;
            case VOP_DROP: stack_shift_left(1, 1); break;
            case VOP_SWAP: st(0, step) = st(1, step - 1); st(1, step) = st(0, step - 1); stack_copy(2); break;
            case VOP_ADD: st(0, step) = hf_add(st(0, step - 1), st(1, step - 1)); stack_shift_left(2, 1); break;
            case VOP_MUL: st(0, step) = hf_mul(st(0, step - 1), st(1, step - 1)); stack_shift_left(2, 1); break;
            default: overflow = true;
        }
    }
    void op(uint8_t code, u128 value = 0) {                  // decoder/mod.rs:232 decode_op + stack execute
        if (!advance(true)) return;
        write_ctx();
        set_bits(VFLOW_HACC, code);
        sponge_round(sponge, code, value, step - 1);
        write_sponge();
        stack_exec(code, value);
    }
    void start_block() {                                     // decoder/mod.rs:160
        if (!advance(false)) return;
        for (size_t i = ctx.size() - 1; i >= 1; i--) ctx[i] = ctx[i - 1];
        ctx[0] = sponge[0];
        write_ctx();
        set_bits(VFLOW_BEGIN, VOP_NOOP);
        sponge[0] = sponge[1] = sponge[2] = sponge[3] = 0;
        write_sponge();
        stack_exec(VOP_NOOP, 0);
    }
    void close_block() {                                     // processor/mod.rs:125 with sibling_hash = 0, true branch
        op(VOP_NOOP);
        if (!advance(false)) return;
        u128 context_hash = ctx[0];
        for (size_t i = 0; i + 1 < ctx.size(); i++) ctx[i] = ctx[i + 1];
        ctx[ctx.size() - 1] = 0;
        write_ctx();
        set_bits(VFLOW_TEND, VOP_NOOP);
        u128 block_hash = sponge[0];
        sponge[0] = context_hash; sponge[1] = block_hash; sponge[2] = 0; sponge[3] = 0;
        write_sponge();
        stack_exec(VOP_NOOP, 0);
        for (int i = 0; i < 14; i++) op(VOP_NOOP);
    }
    void finalize() {                                        // decoder/mod.rs:253, stack/mod.rs:132
        size_t W = 15 + ctx_depth + stack_depth;
        for (size_t r = step + 1; r < n; r++) {
            at(0, r) = at(0, step);
            for (size_t c = 1; c < 5; c++) at(c, r) = at(c, step);
            for (size_t c = 15; c < W; c++) at(c, r) = at(c, step);
        }
        for (size_t r = step; r < n; r++) for (size_t c = 5; c < 15; c++) at(c, r) = 1;
    }
};

// begin repeat.K swap dup.2 drop add end end with K = n/16 - 3 (fills the trace exactly), inputs [1, 0]
inline int fibonacci_trace(uint32_t log_n, u128* cols, u128 program_hash[2], u128* result) {
    size_t n = (size_t)1 << log_n;
    if (n < 128) return -1;
    size_t K = n / 16 - 3;
    u128 inputs[2] = {1, 0};
    TraceWriter w(cols, n, 1, 4, inputs, 2);
    w.op(VOP_BEGIN);                                         // root span: BEGIN + 14 NOOPs (assembly/mod.rs:164-167,253-269)
    for (int i = 0; i < 14; i++) w.op(VOP_NOOP);
    w.start_block();
    for (size_t it = 0; it < K; it++) {                      // merged span of K iterations joined by NOOPs (blocks/mod.rs:163-178)
        if (it) w.op(VOP_NOOP);
        w.op(VOP_SWAP); w.op(VOP_DUP2); w.op(VOP_DROP); w.op(VOP_ADD);
        for (int i = 0; i < 11; i++) w.op(VOP_NOOP);
    }
    w.close_block();
    w.close_block();
    if (w.overflow) return -2;
    w.finalize();
    program_hash[0] = w.at(1, n - 1); program_hash[1] = w.at(2, n - 1);
    *result = w.st(0, n - 1);
    return 0;
}

}  // namespace dsth
