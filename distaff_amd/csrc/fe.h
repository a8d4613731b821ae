// Field arithmetic for the 128-bit STARK field of Distaff, p = 2^128 - 45*2^40 + 1, written for CDNA4 VALU:
// an element is four little-endian 32-bit limbs (one 16-byte global/LDS access), products are formed with
// 32x32+64 multiply-adds (v_mad_u64_u32) and reduced with 2^128 = 45*2^40 - 1 (mod p), i.e. a multiply of the
// high half by the 14-bit constant 45*2^8 plus a limb shift.  No 64x64 multiplies, no divisions, no MFMA.
// Semantics mirror /root/reference/src/math/field.rs (add :27, sub :33, mul :38, exp :201, inv :83): canonical
// inputs in [0, p) give canonical outputs.  The same code compiles for the host (tables are built there).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FE_HD __host__ __device__ __forceinline__
#else
#define FE_HD inline
#endif

struct __attribute__((aligned(16))) fe { uint32_t v[4]; };

#define FE_P0 0x00000001u
#define FE_P1 0xFFFFD300u
#define FE_P2 0xFFFFFFFFu
#define FE_P3 0xFFFFFFFFu
// C128 = 2^128 mod p = 45*2^40 - 1 = 0x00002CFF_FFFFFFFF
#define FE_C0 0xFFFFFFFFu
#define FE_C1 0x00002CFFu
#define FE_K  11520u            // 45 * 2^8: (hi * 45) << 40 == (hi * 11520) << 32

FE_HD fe fe_make(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { fe r; r.v[0] = a; r.v[1] = b; r.v[2] = c; r.v[3] = d; return r; }
FE_HD fe fe_zero() { return fe_make(0, 0, 0, 0); }
FE_HD fe fe_one() { return fe_make(1, 0, 0, 0); }
FE_HD fe fe_from_u64(uint64_t x) { return fe_make((uint32_t)x, (uint32_t)(x >> 32), 0, 0); }
FE_HD bool fe_is_zero(const fe& a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3]) == 0; }
FE_HD bool fe_eq(const fe& a, const fe& b) { return ((a.v[0] ^ b.v[0]) | (a.v[1] ^ b.v[1]) | (a.v[2] ^ b.v[2]) | (a.v[3] ^ b.v[3])) == 0; }

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint32_t fe_addc(uint32_t a, uint32_t b, uint32_t cin, uint32_t* cout) { return __builtin_addc(a, b, cin, cout); }
__device__ __forceinline__ uint32_t fe_subb(uint32_t a, uint32_t b, uint32_t bin, uint32_t* bout) { return __builtin_subc(a, b, bin, bout); }
#endif

FE_HD fe fe_add(const fe& a, const fe& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t c, cs;
    uint32_t s0 = fe_addc(a.v[0], b.v[0], 0, &c), s1 = fe_addc(a.v[1], b.v[1], c, &c), s2 = fe_addc(a.v[2], b.v[2], c, &c), s3 = fe_addc(a.v[3], b.v[3], c, &cs);
    uint32_t z0 = fe_addc(s0, FE_C0, 0, &c), z1 = fe_addc(s1, FE_C1, c, &c), z2 = fe_addc(s2, 0, c, &c), z3 = fe_addc(s3, 0, c, &c);
    bool ov = (cs | c) != 0;
    return fe_make(ov ? z0 : s0, ov ? z1 : s1, ov ? z2 : s2, ov ? z3 : s3);
#else
    // s = a + b; t = s + C128; a + b >= p  <=>  a + b + C128 >= 2^128
    uint64_t c = (uint64_t)a.v[0] + b.v[0];               uint32_t s0 = (uint32_t)c;
    c = (uint64_t)a.v[1] + b.v[1] + (c >> 32);            uint32_t s1 = (uint32_t)c;
    c = (uint64_t)a.v[2] + b.v[2] + (c >> 32);            uint32_t s2 = (uint32_t)c;
    c = (uint64_t)a.v[3] + b.v[3] + (c >> 32);            uint32_t s3 = (uint32_t)c;
    uint32_t cs = (uint32_t)(c >> 32);
    uint64_t d = (uint64_t)s0 + FE_C0;                    uint32_t t0 = (uint32_t)d;
    d = (uint64_t)s1 + FE_C1 + (d >> 32);                 uint32_t t1 = (uint32_t)d;
    d = (uint64_t)s2 + (d >> 32);                         uint32_t t2 = (uint32_t)d;
    d = (uint64_t)s3 + (d >> 32);                         uint32_t t3 = (uint32_t)d;
    bool over = (cs | (uint32_t)(d >> 32)) != 0;
    return over ? fe_make(t0, t1, t2, t3) : fe_make(s0, s1, s2, s3);
#endif
}

FE_HD fe fe_sub(const fe& a, const fe& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t bw;
    uint32_t d0 = fe_subb(a.v[0], b.v[0], 0, &bw), d1 = fe_subb(a.v[1], b.v[1], bw, &bw), d2 = fe_subb(a.v[2], b.v[2], bw, &bw), d3 = fe_subb(a.v[3], b.v[3], bw, &bw);
    uint32_t m = 0u - bw;                              // all ones when a < b: add p == subtract C128 (mod 2^128)
    uint32_t e0 = fe_subb(d0, m, 0, &bw), e1 = fe_subb(d1, m & FE_C1, bw, &bw), e2 = fe_subb(d2, 0, bw, &bw), e3 = fe_subb(d3, 0, bw, &bw);
    return fe_make(e0, e1, e2, e3);
#else
    // d = a - b; on borrow add p, i.e. subtract C128 modulo 2^128
    int64_t c = (int64_t)(uint64_t)a.v[0] - b.v[0];                 uint32_t d0 = (uint32_t)c;
    c = (int64_t)(uint64_t)a.v[1] - b.v[1] + (c >> 32);             uint32_t d1 = (uint32_t)c;
    c = (int64_t)(uint64_t)a.v[2] - b.v[2] + (c >> 32);             uint32_t d2 = (uint32_t)c;
    c = (int64_t)(uint64_t)a.v[3] - b.v[3] + (c >> 32);             uint32_t d3 = (uint32_t)c;
    bool borrow = (c >> 32) != 0;
    int64_t e = (int64_t)(uint64_t)d0 - FE_C0;                      uint32_t e0 = (uint32_t)e;
    e = (int64_t)(uint64_t)d1 - FE_C1 + (e >> 32);                  uint32_t e1 = (uint32_t)e;
    e = (int64_t)(uint64_t)d2 + (e >> 32);                          uint32_t e2 = (uint32_t)e;
    e = (int64_t)(uint64_t)d3 + (e >> 32);                          uint32_t e3 = (uint32_t)e;
    return borrow ? fe_make(e0, e1, e2, e3) : fe_make(d0, d1, d2, d3);
#endif
}

FE_HD fe fe_neg(const fe& a) { return fe_sub(fe_zero(), a); }
FE_HD fe fe_double(const fe& a) { return fe_add(a, a); }

// reduce an 8-limb product t (< p^2) to a canonical element
FE_HD fe fe_reduce8(const uint32_t* t) {
    // fold 1: v = lo + ((hi * K) << 32) - hi   (6 limbs, v < 2^174, never negative because hi*K*2^32 >= hi)
    uint64_t c = (uint64_t)t[4] * FE_K;                 uint32_t m0 = (uint32_t)c;
    c = (uint64_t)t[5] * FE_K + (c >> 32);              uint32_t m1 = (uint32_t)c;
    c = (uint64_t)t[6] * FE_K + (c >> 32);              uint32_t m2 = (uint32_t)c;
    c = (uint64_t)t[7] * FE_K + (c >> 32);              uint32_t m3 = (uint32_t)c;
    uint32_t m4 = (uint32_t)(c >> 32);                  // < 2^14
    // u = lo + (m << 32)
    uint32_t u0 = t[0];
    c = (uint64_t)t[1] + m0;                            uint32_t u1 = (uint32_t)c;
    c = (uint64_t)t[2] + m1 + (c >> 32);                uint32_t u2 = (uint32_t)c;
    c = (uint64_t)t[3] + m2 + (c >> 32);                uint32_t u3 = (uint32_t)c;
    c = (uint64_t)m3 + (c >> 32);                       uint32_t u4 = (uint32_t)c;
    uint32_t u5 = m4 + (uint32_t)(c >> 32);
    // v = u - hi
    int64_t b = (int64_t)(uint64_t)u0 - t[4];                       uint32_t v0 = (uint32_t)b;
    b = (int64_t)(uint64_t)u1 - t[5] + (b >> 32);                   uint32_t v1 = (uint32_t)b;
    b = (int64_t)(uint64_t)u2 - t[6] + (b >> 32);                   uint32_t v2 = (uint32_t)b;
    b = (int64_t)(uint64_t)u3 - t[7] + (b >> 32);                   uint32_t v3 = (uint32_t)b;
    b = (int64_t)(uint64_t)u4 + (b >> 32);                          uint32_t v4 = (uint32_t)b;
    uint32_t v5 = u5 + (uint32_t)(b >> 32);
    // fold 2: y = v_lo + ((vh * K) << 32) - vh, vh = v5:v4 < 2^46, vh*K < 2^60
    uint64_t vh = ((uint64_t)v5 << 32) | v4;
    uint64_t w = vh * FE_K;
    uint32_t w0 = (uint32_t)w, w1 = (uint32_t)(w >> 32);
    uint32_t y0 = v0;
    c = (uint64_t)v1 + w0;                              uint32_t y1 = (uint32_t)c;
    c = (uint64_t)v2 + w1 + (c >> 32);                  uint32_t y2 = (uint32_t)c;
    c = (uint64_t)v3 + (c >> 32);                       uint32_t y3 = (uint32_t)c;
    uint32_t y4 = (uint32_t)(c >> 32);                  // 0 or 1
    b = (int64_t)(uint64_t)y0 - v4;                                 y0 = (uint32_t)b;
    b = (int64_t)(uint64_t)y1 - v5 + (b >> 32);                     y1 = (uint32_t)b;
    b = (int64_t)(uint64_t)y2 + (b >> 32);                          y2 = (uint32_t)b;
    b = (int64_t)(uint64_t)y3 + (b >> 32);                          y3 = (uint32_t)b;
    y4 += (uint32_t)(b >> 32);                          // borrow propagates into the single overflow bit
    // fold 3: if y4 == 1 then y = y_lo + C128 (y_lo < 2^93 in that case: no further carry)
    uint32_t k0 = y4 ? FE_C0 : 0u, k1 = y4 ? FE_C1 : 0u;
    c = (uint64_t)y0 + k0;                              y0 = (uint32_t)c;
    c = (uint64_t)y1 + k1 + (c >> 32);                  y1 = (uint32_t)c;
    c = (uint64_t)y2 + (c >> 32);                       y2 = (uint32_t)c;
    c = (uint64_t)y3 + (c >> 32);                       y3 = (uint32_t)c;
    // final conditional subtraction of p: y >= p <=> y + C128 overflows 2^128
    uint64_t d = (uint64_t)y0 + FE_C0;                  uint32_t z0 = (uint32_t)d;
    d = (uint64_t)y1 + FE_C1 + (d >> 32);               uint32_t z1 = (uint32_t)d;
    d = (uint64_t)y2 + (d >> 32);                       uint32_t z2 = (uint32_t)d;
    d = (uint64_t)y3 + (d >> 32);                       uint32_t z3 = (uint32_t)d;
    bool ge = (d >> 32) != 0;
    return ge ? fe_make(z0, z1, z2, z3) : fe_make(y0, y1, y2, y3);
}

FE_HD fe fe_mul_portable(const fe& a, const fe& b) {
    uint32_t t[8];
    uint64_t c;
    // row 0
    c = (uint64_t)a.v[0] * b.v[0];                              t[0] = (uint32_t)c;
    c = (uint64_t)a.v[0] * b.v[1] + (c >> 32);                  t[1] = (uint32_t)c;
    c = (uint64_t)a.v[0] * b.v[2] + (c >> 32);                  t[2] = (uint32_t)c;
    c = (uint64_t)a.v[0] * b.v[3] + (c >> 32);                  t[3] = (uint32_t)c;
    t[4] = (uint32_t)(c >> 32);
#pragma unroll
    for (int i = 1; i < 4; i++) {
        c = (uint64_t)a.v[i] * b.v[0] + t[i];                       t[i] = (uint32_t)c;
        c = (uint64_t)a.v[i] * b.v[1] + t[i + 1] + (c >> 32);       t[i + 1] = (uint32_t)c;
        c = (uint64_t)a.v[i] * b.v[2] + t[i + 2] + (c >> 32);       t[i + 2] = (uint32_t)c;
        c = (uint64_t)a.v[i] * b.v[3] + t[i + 3] + (c >> 32);       t[i + 3] = (uint32_t)c;
        t[i + 4] = (uint32_t)(c >> 32);
    }
    return fe_reduce8(t);
}

#if defined(__HIP_DEVICE_COMPILE__)
// ---- gfx950 formulation --------------------------------------------------------------------------------------------------
// 16 v_mad_u64_u32 for the 256-bit product: partial products with i + j even / odd accumulate in separate 64-bit windows
// (E0..E3 at limbs 0,2,4,6 and O0..O2 at limbs 1,3,5), the carry-out of each accumulating mad is counted with one
// v_addc_co_u32, and the windows are merged with two carry chains.  The reduction multiplies the four high limbs by
// K = 45*2^8 with independent mads (no carry chain between them) and folds with add/sub-with-carry chains.
__device__ __forceinline__ void fe_mac_c(uint64_t& acc, uint32_t& cnt, uint32_t a, uint32_t b) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(cnt) : "v"(a), "v"(b) : "vcc");
}
// first accumulation into a window: the carry count starts as the carry-out itself (no zero-initialised register)
__device__ __forceinline__ void fe_mac_c0(uint64_t& acc, uint32_t& cnt, uint32_t a, uint32_t b) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, 0, vcc" : "+v"(acc), "=v"(cnt) : "v"(a), "v"(b) : "vcc");
}
__device__ __forceinline__ fe fe_mul_gfx950(const fe& a, const fe& b) {
#define FE_LO(x) ((uint32_t)(x))
#define FE_HI(x) ((uint32_t)((x) >> 32))
    uint64_t E0 = (uint64_t)a.v[0] * b.v[0];
    uint64_t E1 = (uint64_t)a.v[0] * b.v[2]; uint32_t ce1; fe_mac_c0(E1, ce1, a.v[1], b.v[1]); fe_mac_c(E1, ce1, a.v[2], b.v[0]);
    uint64_t E2 = (uint64_t)a.v[1] * b.v[3]; uint32_t ce2; fe_mac_c0(E2, ce2, a.v[2], b.v[2]); fe_mac_c(E2, ce2, a.v[3], b.v[1]);
    uint64_t E3 = (uint64_t)a.v[3] * b.v[3];
    uint64_t O0 = (uint64_t)a.v[0] * b.v[1]; uint32_t co0; fe_mac_c0(O0, co0, a.v[1], b.v[0]);
    uint64_t O1 = (uint64_t)a.v[0] * b.v[3]; uint32_t co1; fe_mac_c0(O1, co1, a.v[1], b.v[2]); fe_mac_c(O1, co1, a.v[2], b.v[1]); fe_mac_c(O1, co1, a.v[3], b.v[0]);
    uint64_t O2 = (uint64_t)a.v[2] * b.v[3]; uint32_t co2; fe_mac_c0(O2, co2, a.v[3], b.v[2]);
    uint32_t t0, t1, t2, t3, t4, t5, t6, t7, c, bw;
    t0 = FE_LO(E0);
    t1 = fe_addc(FE_HI(E0), FE_LO(O0), 0, &c);
    t2 = fe_addc(FE_LO(E1), FE_HI(O0), c, &c);
    t3 = fe_addc(FE_HI(E1), FE_LO(O1), c, &c);
    t4 = fe_addc(FE_LO(E2), FE_HI(O1), c, &c);
    t5 = fe_addc(FE_HI(E2), FE_LO(O2), c, &c);
    t6 = fe_addc(FE_LO(E3), FE_HI(O2), c, &c);
    t7 = FE_HI(E3) + c;
    t3 = fe_addc(t3, co0, 0, &c);
    t4 = fe_addc(t4, ce1, c, &c);
    t5 = fe_addc(t5, co1, c, &c);
    t6 = fe_addc(t6, ce2, c, &c);
    t7 = t7 + co2 + c;
    // fold 1: v = lo + ((hi * K) << 32) - hi
    uint64_t P0 = (uint64_t)t4 * FE_K, P1 = (uint64_t)t5 * FE_K, P2 = (uint64_t)t6 * FE_K, P3 = (uint64_t)t7 * FE_K;
    uint32_t u1, u2, u3, u4, u5;
    u1 = fe_addc(t1, FE_LO(P0), 0, &c);
    u2 = fe_addc(t2, FE_LO(P1), c, &c);
    u3 = fe_addc(t3, FE_LO(P2), c, &c);
    u4 = fe_addc(FE_LO(P3), 0, c, &c);
    u5 = c;
    u2 = fe_addc(u2, FE_HI(P0), 0, &c);
    u3 = fe_addc(u3, FE_HI(P1), c, &c);
    u4 = fe_addc(u4, FE_HI(P2), c, &c);
    u5 = u5 + FE_HI(P3) + c;
    uint32_t v0, v1, v2, v3, v4, v5;
    v0 = fe_subb(t0, t4, 0, &bw);
    v1 = fe_subb(u1, t5, bw, &bw);
    v2 = fe_subb(u2, t6, bw, &bw);
    v3 = fe_subb(u3, t7, bw, &bw);
    v4 = fe_subb(u4, 0, bw, &bw);
    v5 = u5 - bw;
    // fold 2: y = v_lo + (((v5:v4) * K) << 32) - (v5:v4)
    uint64_t w = (uint64_t)v4 * FE_K;
    uint32_t w0 = FE_LO(w), w1 = FE_HI(w) + v5 * FE_K;
    uint32_t y0, y1, y2, y3, y4;
    y1 = fe_addc(v1, w0, 0, &c);
    y2 = fe_addc(v2, w1, c, &c);
    y3 = fe_addc(v3, 0, c, &c);
    y4 = c;
    y0 = fe_subb(v0, v4, 0, &bw);
    y1 = fe_subb(y1, v5, bw, &bw);
    y2 = fe_subb(y2, 0, bw, &bw);
    y3 = fe_subb(y3, 0, bw, &bw);
    y4 = y4 - bw;
    // fold 3 and canonical form in one step.  The value is Y = y + y4 * 2^128 with y4 in {0, 1} and Y < 2p.  z = y + C128 (mod 2^128)
    // is Y - p whenever Y >= p, and Y >= p <=> y4 = 1 or the addition carries out (when y4 = 1, y < p - C128, so no carry).
    uint32_t z0, z1, z2, z3;
    z0 = fe_addc(y0, FE_C0, 0, &c);
    z1 = fe_addc(y1, FE_C1, c, &c);
    z2 = fe_addc(y2, 0, c, &c);
    z3 = fe_addc(y3, 0, c, &c);
    c |= y4;
#undef FE_LO
#undef FE_HI
    return c ? fe_make(z0, z1, z2, z3) : fe_make(y0, y1, y2, y3);
}

// ---- sums of products with ONE reduction ----------------------------------------------------------------------------------------
// A dot product sum_i a_i * b_i accumulates the 256-bit partial products in the same even/odd 64-bit windows as fe_mul (every
// accumulating mad counts its carry-out) and folds once at the end: 32 instructions per term + one reduction instead of a full
// multiplication (89) and a modular addition (14) per term.  Up to 64 terms (the overflow limb stays below 2^7).
struct fe_acc {
    uint64_t E0, E1, E2, E3, O0, O1, O2;              // windows at limbs 0, 2, 4, 6 and 1, 3, 5
    uint32_t cE0, cE1, cE2, cE3, cO0, cO1, cO2;       // 2^64 overflows of each window
};
__device__ __forceinline__ void fe_acc_zero(fe_acc& A) {
    A.E0 = A.E1 = A.E2 = A.E3 = A.O0 = A.O1 = A.O2 = 0;
    A.cE0 = A.cE1 = A.cE2 = A.cE3 = A.cO0 = A.cO1 = A.cO2 = 0;
}
__device__ __forceinline__ void fe_acc_mac(fe_acc& A, const fe& a, const fe& b) {
    fe_mac_c(A.E0, A.cE0, a.v[0], b.v[0]);
    fe_mac_c(A.E1, A.cE1, a.v[0], b.v[2]); fe_mac_c(A.E1, A.cE1, a.v[1], b.v[1]); fe_mac_c(A.E1, A.cE1, a.v[2], b.v[0]);
    fe_mac_c(A.E2, A.cE2, a.v[1], b.v[3]); fe_mac_c(A.E2, A.cE2, a.v[2], b.v[2]); fe_mac_c(A.E2, A.cE2, a.v[3], b.v[1]);
    fe_mac_c(A.E3, A.cE3, a.v[3], b.v[3]);
    fe_mac_c(A.O0, A.cO0, a.v[0], b.v[1]); fe_mac_c(A.O0, A.cO0, a.v[1], b.v[0]);
    fe_mac_c(A.O1, A.cO1, a.v[0], b.v[3]); fe_mac_c(A.O1, A.cO1, a.v[1], b.v[2]); fe_mac_c(A.O1, A.cO1, a.v[2], b.v[1]); fe_mac_c(A.O1, A.cO1, a.v[3], b.v[0]);
    fe_mac_c(A.O2, A.cO2, a.v[2], b.v[3]); fe_mac_c(A.O2, A.cO2, a.v[3], b.v[2]);
}
// adds a field element (weight 1) to the sum
__device__ __forceinline__ void fe_acc_add(fe_acc& A, const fe& a) {
    fe_mac_c(A.E0, A.cE0, a.v[0], 1u); fe_mac_c(A.O0, A.cO0, a.v[1], 1u); fe_mac_c(A.E1, A.cE1, a.v[2], 1u); fe_mac_c(A.O1, A.cO1, a.v[3], 1u);
}
__device__ __forceinline__ fe fe_acc_reduce(const fe_acc& A) {
#define FE_LO(x) ((uint32_t)(x))
#define FE_HI(x) ((uint32_t)((x) >> 32))
    uint32_t t0, t1, t2, t3, t4, t5, t6, t7, t8, c, bw;
    t0 = FE_LO(A.E0);
    t1 = fe_addc(FE_HI(A.E0), FE_LO(A.O0), 0, &c);
    t2 = fe_addc(FE_LO(A.E1), FE_HI(A.O0), c, &c);
    t3 = fe_addc(FE_HI(A.E1), FE_LO(A.O1), c, &c);
    t4 = fe_addc(FE_LO(A.E2), FE_HI(A.O1), c, &c);
    t5 = fe_addc(FE_HI(A.E2), FE_LO(A.O2), c, &c);
    t6 = fe_addc(FE_LO(A.E3), FE_HI(A.O2), c, &c);
    t7 = fe_addc(FE_HI(A.E3), 0, c, &c);
    t8 = c;
    t2 = fe_addc(t2, A.cE0, 0, &c);
    t3 = fe_addc(t3, A.cO0, c, &c);
    t4 = fe_addc(t4, A.cE1, c, &c);
    t5 = fe_addc(t5, A.cO1, c, &c);
    t6 = fe_addc(t6, A.cE2, c, &c);
    t7 = fe_addc(t7, A.cO2, c, &c);
    t8 = t8 + A.cE3 + c;
    // fold 1: v = lo + ((hi * K) << 32) - hi, hi = t8:t7:t6:t5:t4 < 2^135
    uint64_t P0 = (uint64_t)t4 * FE_K, P1 = (uint64_t)t5 * FE_K, P2 = (uint64_t)t6 * FE_K, P3 = (uint64_t)t7 * FE_K, P4 = (uint64_t)t8 * FE_K;
    uint32_t u1, u2, u3, u4, u5;
    u1 = fe_addc(t1, FE_LO(P0), 0, &c);
    u2 = fe_addc(t2, FE_LO(P1), c, &c);
    u3 = fe_addc(t3, FE_LO(P2), c, &c);
    u4 = fe_addc(FE_LO(P3), 0, c, &c);
    u5 = FE_LO(P4) + c;                                  // P4 < 2^21: no carry out
    u2 = fe_addc(u2, FE_HI(P0), 0, &c);
    u3 = fe_addc(u3, FE_HI(P1), c, &c);
    u4 = fe_addc(u4, FE_HI(P2), c, &c);
    u5 = u5 + FE_HI(P3) + c;                             // < 2^22 + 2^14
    uint32_t v0, v1, v2, v3, v4, v5;
    v0 = fe_subb(t0, t4, 0, &bw);
    v1 = fe_subb(u1, t5, bw, &bw);
    v2 = fe_subb(u2, t6, bw, &bw);
    v3 = fe_subb(u3, t7, bw, &bw);
    v4 = fe_subb(u4, t8, bw, &bw);
    v5 = u5 - bw;
    // fold 2: y = v_lo + ((V * K) << 32) - V, V = v5:v4 < 2^54, V * K = w0 + x0 * 2^32 + x1 * 2^64
    uint64_t w = (uint64_t)v4 * FE_K;
    uint64_t x = (uint64_t)v5 * FE_K + FE_HI(w);
    uint32_t y0, y1, y2, y3, y4;
    y1 = fe_addc(v1, FE_LO(w), 0, &c);
    y2 = fe_addc(v2, FE_LO(x), c, &c);
    y3 = fe_addc(v3, FE_HI(x), c, &c);
    y4 = c;
    y0 = fe_subb(v0, v4, 0, &bw);
    y1 = fe_subb(y1, v5, bw, &bw);
    y2 = fe_subb(y2, 0, bw, &bw);
    y3 = fe_subb(y3, 0, bw, &bw);
    y4 = y4 - bw;
    // fold 3 + canonical form (see fe_mul_gfx950)
    uint32_t z0, z1, z2, z3;
    z0 = fe_addc(y0, FE_C0, 0, &c);
    z1 = fe_addc(y1, FE_C1, c, &c);
    z2 = fe_addc(y2, 0, c, &c);
    z3 = fe_addc(y3, 0, c, &c);
    c |= y4;
#undef FE_LO
#undef FE_HI
    return c ? fe_make(z0, z1, z2, z3) : fe_make(y0, y1, y2, y3);
}
#else
// host pass of the same translation units: same interface, plain modular arithmetic (never on a hot path)
struct fe_acc { fe sum; };
FE_HD void fe_acc_zero(fe_acc& A) { A.sum = fe_zero(); }
FE_HD void fe_acc_mac(fe_acc& A, const fe& a, const fe& b) { A.sum = fe_add(A.sum, fe_mul_portable(a, b)); }
FE_HD void fe_acc_add(fe_acc& A, const fe& a) { A.sum = fe_add(A.sum, a); }
FE_HD fe fe_acc_reduce(const fe_acc& A) { return A.sum; }
#endif

FE_HD fe fe_mul(const fe& a, const fe& b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FE_PORTABLE_MUL)
    return fe_mul_gfx950(a, b);
#else
    return fe_mul_portable(a, b);
#endif
}

FE_HD fe fe_sqr(const fe& a) { return fe_mul(a, a); }

// multiplication by a small constant (< 2^32)
FE_HD fe fe_mul_small(const fe& a, uint32_t k) {
    uint32_t t[8];
    uint64_t c = (uint64_t)a.v[0] * k;                  t[0] = (uint32_t)c;
    c = (uint64_t)a.v[1] * k + (c >> 32);               t[1] = (uint32_t)c;
    c = (uint64_t)a.v[2] * k + (c >> 32);               t[2] = (uint32_t)c;
    c = (uint64_t)a.v[3] * k + (c >> 32);               t[3] = (uint32_t)c;
    t[4] = (uint32_t)(c >> 32); t[5] = 0; t[6] = 0; t[7] = 0;
    return fe_reduce8(t);
}

FE_HD fe fe_cube(const fe& a) { return fe_mul(fe_sqr(a), a); }

// b^e for a 128-bit exponent given as limbs (0^e = 0, b^0 = 1 for b != 0, as field.rs:201-203)
FE_HD fe fe_pow(fe b, const fe& e) {
    if (fe_is_zero(b)) return fe_zero();
    fe r = fe_one();
    for (int l = 0; l < 4; l++) {
        uint32_t w = e.v[l];
        bool rest = false;
        for (int k = l + 1; k < 4; k++) rest |= e.v[k] != 0;
        for (int i = 0; i < 32; i++) {
            if (w & 1) r = fe_mul(r, b);
            w >>= 1;
            if (w == 0 && !rest) break;
            b = fe_sqr(b);
        }
        if (!rest) break;
    }
    return r;
}
FE_HD fe fe_pow_u64(const fe& b, uint64_t e) { return fe_pow(b, fe_from_u64(e)); }
FE_HD fe fe_inv(const fe& a) {          // a^(p-2); inv(0) = 0 (field.rs:84)
    return fe_pow(a, fe_make(0xFFFFFFFFu, 0xFFFFD2FFu, 0xFFFFFFFFu, 0xFFFFFFFFu));   // exponent p - 2
}
