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

// ---- carry primitives ----------------------------------------------------------------------------------------------------------
// Everything below is written over a handful of primitives with two forms.  On gfx950 each one is a single VALU instruction
// in inline asm whose carry / borrow is a LANE MASK in an SGPR pair (fe_cf): the compiler allocates the pairs, schedules the
// instructions of independent chains between each other and pads the VALU-writes-SGPR -> VALU-reads-SGPR wait states; masks are
// combined on the scalar unit (s_or / s_andn2), never materialised as 0/1 values in vector registers.  On the host (tables, the
// host-emulated test build) the same functions are plain integer arithmetic, so the limb-level dataflow of every formulation is exercised by the
// CPU tests as well (tools/felab holds the standalone CPU and GPU checks of this file).
#if defined(__HIP_DEVICE_COMPILE__)
typedef uint64_t fe_cf;
__device__ __forceinline__ uint64_t fe_madc(uint32_t a, uint32_t b, uint64_t c, fe_cf& co) {          // a * b + c, carry out of bit 64
    uint64_t d; asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(co) : "v"(a), "v"(b), "v"(c)); return d;
}
__device__ __forceinline__ uint32_t fe_cnt0(fe_cf c) { uint32_t r; asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(r) : "s"(c)); return r; }
__device__ __forceinline__ uint32_t fe_cnt(uint32_t n, fe_cf c) { uint32_t r; fe_cf o; asm("v_addc_co_u32_e64 %0, %1, 0, %2, %3" : "=v"(r), "=s"(o) : "v"(n), "s"(c)); return r; }
__device__ __forceinline__ uint32_t fe_add_co(uint32_t a, uint32_t b, fe_cf& co) { uint32_t r; asm("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(r), "=s"(co) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ uint32_t fe_addc_co(uint32_t a, uint32_t b, fe_cf ci, fe_cf& co) { uint32_t r; asm("v_addc_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(r), "=s"(co) : "v"(a), "v"(b), "s"(ci)); return r; }
__device__ __forceinline__ uint32_t fe_addc0_co(uint32_t a, fe_cf ci, fe_cf& co) { uint32_t r; asm("v_addc_co_u32_e64 %0, %1, 0, %2, %3" : "=v"(r), "=s"(co) : "v"(a), "s"(ci)); return r; }
__device__ __forceinline__ uint32_t fe_addm1_co(uint32_t a, fe_cf& co) { uint32_t r; asm("v_add_co_u32_e64 %0, %1, -1, %2" : "=v"(r), "=s"(co) : "v"(a)); return r; }      // a + 0xFFFFFFFF
__device__ __forceinline__ uint32_t fe_sub_co(uint32_t a, uint32_t b, fe_cf& bo) { uint32_t r; asm("v_sub_co_u32_e64 %0, %1, %2, %3" : "=v"(r), "=s"(bo) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ uint32_t fe_subb_co(uint32_t a, uint32_t b, fe_cf bi, fe_cf& bo) { uint32_t r; asm("v_subb_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(r), "=s"(bo) : "v"(a), "v"(b), "s"(bi)); return r; }
__device__ __forceinline__ uint32_t fe_subb0_co(uint32_t a, fe_cf bi, fe_cf& bo) { uint32_t r; asm("v_subbrev_co_u32_e64 %0, %1, 0, %2, %3" : "=v"(r), "=s"(bo) : "v"(a), "s"(bi)); return r; }   // a - 0 - borrow
__device__ __forceinline__ uint32_t fe_sel(fe_cf m, uint32_t a, uint32_t b) { uint32_t r; asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m)); return r; }   // m ? a : b
__device__ __forceinline__ uint32_t fe_sel0(fe_cf m, uint32_t a) { uint32_t r; asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(a), "s"(m)); return r; }                        // m ? a : 0
__device__ __forceinline__ uint32_t fe_selm1(fe_cf m) { uint32_t r; asm("v_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(r) : "s"(m)); return r; }                                           // m ? 0xFFFFFFFF : 0
#else
typedef uint32_t fe_cf;
FE_HD uint64_t fe_madc(uint32_t a, uint32_t b, uint64_t c, fe_cf& co) { unsigned __int128 t = (unsigned __int128)a * b + c; co = (fe_cf)(t >> 64); return (uint64_t)t; }
FE_HD uint32_t fe_cnt0(fe_cf c) { return c; }
FE_HD uint32_t fe_cnt(uint32_t n, fe_cf c) { return n + c; }
FE_HD uint32_t fe_add_co(uint32_t a, uint32_t b, fe_cf& co) { uint64_t t = (uint64_t)a + b; co = (fe_cf)(t >> 32); return (uint32_t)t; }
FE_HD uint32_t fe_addc_co(uint32_t a, uint32_t b, fe_cf ci, fe_cf& co) { uint64_t t = (uint64_t)a + b + ci; co = (fe_cf)(t >> 32); return (uint32_t)t; }
FE_HD uint32_t fe_addc0_co(uint32_t a, fe_cf ci, fe_cf& co) { return fe_addc_co(a, 0u, ci, co); }
FE_HD uint32_t fe_addm1_co(uint32_t a, fe_cf& co) { return fe_add_co(a, 0xFFFFFFFFu, co); }
FE_HD uint32_t fe_sub_co(uint32_t a, uint32_t b, fe_cf& bo) { bo = a < b; return a - b; }
FE_HD uint32_t fe_subb_co(uint32_t a, uint32_t b, fe_cf bi, fe_cf& bo) { uint64_t t = (uint64_t)a - b - bi; bo = (fe_cf)((t >> 32) & 1); return (uint32_t)t; }
FE_HD uint32_t fe_subb0_co(uint32_t a, fe_cf bi, fe_cf& bo) { return fe_subb_co(a, 0u, bi, bo); }
FE_HD uint32_t fe_sel(fe_cf m, uint32_t a, uint32_t b) { return m ? a : b; }
FE_HD uint32_t fe_sel0(fe_cf m, uint32_t a) { return m ? a : 0u; }
FE_HD uint32_t fe_selm1(fe_cf m) { return m ? 0xFFFFFFFFu : 0u; }
#endif
#define FE_LO(x) ((uint32_t)(x))
#define FE_HI(x) ((uint32_t)((x) >> 32))
#define FE_PAIR(lo, hi) ((uint64_t)(lo) | ((uint64_t)(hi) << 32))
FE_HD uint32_t fe_mad24(uint32_t a, uint32_t b, uint32_t c) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu) + c; }      // operands below 2^24: v_mad_u32_u24

// canonical addition: s = a + b, z = s - p (= s + C128 mod 2^128); z is the answer when either addition carries.  The trial subtraction
// runs one link behind the addition (statement order = issue order, see fe_mul_tw).
FE_HD fe fe_add(const fe& a, const fe& b) {
    fe_cf c, d;
    uint32_t s0 = fe_add_co(a.v[0], b.v[0], c);
    uint32_t s1 = fe_addc_co(a.v[1], b.v[1], c, c);
    uint32_t z0 = fe_addm1_co(s0, d);
    uint32_t s2 = fe_addc_co(a.v[2], b.v[2], c, c);
    uint32_t z1 = fe_addc_co(s1, FE_C1, d, d);
    uint32_t s3 = fe_addc_co(a.v[3], b.v[3], c, c);
    uint32_t z2 = fe_addc0_co(s2, d, d);
    uint32_t z3 = fe_addc0_co(s3, d, d);
    fe_cf s = c | d;
    return fe_make(fe_sel(s, z0, s0), fe_sel(s, z1, s1), fe_sel(s, z2, s2), fe_sel(s, z3, s3));
}
// A store the L2 should not keep: data that streams out of a kernel and is next read by ANOTHER kernel (the staging array between the two
// passes of a transform).  global_store_dwordx4 ... nt: the line is marked for early eviction, so the tiles a pass re-reads (four-step
// twiddles shared by the registers, coefficient tiles shared by the cosets) stay resident in the XCD's 4 MiB L2 instead of being pushed
// out by 40 MiB of output per tile group.
FE_HD void fe_store_stream(fe* p, const fe& v) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef uint32_t fe_u32x4 __attribute__((ext_vector_type(4)));
    const fe_u32x4 t = {v.v[0], v.v[1], v.v[2], v.v[3]};
    __builtin_nontemporal_store(t, reinterpret_cast<fe_u32x4*>(p));
#else
    *p = v;
#endif
}
// the matching load: data read exactly once by this kernel (global_load_dwordx4 ... nt)
FE_HD fe fe_load_stream(const fe* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef uint32_t fe_u32x4 __attribute__((ext_vector_type(4)));
    const fe_u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const fe_u32x4*>(p));
    return fe_make(t.x, t.y, t.z, t.w);
#else
    return *p;
#endif
}
// canonical subtraction: d = a - b; on borrow add p, i.e. subtract C128 modulo 2^128
FE_HD fe fe_sub(const fe& a, const fe& b) {
    fe_cf c, d;
    uint32_t d0 = fe_sub_co(a.v[0], b.v[0], c), d1 = fe_subb_co(a.v[1], b.v[1], c, c), d2 = fe_subb_co(a.v[2], b.v[2], c, c), d3 = fe_subb_co(a.v[3], b.v[3], c, c);
    uint32_t k0 = fe_selm1(c), k1 = fe_sel0(c, FE_C1);
    uint32_t e0 = fe_sub_co(d0, k0, d), e1 = fe_subb_co(d1, k1, d, d), e2 = fe_subb0_co(d2, d, d), e3 = fe_subb0_co(d3, d, d);
    return fe_make(e0, e1, e2, e3);
}
// sum and difference of one pair (every butterfly): the four carry chains alternate, so none of them waits on its own previous link
FE_HD void fe_addsub(const fe& a, const fe& b, fe& sum, fe& dif) {
    fe_cf c, d, g, h;
    uint32_t s0 = fe_add_co(a.v[0], b.v[0], c);
    uint32_t d0 = fe_sub_co(a.v[0], b.v[0], g);
    uint32_t s1 = fe_addc_co(a.v[1], b.v[1], c, c);
    uint32_t d1 = fe_subb_co(a.v[1], b.v[1], g, g);
    uint32_t z0 = fe_addm1_co(s0, d);
    uint32_t s2 = fe_addc_co(a.v[2], b.v[2], c, c);
    uint32_t d2 = fe_subb_co(a.v[2], b.v[2], g, g);
    uint32_t z1 = fe_addc_co(s1, FE_C1, d, d);
    uint32_t s3 = fe_addc_co(a.v[3], b.v[3], c, c);
    uint32_t d3 = fe_subb_co(a.v[3], b.v[3], g, g);
    uint32_t z2 = fe_addc0_co(s2, d, d);
    uint32_t k0 = fe_selm1(g), k1 = fe_sel0(g, FE_C1);
    uint32_t z3 = fe_addc0_co(s3, d, d);
    uint32_t e0 = fe_sub_co(d0, k0, h);
    fe_cf s = c | d;
    uint32_t r0 = fe_sel(s, z0, s0);
    uint32_t e1 = fe_subb_co(d1, k1, h, h);
    uint32_t r1 = fe_sel(s, z1, s1);
    uint32_t e2 = fe_subb0_co(d2, h, h);
    uint32_t r2 = fe_sel(s, z2, s2);
    uint32_t e3 = fe_subb0_co(d3, h, h);
    sum = fe_make(r0, r1, r2, fe_sel(s, z3, s3));
    dif = fe_make(e0, e1, e2, e3);
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

// ---- gfx950 formulations -----------------------------------------------------------------------------------------------------------
// Products are sums of 32x32+64 multiply-adds (v_mad_u64_u32, the widest integer multiplier of the VALU, issued at the rate of an
// ordinary 32-bit instruction) accumulated in 64-bit windows at limb positions; every accumulating mad hands its carry-out to a one-
// instruction counter.  The counters of two neighbouring windows are the INITIAL ADDEND (low word, high word) of the window two
// limbs up, so they cost no separate carry chain.  The windows are merged by one add-with-carry chain, the part above 2^128 is folded
// with 2^128 = K * 2^32 - 1 (mod p), K = 45 * 2^8, and the last conditional subtraction of p selects with a mask formed on the
// scalar unit.  VALU instructions: general multiplication 71 (21 of them multiplies), multiplication by a table entry (fe_tw) 55 (18).

// reduction of the nine-limb value t0..t8 (t8 < 2^7; the product of two 128-bit values, or a sum of up to 64 of them).  Statement order
// = issue order (see fe_mul_tw): the two add chains of fold 1 and its subtract chain run one link apart, likewise fold 2 and the trial
// subtraction of p.
FE_HD fe fe_fold9(uint32_t t0, uint32_t t1, uint32_t t2, uint32_t t3, uint32_t t4, uint32_t t5, uint32_t t6, uint32_t t7, uint32_t t8, bool has_t8) {
    fe_cf c, e, b, d;
    // fold 1: v = lo + ((hi * K) << 32) - hi
    uint64_t P0 = (uint64_t)t4 * FE_K, P1 = (uint64_t)t5 * FE_K, P2 = (uint64_t)t6 * FE_K, P3 = (uint64_t)t7 * FE_K;
    uint32_t u1 = fe_add_co(t1, FE_LO(P0), c);
    uint32_t v0 = fe_sub_co(t0, t4, b);
    uint32_t u2 = fe_addc_co(t2, FE_LO(P1), c, c);
    uint32_t v1 = fe_subb_co(u1, t5, b, b);
    uint32_t h5 = has_t8 ? fe_mad24(t8, FE_K, FE_HI(P3)) : FE_HI(P3);               // t8 * K < 2^21
    uint32_t u3 = fe_addc_co(t3, FE_LO(P2), c, c);
    u2 = fe_add_co(u2, FE_HI(P0), e);
    uint32_t u4 = fe_addc0_co(FE_LO(P3), c, c);
    u3 = fe_addc_co(u3, FE_HI(P1), e, e);
    uint32_t v2 = fe_subb_co(u2, t6, b, b);
    uint32_t u5 = fe_cnt(h5, c);
    u4 = fe_addc_co(u4, FE_HI(P2), e, e);
    uint32_t v3 = fe_subb_co(u3, t7, b, b);
    u5 = fe_cnt(u5, e);
    uint32_t v4 = has_t8 ? fe_subb_co(u4, t8, b, b) : fe_subb0_co(u4, b, b);
    // fold 2: y = v_lo + ((V * K) << 32) - V, V = v5:v4 (< 2^46, < 2^54 with t8)
    uint64_t w = (uint64_t)v4 * FE_K;
    uint32_t v5 = fe_subb0_co(u5, b, b);
    uint32_t y0 = fe_sub_co(v0, v4, d);
    uint32_t y1 = fe_add_co(v1, FE_LO(w), c), y2, y3;
    if (has_t8) {
        uint64_t x = (uint64_t)v5 * FE_K + FE_HI(w);
        y2 = fe_addc_co(v2, FE_LO(x), c, c);
        y1 = fe_subb_co(y1, v5, d, d);
        y3 = fe_addc_co(v3, FE_HI(x), c, c);
    } else {
        uint32_t m = fe_mad24(v5, FE_K, FE_HI(w));
        y2 = fe_addc_co(v2, m, c, c);
        y1 = fe_subb_co(y1, v5, d, d);
        y3 = fe_addc0_co(v3, c, c);
    }
    uint32_t z0 = fe_addm1_co(y0, e);
    y2 = fe_subb0_co(y2, d, d);
    uint32_t z1 = fe_addc_co(y1, FE_C1, e, e);
    y3 = fe_subb0_co(y3, d, d);
    uint32_t z2 = fe_addc0_co(y2, e, e);
    fe_cf y4 = c & ~d;                               // the value is non-negative: a borrow only ever cancels a carry
    uint32_t z3 = fe_addc0_co(y3, e, e);
    fe_cf s = y4 | e;                                // y = r + y4 * 2^128 < 2p: subtract p when bit 128 is set or the trial subtraction did not borrow
    return fe_make(fe_sel(s, z0, y0), fe_sel(s, z1, y1), fe_sel(s, z2, y2), fe_sel(s, z3, y3));
}

// general multiplication; the operands may be ANY 128-bit values (not only canonical ones).  Statement order = issue order (see fe_mul_tw).
FE_HD fe fe_mul_wide(const fe& a, const fe& b) {
    const uint32_t a0 = a.v[0], a1 = a.v[1], a2 = a.v[2], a3 = a.v[3], b0 = b.v[0], b1 = b.v[1], b2 = b.v[2], b3 = b.v[3];
    fe_cf k0, k1, k2, c;
    uint64_t E0 = (uint64_t)a0 * b0;
    uint64_t O0 = (uint64_t)a0 * b1;
    uint64_t E1 = (uint64_t)a0 * b2;
    uint64_t E2 = (uint64_t)a1 * b3;
    uint64_t E3 = (uint64_t)a3 * b3;
    O0 = fe_madc(a1, b0, O0, k0);
    E1 = fe_madc(a1, b1, E1, k1);
    E2 = fe_madc(a2, b2, E2, k2);
    uint32_t co0 = fe_cnt0(k0);
    E1 = fe_madc(a2, b0, E1, k0);
    uint32_t ce1 = fe_cnt0(k1);
    E2 = fe_madc(a3, b1, E2, k1);
    uint32_t ce2 = fe_cnt0(k2);
    uint32_t t0 = FE_LO(E0);
    uint32_t t1 = fe_add_co(FE_HI(E0), FE_LO(O0), c);            // the merge chain starts while the upper windows still accumulate
    ce1 = fe_cnt(ce1, k0);
    ce2 = fe_cnt(ce2, k1);
    uint64_t O1 = fe_madc(a0, b3, FE_PAIR(co0, ce1), k2);
    uint32_t t2 = fe_addc_co(FE_LO(E1), FE_HI(O0), c, c);
    O1 = fe_madc(a1, b2, O1, k0);
    uint32_t co1 = fe_cnt0(k2);
    O1 = fe_madc(a2, b1, O1, k1);
    co1 = fe_cnt(co1, k0);
    O1 = fe_madc(a3, b0, O1, k2);
    co1 = fe_cnt(co1, k1);
    uint32_t t3 = fe_addc_co(FE_HI(E1), FE_LO(O1), c, c);
    co1 = fe_cnt(co1, k2);
    uint64_t O2 = fe_madc(a2, b3, FE_PAIR(co1, ce2), k0);
    uint32_t t4 = fe_addc_co(FE_LO(E2), FE_HI(O1), c, c);
    O2 = fe_madc(a3, b2, O2, k1);
    uint32_t co2 = fe_cnt0(k0);
    uint32_t t5 = fe_addc_co(FE_HI(E2), FE_LO(O2), c, c);
    co2 = fe_cnt(co2, k1);
    uint32_t t6 = fe_addc_co(FE_LO(E3), FE_HI(O2), c, c);
    uint32_t h7 = FE_HI(E3) + co2;
    uint32_t t7 = fe_cnt(h7, c);
    return fe_fold9(t0, t1, t2, t3, t4, t5, t6, t7, 0u, false);
}

// ---- multiplication by a table entry ---------------------------------------------------------------------------------------------------
// A twiddle w is stored as the pair (P, Q) = (w, w * 2^64 mod p).  Then  a * w = (a0 + a1 X) * P + (a2 + a3 X) * Q  (X = 2^32) is a sum
// below 2^193 for ANY 128-bit a: five product columns in windows W0..W4, one merge chain, ONE fold of the 65 bits above 2^128.
struct __attribute__((aligned(16))) fe_tw { fe p, q; };
// Statement order = issue order wanted on gfx950: an instruction that reads a carry mask (SGPR pair) needs one other instruction between
// itself and the instruction that wrote the mask, or the compiler pads with s_nop.  The windows are independent until their counters
// meet, so the multiply-adds of different windows alternate and every counter update trails its multiply-add by one or two statements;
// in the tail the add chain, the subtract chain and the trial subtraction of p run one link apart.
FE_HD fe fe_mul_tw(const fe& a, const fe& P, const fe& Q) {
    const uint32_t a0 = a.v[0], a1 = a.v[1], a2 = a.v[2], a3 = a.v[3];
    fe_cf k0, k1, k2, c, b, d;
    uint64_t W0 = (uint64_t)a0 * P.v[0];
    uint64_t W1 = (uint64_t)a0 * P.v[1];
    uint64_t W3 = (uint64_t)a0 * P.v[3];
    W0 = fe_madc(a2, Q.v[0], W0, k0);
    W1 = fe_madc(a1, P.v[0], W1, k1);
    W3 = fe_madc(a1, P.v[2], W3, k2);
    uint32_t c0 = fe_cnt0(k0);
    W1 = fe_madc(a2, Q.v[1], W1, k0);
    uint32_t c1 = fe_cnt0(k1);
    W3 = fe_madc(a2, Q.v[3], W3, k1);
    uint32_t c3 = fe_cnt0(k2);
    W1 = fe_madc(a3, Q.v[0], W1, k2);
    c1 = fe_cnt(c1, k0);
    W3 = fe_madc(a3, Q.v[2], W3, k0);
    c3 = fe_cnt(c3, k1);
    c1 = fe_cnt(c1, k2);
    uint32_t T0 = FE_LO(W0);
    uint32_t T1 = fe_add_co(FE_HI(W0), FE_LO(W1), c);            // the merge chain starts while the upper windows still accumulate
    c3 = fe_cnt(c3, k0);
    uint64_t W2 = fe_madc(a0, P.v[2], FE_PAIR(c0, c1), k1);
    W2 = fe_madc(a1, P.v[1], W2, k2);
    uint32_t c2 = fe_cnt0(k1);
    W2 = fe_madc(a2, Q.v[2], W2, k0);
    c2 = fe_cnt(c2, k2);
    W2 = fe_madc(a3, Q.v[1], W2, k1);
    c2 = fe_cnt(c2, k0);
    uint32_t T2 = fe_addc_co(FE_HI(W1), FE_LO(W2), c, c);
    c2 = fe_cnt(c2, k1);
    uint64_t W4 = fe_madc(a1, P.v[3], FE_PAIR(c2, c3), k2);
    uint32_t T3 = fe_addc_co(FE_HI(W2), FE_LO(W3), c, c);
    W4 = fe_madc(a3, Q.v[3], W4, k0);
    uint32_t c4 = fe_cnt0(k2);
    uint32_t T4 = fe_addc_co(FE_HI(W3), FE_LO(W4), c, c);
    c4 = fe_cnt(c4, k0);
    // fold H = T6:T5:T4 (< 2^65): r = L + ((H * K) << 32) - H
    uint64_t P4 = (uint64_t)T4 * FE_K;
    uint32_t T5 = fe_addc0_co(FE_HI(W4), c, c);
    uint32_t r0 = fe_sub_co(T0, T4, b);
    uint64_t P5 = (uint64_t)T5 * FE_K;
    uint32_t T6 = fe_cnt(c4, c);                              // 0 or 1: the sum is below 2^193
    uint32_t u1 = fe_add_co(T1, FE_LO(P4), c);
    uint32_t X1 = fe_add_co(FE_HI(P4), FE_LO(P5), d);
    uint32_t x2 = fe_mad24(T6, FE_K, FE_HI(P5));
    uint32_t r1 = fe_subb_co(u1, T5, b, b);
    uint32_t X2 = fe_cnt(x2, d);
    uint32_t u2 = fe_addc_co(T2, X1, c, c);
    uint32_t z0 = fe_addm1_co(r0, d);
    uint32_t r2 = fe_subb_co(u2, T6, b, b);
    uint32_t u3 = fe_addc_co(T3, X2, c, c);
    uint32_t z1 = fe_addc_co(r1, FE_C1, d, d);
    uint32_t r3 = fe_subb0_co(u3, b, b);
    uint32_t z2 = fe_addc0_co(r2, d, d);
    fe_cf y4 = c & ~b;                                        // the value is below 2^128 + 2^111 < 2p and non-negative: a borrow only cancels a carry
    uint32_t z3 = fe_addc0_co(r3, d, d);
    fe_cf s = y4 | d;
    return fe_make(fe_sel(s, z0, r0), fe_sel(s, z1, r1), fe_sel(s, z2, r2), fe_sel(s, z3, r3));
}
FE_HD fe fe_mul_tw(const fe& a, const fe_tw& w) { return fe_mul_tw(a, w.p, w.q); }
// Q = w * 2^64 mod p (table construction; also on the fly where one multiplier serves several products)
FE_HD fe fe_shift64(const fe& w) {
    uint32_t t[8] = {0, 0, w.v[0], w.v[1], w.v[2], w.v[3], 0, 0};
    return fe_reduce8(t);
}
FE_HD fe_tw fe_tw_make(const fe& w) { fe_tw t; t.p = w; t.q = fe_shift64(w); return t; }

#if defined(__HIP_DEVICE_COMPILE__) || defined(FE_EMULATE_GFX950)
// ---- sums of products with ONE reduction ----------------------------------------------------------------------------------------
// A dot product sum_i a_i * b_i accumulates the 256-bit partial products in even/odd 64-bit windows (every accumulating mad counts its
// carry-out) and folds once at the end: 32 instructions per term + one reduction instead of a full multiplication and a modular
// addition per term.  Up to 64 terms (the overflow limb stays below 2^7).  Statement order = issue order (see fe_mul_tw): consecutive
// multiply-adds go to different windows and a counter trails its multiply-add by two statements.
struct fe_acc {
    uint64_t E0, E1, E2, E3, O0, O1, O2;              // windows at limbs 0, 2, 4, 6 and 1, 3, 5
    uint32_t cE0, cE1, cE2, cE3, cO0, cO1, cO2;       // 2^64 overflows of each window
};
FE_HD void fe_acc_zero(fe_acc& A) {
    A.E0 = A.E1 = A.E2 = A.E3 = A.O0 = A.O1 = A.O2 = 0;
    A.cE0 = A.cE1 = A.cE2 = A.cE3 = A.cO0 = A.cO1 = A.cO2 = 0;
}
FE_HD void fe_acc_mac(fe_acc& A, const fe& a, const fe& b) {
    const uint32_t a0 = a.v[0], a1 = a.v[1], a2 = a.v[2], a3 = a.v[3], b0 = b.v[0], b1 = b.v[1], b2 = b.v[2], b3 = b.v[3];
    fe_cf k0, k1, k2;
    A.E0 = fe_madc(a0, b0, A.E0, k0);
    A.O0 = fe_madc(a0, b1, A.O0, k1);
    A.E1 = fe_madc(a0, b2, A.E1, k2);   A.cE0 = fe_cnt(A.cE0, k0);
    A.O1 = fe_madc(a0, b3, A.O1, k0);   A.cO0 = fe_cnt(A.cO0, k1);
    A.E2 = fe_madc(a1, b3, A.E2, k1);   A.cE1 = fe_cnt(A.cE1, k2);
    A.O2 = fe_madc(a2, b3, A.O2, k2);   A.cO1 = fe_cnt(A.cO1, k0);
    A.E3 = fe_madc(a3, b3, A.E3, k0);   A.cE2 = fe_cnt(A.cE2, k1);
    A.O0 = fe_madc(a1, b0, A.O0, k1);   A.cO2 = fe_cnt(A.cO2, k2);
    A.E1 = fe_madc(a1, b1, A.E1, k2);   A.cE3 = fe_cnt(A.cE3, k0);
    A.O1 = fe_madc(a1, b2, A.O1, k0);   A.cO0 = fe_cnt(A.cO0, k1);
    A.E2 = fe_madc(a2, b2, A.E2, k1);   A.cE1 = fe_cnt(A.cE1, k2);
    A.O2 = fe_madc(a3, b2, A.O2, k2);   A.cO1 = fe_cnt(A.cO1, k0);
    A.E1 = fe_madc(a2, b0, A.E1, k0);   A.cE2 = fe_cnt(A.cE2, k1);
    A.O1 = fe_madc(a2, b1, A.O1, k1);   A.cO2 = fe_cnt(A.cO2, k2);
    A.E2 = fe_madc(a3, b1, A.E2, k2);   A.cE1 = fe_cnt(A.cE1, k0);
    A.O1 = fe_madc(a3, b0, A.O1, k0);   A.cO1 = fe_cnt(A.cO1, k1);
    A.cE2 = fe_cnt(A.cE2, k2);
    A.cO1 = fe_cnt(A.cO1, k0);
}
// adds a field element (weight 1) to the sum
FE_HD void fe_acc_add(fe_acc& A, const fe& a) {
    fe_cf k0, k1, k2;
    A.E0 = fe_madc(a.v[0], 1u, A.E0, k0);
    A.O0 = fe_madc(a.v[1], 1u, A.O0, k1);
    A.E1 = fe_madc(a.v[2], 1u, A.E1, k2);   A.cE0 = fe_cnt(A.cE0, k0);
    A.O1 = fe_madc(a.v[3], 1u, A.O1, k0);   A.cO0 = fe_cnt(A.cO0, k1);
    A.cE1 = fe_cnt(A.cE1, k2);
    A.cO1 = fe_cnt(A.cO1, k0);
}
FE_HD fe fe_acc_reduce(const fe_acc& A) {
    fe_cf c, d;
    uint32_t t0 = FE_LO(A.E0);
    uint32_t t1 = fe_add_co(FE_HI(A.E0), FE_LO(A.O0), c);
    uint32_t t2 = fe_addc_co(FE_LO(A.E1), FE_HI(A.O0), c, c);
    uint32_t t3 = fe_addc_co(FE_HI(A.E1), FE_LO(A.O1), c, c);
    t2 = fe_add_co(t2, A.cE0, d);                                  // the counters' chain runs two links behind the windows' chain
    uint32_t t4 = fe_addc_co(FE_LO(A.E2), FE_HI(A.O1), c, c);
    t3 = fe_addc_co(t3, A.cO0, d, d);
    uint32_t t5 = fe_addc_co(FE_HI(A.E2), FE_LO(A.O2), c, c);
    t4 = fe_addc_co(t4, A.cE1, d, d);
    uint32_t t6 = fe_addc_co(FE_LO(A.E3), FE_HI(A.O2), c, c);
    t5 = fe_addc_co(t5, A.cO1, d, d);
    uint32_t t7 = fe_addc0_co(FE_HI(A.E3), c, c);
    t6 = fe_addc_co(t6, A.cE2, d, d);
    uint32_t t8 = fe_cnt(A.cE3, c);
    t7 = fe_addc_co(t7, A.cO2, d, d);
    t8 = fe_cnt(t8, d);
    return fe_fold9(t0, t1, t2, t3, t4, t5, t6, t7, t8, true);
}
#else
// host pass of the same translation units: same interface, plain modular arithmetic (never on a hot path)
struct fe_acc { fe sum; };
FE_HD void fe_acc_zero(fe_acc& A) { A.sum = fe_zero(); }
FE_HD void fe_acc_mac(fe_acc& A, const fe& a, const fe& b) { A.sum = fe_add(A.sum, fe_mul_portable(a, b)); }
FE_HD void fe_acc_add(fe_acc& A, const fe& a) { A.sum = fe_add(A.sum, a); }
FE_HD fe fe_acc_reduce(const fe_acc& A) { return A.sum; }
#endif

// On the device, and in the host-emulated test build (FE_EMULATE_GFX950: the limb-level dataflow on plain integers), the windowed
// formulation; elsewhere on the host the portable one (tables).
FE_HD fe fe_mul(const fe& a, const fe& b) {
#if (defined(__HIP_DEVICE_COMPILE__) || defined(FE_EMULATE_GFX950)) && !defined(FE_PORTABLE_MUL)
    return fe_mul_wide(a, b);
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
    // square-and-multiply from the low bit; the exponent is shifted through four scalars (no run-time indexed limb array: no scratch)
    uint32_t w0 = e.v[0], w1 = e.v[1], w2 = e.v[2], w3 = e.v[3];
    fe r = fe_one();
    while ((w0 | w1 | w2 | w3) != 0) {
        if (w0 & 1u) r = fe_mul(r, b);
        w0 = (w0 >> 1) | (w1 << 31); w1 = (w1 >> 1) | (w2 << 31); w2 = (w2 >> 1) | (w3 << 31); w3 >>= 1;
        if ((w0 | w1 | w2 | w3) != 0) b = fe_sqr(b);
    }
    return r;
}
FE_HD fe fe_pow_u64(const fe& b, uint64_t e) { return fe_pow(b, fe_from_u64(e)); }
FE_HD fe fe_inv(const fe& a) {          // a^(p-2); inv(0) = 0 (field.rs:84)
    return fe_pow(a, fe_make(0xFFFFFFFFu, 0xFFFFD2FFu, 0xFFFFFFFFu, 0xFFFFFFFFu));   // exponent p - 2
}
