// AIR constraint evaluation over the degree-8 sub-domain of the LDE domain, one lane per evaluation point.
//
// Replaces the loop at /root/reference/src/stark/prover.rs:53-64: for every i in (0..N).step_by(B/8) it fills `current` =
// row i and `next` = row (i+B) mod N (trace_state.rs:251-277), derives the op flags (trace_state.rs:281-350), evaluates
// the boundary combinations (constraints/evaluator.rs:181-326) and the 20 + ctx + loop decoder constraints
// (constraints/decoder/{mod,op_bits,sponge,flow_ops}.rs) and 2 + stack_depth stack constraints (constraints/stack/*.rs),
// and folds them into one value per point with the random coefficients in degree-group order
// (evaluator.rs:335-358,385-406).  Instead of materialising the per-constraint vector the kernel accumulates
// sum cc[2i]*D and, per degree group, sum cc[2i+1]*D on the fly; x^p factors are table look-ups w_N^(i*p mod N)
// instead of field::exp.  The reference's quirks are reproduced on purpose: ld flag 2 uses
// cf bit 1, SWAP aggregates both constraints into slot 0, PUSH/ASSERT flag adjustments.
//
// Data access: with the coset-major LDE a point is (coset jl, index k); rows i and i+B are (jl,k) and (jl,k+1 mod n),
// so a wavefront reads 64 consecutive elements of each register -- unit stride, no halo.
#pragma once
#include "ctx.h"
#include <type_traits>

#ifndef AIR_THREADS
#define AIR_THREADS 128
#endif
// AIR_CONVOY (experiment, tools/ab_variant.sh): workgroup barriers between the sections of a launch keep the wavefronts of a workgroup
// at about the same program counter, so that they share the instruction-cache lines of straight-line code
#ifdef AIR_CONVOY
#define AIR_SYNC() __builtin_amdgcn_s_barrier()
#else
#define AIR_SYNC()
#endif
#ifndef AIR_WAVES_PER_SIMD
#define AIR_WAVES_PER_SIMD 2      // caps the kernel at 256 registers per lane
#endif
// Waves per SIMD the compiler must make room for: two = at most 256 registers per lane.  Since the evaluation is cut by operation group
// the nested-sum launches need 117 - 206 registers and no scratch (three to four waves per SIMD are resident); only the per-operation
// instance (tests) reaches the cap.  History: round 2's 172 KB stack launch ran three times slower on one lease and on the driver's box
// (12.7 - 13.4 against 4.3 - 4.6 ms); it had no scratch, so the size of its straight-line code is the suspect -- no
// launch is larger than the 64 KiB instruction cache any more.  -DAIR_WAVES_FORCE=3 builds a three-wave form for comparison.
constexpr int air_waves_per_simd(int sd, int slcap, int sect) {
    (void)sd; (void)slcap; (void)sect;
#ifdef AIR_WAVES_FORCE
    return AIR_WAVES_FORCE;
#else
    return AIR_WAVES_PER_SIMD;
#endif
}

// Rescue matrices (utils/sponge.rs:72-83, utils/hasher.rs:97-113), uploaded once per context to device memory
struct AirConsts { fe sponge_mds[16], sponge_inv_mds[16], hasher_mds[36], hasher_inv_mds[36]; };

struct AirArgs {
    const fe* lde;               // [W][Bc][n]
    fe* out;                     // [3][Q][n]  (i, f, t), Q = local constraint cosets
    const fe* coef;              // 344 raw draws: [0,94) first-step boundary, [94,188) last-step boundary
    const fe* tc;                // [2][NC] transition coefficients by constraint index: plain then degree-adjusted
    const fe* periodic;          // [128][AIR_PERIODIC_STRIDE]: 8 sponge ark, 12 hasher ark, 3 masks, cubes of hasher ark 0..5
    const AirConsts* consts;
    fe* partial;                 // [7][Q][n] partial sums between launches
    fe* ev[10];                  // [Q][n] each: partial values of the stack constraints between launches (slots 0..7, two auxiliary)
    const fe* tw_lo; const fe* tw_hi; uint32_t lo_bits;
    unsigned long long* bad_step;
    size_t n, col_stride;        // col_stride = Bc * n
    uint32_t log_n, log_N, log_b, W, ctx_depth, loop_depth, stack_depth;
    uint32_t cl, ll, sl;         // lengths of the ctx / loop / user stack slices: max(depth, 1), max(depth, 1), max(depth, 8) (trace_state.rs:58-60)
    uint32_t coset_step;         // B / 8: local lde coset of evaluation coset q is q * coset_step (within the local range)
    uint32_t q0;                 // global index of the first local evaluation coset
    uint32_t num_inputs, num_outputs;
    fe inputs[8], outputs[8], program_hash[2], op_count;
};



__device__ __forceinline__ fe dpow(const AirArgs& a, uint64_t e) {
    uint32_t l = (uint32_t)e & ((1u << a.lo_bits) - 1u), h = (uint32_t)(e >> a.lo_bits);
    fe v = a.tw_lo[l];
    return h ? fe_mul(v, a.tw_hi[h]) : v;
}

__device__ __forceinline__ fe bnot(const fe& v) { return fe_sub(fe_one(), v); }
__device__ __forceinline__ fe is_bin(const fe& v) { return fe_sub(fe_sqr(v), v); }

// MDS matrix times state: every row is a sum of Wd products with one reduction (fe_acc)
template <int Wd>
__device__ __forceinline__ void matmul(fe* s, const fe* m) {
    fe r[Wd];
#pragma unroll
    for (int i = 0; i < Wd; i++) {
        fe_acc A; fe_acc_zero(A);
#pragma unroll
        for (int j = 0; j < Wd; j++) fe_acc_mac(A, m[i * Wd + j], s[j]);
        r[i] = fe_acc_reduce(A);
    }
#pragma unroll
    for (int i = 0; i < Wd; i++) s[i] = r[i];
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for_air(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for_air<I + 1, N>(f); }
}

// ---- low-degree stack operations as nested sums (stack depth 4) ---------------------------------------------------------------
// The flag of low-degree operation `op` is lo2[op & 3] * mid[(op >> 2) & 3] * top[op >> 4] (trace_state.rs:281-350), and every
// stack constraint is sum_op flag(op) * term(op, slot) (constraints/stack/mod.rs:117-195).  Written as
//     sum_c top[c] * ( sum_b mid[b] * ( sum_a lo2[a] * term(a + 4b + 16c, slot) ) )
// the 32 flags are never formed and every level is a sum of products with one reduction (fe_acc): ~13k instructions per point
// instead of ~21k for flag products + per-term multiply-adds.  st_term states, per operation and slot, exactly what the
// per-operation code below passes to agg(): slots 0..3 are the stack constraints, 4 and 5 the two auxiliary constraints.
struct StackRows { fe o[12], nw[8], hd0; int sl; };       // o[8..11] and sl only matter to the deep instance (slice longer than 8)
#define ST_AUX0 8
#define ST_AUX1 9

// stack slots 0..7 (the slice always has max(depth, 8) = 8 items for depth <= 8), ST_AUX0, ST_AUX1
template <int OP, int I> constexpr bool st_has() {
    constexpr bool slot = I < 8, aux0 = I == ST_AUX0, aux1 = I == ST_AUX1;
    switch (OP) {
        case 0x00: case 0x01: case 0x05: case 0x06: case 0x07: return slot || aux0;       // ASSERT, ASSERTEQ, CHOOSE, CHOOSE2, CSWAP2
        case 0x02: case 0x0E: return slot || aux0;                                       // EQ, NOT
        case 0x03: case 0x04: case 0x08: case 0x09: case 0x0C: case 0x0D: return slot;   // DROP, DROP4, ADD, MUL, INV, NEG
        case 0x0A: case 0x0B: return slot || aux0 || aux1;                               // AND, OR
        case 0x10: return I >= 1 && slot;                                                // READ
        case 0x11: return I >= 2 && slot;                                                // READ2
        case 0x12: case 0x13: case 0x14: case 0x15: return slot;                         // DUP, DUP2, DUP4, PAD2
        case 0x18: return slot && I != 1;                                                // SWAP: both constraints in slot 0 (manipulation.rs:63-64)
        case 0x19: case 0x1A: case 0x1B: case 0x1C: case 0x1D: return slot;              // SWAP2, SWAP4, ROLL4, ROLL8, BINACC
        default: return false;
    }
}

template <int OP, int I>
__device__ __forceinline__ fe st_term(const StackRows& s) {
    const fe* o = s.o; const fe* nw = s.nw;
    constexpr int i = I < 8 ? I : 0;
    // the shift helpers of constraints/stack/mod.rs on a slice of s.sl >= 8 items (8 unless the stack is deeper): a left shift by
    // num zero-fills the last num slots
    auto L = [&](int num) {
        if (i + num < 8) return fe_sub(o[i + num < 8 ? i + num : 0], nw[i]);
        return i + num < s.sl ? fe_sub(o[i + num < 12 ? i + num : 0], nw[i]) : nw[i];
    };
    auto R = [&](int num) { return fe_sub(o[i - num < 0 ? 0 : i - num], nw[i]); };   // right shift by num (callers: i >= num)
    auto C = [&]() { return fe_sub(o[i], nw[i]); };                       // copy
    auto sel = [&](const fe& c, const fe& x, const fe& y) { return fe_add(y, fe_mul(c, fe_sub(x, y))); };   // c*x + (1-c)*y with one multiplication
    if constexpr (OP == 0x00) { if constexpr (I == ST_AUX0) return bnot(o[0]); else return L(1); }   // the ASSERT flag carries hd[0] (trace_state.rs:346): st_output puts it into the coefficient
    else if constexpr (OP == 0x01) { if constexpr (I == ST_AUX0) return fe_sub(o[0], o[1]); else return L(2); }
    else if constexpr (OP == 0x02) {
        const fe diff = fe_sub(o[1], o[2]);
        if constexpr (I == ST_AUX0) return fe_mul(nw[0], diff);
        else if constexpr (I == 0) return fe_sub(nw[0], bnot(fe_mul(diff, o[0])));
        else return L(2);
    }
    else if constexpr (OP == 0x03) return L(1);
    else if constexpr (OP == 0x04) return L(4);
    else if constexpr (OP == 0x05) {
        if constexpr (I == ST_AUX0) return is_bin(o[2]);
        else if constexpr (I == 0) return fe_sub(nw[0], sel(o[2], o[0], o[1]));
        else return L(2);
    }
    else if constexpr (OP == 0x06) {
        if constexpr (I == ST_AUX0) return is_bin(o[4]);
        else if constexpr (I == 0) return fe_sub(nw[0], sel(o[4], o[0], o[2]));
        else if constexpr (I == 1) return fe_sub(nw[1], sel(o[4], o[1], o[3]));
        else return L(4);
    }
    else if constexpr (OP == 0x07) {
        if constexpr (I == ST_AUX0) return is_bin(o[4]);
        else if constexpr (I == 0) return fe_sub(nw[0], sel(o[4], o[2], o[0]));
        else if constexpr (I == 1) return fe_sub(nw[1], sel(o[4], o[3], o[1]));
        else if constexpr (I == 2) return fe_sub(nw[2], sel(o[4], o[0], o[2]));
        else if constexpr (I == 3) return fe_sub(nw[3], sel(o[4], o[1], o[3]));
        else return L(2);
    }
    else if constexpr (OP == 0x08) { if constexpr (I == 0) return fe_sub(nw[0], fe_add(o[0], o[1])); else return L(1); }
    else if constexpr (OP == 0x09) { if constexpr (I == 0) return fe_sub(nw[0], fe_mul(o[0], o[1])); else return L(1); }
    else if constexpr (OP == 0x0A) {
        if constexpr (I == ST_AUX0) return is_bin(o[0]);
        else if constexpr (I == ST_AUX1) return is_bin(o[1]);
        else if constexpr (I == 0) return fe_sub(nw[0], fe_mul(o[0], o[1]));
        else return L(1);
    }
    else if constexpr (OP == 0x0B) {
        if constexpr (I == ST_AUX0) return is_bin(o[0]);
        else if constexpr (I == ST_AUX1) return is_bin(o[1]);
        else if constexpr (I == 0) return fe_sub(nw[0], fe_sub(fe_add(o[0], o[1]), fe_mul(o[0], o[1])));     // 1 - (1-x)(1-y) = x + y - xy (shares xy with MUL / AND)
        else return L(1);
    }
    else if constexpr (OP == 0x0C) { if constexpr (I == 0) return fe_sub(fe_one(), fe_mul(nw[0], o[0])); else return C(); }
    else if constexpr (OP == 0x0D) { if constexpr (I == 0) return fe_add(nw[0], o[0]); else return C(); }
    else if constexpr (OP == 0x0E) {
        if constexpr (I == ST_AUX0) return is_bin(o[0]);
        else if constexpr (I == 0) return fe_sub(nw[0], bnot(o[0]));
        else return C();
    }
    else if constexpr (OP == 0x10) return R(1);
    else if constexpr (OP == 0x11) return R(2);
    else if constexpr (OP == 0x12) { if constexpr (I == 0) return fe_sub(nw[0], o[0]); else return R(1); }
    else if constexpr (OP == 0x13) { if constexpr (I < 2) return fe_sub(nw[i], o[i]); else return R(2); }
    else if constexpr (OP == 0x14) { if constexpr (I < 4) return fe_sub(nw[i], o[i]); else return R(4); }
    else if constexpr (OP == 0x15) { if constexpr (I < 2) return nw[i]; else return R(2); }
    else if constexpr (OP == 0x18) { if constexpr (I == 0) return fe_add(fe_sub(nw[0], o[1]), fe_sub(nw[1], o[0])); else return C(); }
    else if constexpr (OP == 0x19) { if constexpr (I < 4) return fe_sub(nw[i], o[i ^ 2]); else return C(); }
    else if constexpr (OP == 0x1A) return fe_sub(nw[i], o[i ^ 4]);
    else if constexpr (OP == 0x1B) { if constexpr (I < 4) return fe_sub(nw[i], o[(i + 3) & 3]); else return C(); }
    else if constexpr (OP == 0x1C) return fe_sub(nw[i], o[(i + 7) & 7]);
    else if constexpr (OP == 0x1D) {
        if constexpr (I == 0) return is_bin(nw[0]);
        else if constexpr (I == 1) return nw[1];
        else if constexpr (I == 2) return fe_sub(nw[2], fe_double(o[2]));
        else if constexpr (I == 3) return fe_sub(nw[3], fe_add(o[3], fe_mul(nw[0], o[2])));
        else return C();
    }
    else return fe_zero();
}

// ---- which operations a launch evaluates ---------------------------------------------------------------------------------------------
// The stack constraints are linear in the operation flags, so a launch may evaluate any subset of the operations for all slots and hand
// the partial values on through memory (AirArgs::ev); the launch that holds the last subset emits the constraints.  Cutting by operation
// group rather than by slot keeps what a launch has to derive first small: the low-degree groups of one `c` need lo2, mid and one top
// entry, the high-degree / flow operations the four hd flags -- and no straight-line kernel grows past the 64 KiB instruction cache of a
// CU pair (tools/codeobj_info.py, gated by build()).
//   bits 0..3: low-degree operations 4b .. 4b+3 (c = 0), bits 4..7: 16 + 4b .. (c = 1), then PUSH, CMP, RESCR, BEGIN / NOOP ("keep"),
//   and the slots >= 8 of the deep instance.
#define AG_LOW(c, b) (1u << ((c) * 4 + (b)))
#define AG_LOW0 0x00Fu
#define AG_LOW1 0x0F0u
#define AG_PUSH 0x100u
#define AG_CMP 0x200u
#define AG_RESCR 0x400u
#define AG_KEEP 0x800u
#define AG_HIGH 0xF00u
#define AG_DEEP 0x1000u
#define AG_STACK_ALL 0x1FFFu
// FLAGS of a launch: FIRST / LAST in the chain of partial sums (res, adj[6]); EV_IN: partial stack values come in; EV_OUT: they go out
// (else this launch emits the stack constraints)
#define AF_FIRST 1
#define AF_LAST 2
#define AF_EV_IN 4
#define AF_EV_OUT 8

// What term(op, slot) is, for the terms that are plain differences of stack items: kind 1: o[j] - nw[I], 2: nw[I] - o[j], 3: nw[I],
// 4: -nw[I], 5: anything else (st_term computes it), 0: the operation does not constrain the slot.  SDK = compile-time stack depth (0:
// known at run time only): items o[j], j >= SDK, are zeros then (trace_state.rs:58-60 pads the slice), which turns differences into +-nw[I]
// and removes the selector of CHOOSE2 / CSWAP2 (item 4).  DEEP: the slice may be longer than 8, left shifts past item 7 stay with st_term.
struct StDesc { int kind, j; };
template <int SDK, bool DEEP>
constexpr StDesc st_desc(int op, int I) {
    const bool o4_zero = SDK != 0 && SDK <= 4;
    if (I >= 8) {                                                      // auxiliary constraints: computed; j >= 100 names values that several operations share
        if ((op == 0x0A || op == 0x0B)) return StDesc{5, I == ST_AUX0 ? 100 : 101};         // is_bin(o[0]) / is_bin(o[1])
        if ((op == 0x06 || op == 0x07) && o4_zero) return StDesc{0, 0};                     // is_bin(o[4]) of a zero
        return StDesc{5, -1};
    }
    const int i = I;
    auto diff = [&](int j) { return (SDK != 0 && j >= SDK) ? StDesc{4, 0} : StDesc{1, j}; };
    auto rdiff = [&](int j) { return (SDK != 0 && j >= SDK) ? StDesc{3, 0} : StDesc{2, j}; };
    auto L = [&](int num) { return i + num < 8 ? diff(i + num) : (DEEP ? StDesc{5, -1} : StDesc{3, 0}); };
    auto R = [&](int num) { return diff(i - num < 0 ? 0 : i - num); };
    switch (op) {
        case 0x00: return L(1);
        case 0x01: return L(2);
        case 0x02: return i == 0 ? StDesc{5, -1} : L(2);
        case 0x03: return L(1);
        case 0x04: return L(4);
        case 0x05: return i == 0 ? StDesc{5, -1} : L(2);
        case 0x06: return i < 2 ? (o4_zero ? rdiff(i + 2) : StDesc{5, -1}) : L(4);          // selector o[4] = 0: the second operand
        case 0x07: return i < 4 ? (o4_zero ? rdiff(i) : StDesc{5, -1}) : L(2);
        case 0x08: case 0x0B: return i == 0 ? StDesc{5, -1} : L(1);
        case 0x09: case 0x0A: return i == 0 ? StDesc{5, 102} : L(1);                         // MUL and AND: nw[0] - o[0] * o[1]
        case 0x0C: case 0x0D: case 0x0E: return i == 0 ? StDesc{5, -1} : diff(i);
        case 0x10: return R(1);
        case 0x11: return R(2);
        case 0x12: return i == 0 ? rdiff(0) : R(1);
        case 0x13: return i < 2 ? rdiff(i) : R(2);
        case 0x14: return i < 4 ? rdiff(i) : R(4);
        case 0x15: return i < 2 ? StDesc{3, 0} : R(2);
        case 0x18: return i == 0 ? StDesc{5, -1} : diff(i);
        case 0x19: return i < 4 ? rdiff(i ^ 2) : diff(i);
        case 0x1A: return rdiff(i ^ 4);
        case 0x1B: return i < 4 ? rdiff((i + 3) & 3) : diff(i);
        case 0x1C: return rdiff((i + 7) & 7);
        case 0x1D: return i == 1 ? StDesc{3, 0} : (i < 4 ? StDesc{5, -1} : diff(i));
        default: return StDesc{0, 0};
    }
}
// operations of one group (4 consecutive opcodes) whose terms for slot I coincide up to sign share ONE product: entry e of the merge
// holds the basis term (kind / j of its first member) and the sign of every member's coefficient relative to it
struct StMerge { int n; int kind[4], j[4], lead[4]; int sgn[4][4]; };
template <int BASE, int I, int SDK, bool DEEP>
constexpr StMerge st_merge() {
    StMerge m{};
    for (int a = 0; a < 4; a++) {
        const StDesc d = st_desc<SDK, DEEP>(BASE + a, I);
        // has the operation a term here at all?  (st_has is the authority: st_desc only classifies)
        bool has = false;
        switch (a) { case 0: has = st_has<BASE, I>(); break; case 1: has = st_has<BASE + 1, I>(); break; case 2: has = st_has<BASE + 2, I>(); break; default: has = st_has<BASE + 3, I>(); }
        if (!has || d.kind == 0) continue;
        const int cls = d.kind == 5 ? 5 : (d.kind <= 2 ? 1 : 3);           // 1: differences with o[j], 3: +-nw[I], 5: computed
        const int sign = (d.kind == 2 || d.kind == 4) ? -1 : 1;
        int e = -1;
        for (int k = 0; k < m.n; k++) {
            const int kc = m.kind[k] == 5 ? 5 : (m.kind[k] <= 2 ? 1 : 3);
            if (kc != cls) continue;
            if (cls == 3 || (cls == 1 && m.j[k] == d.j) || (cls == 5 && d.j >= 100 && m.j[k] == d.j)) e = k;
        }
        if (e < 0) { e = m.n++; m.kind[e] = d.kind; m.j[e] = d.j; m.lead[e] = a; m.sgn[e][a] = 1; }
        else { const int lead_sign = (m.kind[e] == 2 || m.kind[e] == 4) ? -1 : 1; m.sgn[e][a] = sign * lead_sign; }
    }
    return m;
}

// The operations selected by the high-degree bits and the flow flags join the outermost sum of a slot (one more product each,
// no separate multiplication and modular addition): PUSH (input.rs:6), CMP (comparison.rs:64-105), RESCR (hash.rs:9-35; its six
// differences are computed once by the caller), BEGIN / NOOP (stack untouched).
struct StackHigh { fe push, cmp, rescr, keep; fe resc[6]; };       // the four flags and the RESCR differences

template <int I, uint32_t GROUPS>
__device__ __forceinline__ void st_high_degree(fe_acc& outer, const StackRows& s, const StackHigh& h) {
    if constexpr (I < 8) {
        const fe* o = s.o; const fe* nw = s.nw;
        if constexpr (I >= 1 && (GROUPS & AG_PUSH) != 0) fe_acc_mac(outer, h.push, fe_sub(o[I - 1 < 0 ? 0 : I - 1], nw[I]));
        if constexpr ((GROUPS & AG_CMP) != 0) {
            const fe x_bit = nw[1], y_bit = nw[2], not_set = nw[3];
            fe v;
            if constexpr (I == 0) v = is_bin(x_bit);
            else if constexpr (I == 1) v = is_bin(y_bit);
            else if constexpr (I == 2) v = fe_sub(nw[4], fe_add(o[4], fe_mul(fe_mul(x_bit, bnot(y_bit)), not_set)));
            else if constexpr (I == 3) v = fe_sub(nw[5], fe_add(o[5], fe_mul(fe_mul(y_bit, bnot(x_bit)), not_set)));
            else if constexpr (I == 4) v = fe_sub(nw[6], fe_add(o[6], fe_mul(y_bit, o[0])));
            else if constexpr (I == 5) v = fe_sub(nw[7], fe_add(o[7], fe_mul(x_bit, o[0])));
            else if constexpr (I == 6) v = fe_sub(not_set, fe_mul(bnot(o[5]), bnot(o[4])));
            else v = fe_sub(fe_double(nw[0]), o[0]);
            fe_acc_mac(outer, h.cmp, v);
        }
        if constexpr ((GROUPS & (AG_RESCR | AG_KEEP)) != 0) {
            const fe keep = fe_sub(o[I < 8 ? I : 0], nw[I < 8 ? I : 0]);
            if constexpr ((GROUPS & AG_RESCR) != 0) { if constexpr (I < 6) fe_acc_mac(outer, h.rescr, h.resc[I < 6 ? I : 0]); else fe_acc_mac(outer, h.rescr, keep); }
            if constexpr ((GROUPS & AG_KEEP) != 0) fe_acc_mac(outer, h.keep, keep);
        }
    }
}

// does the launch's selection contribute anything to output I?  (the auxiliary constraints only hear from low-degree operations of c = 0)
template <int I, uint32_t GROUPS> constexpr bool st_touches() {
    if (I >= 8) return (GROUPS & AG_LOW0) != 0;
    return (GROUPS & (AG_LOW0 | AG_LOW1 | AG_HIGH)) != 0;
}

// output I (slot 0..7, ST_AUX0, ST_AUX1) of the selected operations:
//     sum_c top[c] * ( sum_b mid[b] * ( sum_e coefficient_e * basis term_e ) )  +  high-degree / flow flags * their terms
// every level one sum of products with a single reduction (fe_acc); coefficient_e = +-lo2[a] summed over the merged members
// (lo0h = lo2[0] * hd[0] stands for lo2[0] in group 0: the ASSERT flag carries hd[0], trace_state.rs:346)
template <int I, uint32_t GROUPS, int SDK, bool DEEP>
__device__ __forceinline__ fe st_output(const StackRows& s, const fe* lo2, const fe& lo0h, const fe* mid, const fe* top, const StackHigh& h) {
    fe_acc outer; fe_acc_zero(outer);
    static_for_air<0, 2>([&](auto c_) {
        constexpr int c = decltype(c_)::value;
        if constexpr (((GROUPS >> (4 * c)) & 0xFu) != 0) {
            fe_acc middle; fe_acc_zero(middle);
            static_for_air<0, 4>([&](auto b_) {
                constexpr int b = decltype(b_)::value;
                constexpr int base = 4 * b + 16 * c;
                if constexpr ((GROUPS & AG_LOW(c, b)) != 0) {
                    static constexpr StMerge M = st_merge<base, I, SDK, DEEP>();
                    if constexpr (M.n > 0) {
                        fe_acc inner; fe_acc_zero(inner);
                        static_for_air<0, M.n>([&](auto e_) {
                            constexpr int e = decltype(e_)::value;
                            // coefficient: the leader's, plus / minus the other members'
                            fe cf = (base == 0 && M.lead[e] == 0) ? lo0h : lo2[M.lead[e]];
                            static_for_air<0, 4>([&](auto a_) {
                                constexpr int a = decltype(a_)::value;
                                if constexpr (a != M.lead[e] && M.sgn[e][a] != 0) {
                                    const fe other = (base == 0 && a == 0) ? lo0h : lo2[a];
                                    cf = M.sgn[e][a] > 0 ? fe_add(cf, other) : fe_sub(cf, other);
                                }
                            });
                            constexpr int ii = I < 8 ? I : 0;
                            fe term;
                            if constexpr (M.kind[e] == 1) term = fe_sub(s.o[M.j[e]], s.nw[ii]);
                            else if constexpr (M.kind[e] == 2) term = fe_sub(s.nw[ii], s.o[M.j[e]]);
                            else if constexpr (M.kind[e] == 3) term = s.nw[ii];
                            else if constexpr (M.kind[e] == 4) term = fe_neg(s.nw[ii]);
                            else term = st_term<base + M.lead[e], I>(s);
                            fe_acc_mac(inner, cf, term);
                        });
                        fe_acc_mac(middle, mid[b], fe_acc_reduce(inner));
                    }
                }
            });
            fe_acc_mac(outer, top[c], fe_acc_reduce(middle));
        }
    });
    if constexpr ((GROUPS & AG_HIGH) != 0) st_high_degree<I, GROUPS>(outer, s, h);
    return fe_acc_reduce(outer);
}

// degree-group slots: 2->0 3->1 4->2 6->3 7->4 8->5
// ACC_MASK: the degree-group sums (bit = slot) that are fe_acc accumulators too; an accumulator is 21 registers, so launches that
// emit into many groups keep the rarely used ones as plain field elements
template <uint32_t ACC_MASK>
struct Acc {
    fe_acc res_acc;        // sum over all constraints of value * coefficient: one reduction at the end of the launch (fe_acc)
    fe_acc adj_acc[6];     // only the entries selected by ACC_MASK are ever touched (the others are eliminated)
    fe res, adj[6];
    bool nonzero;
    const fe* tc; uint32_t nc;
    __device__ __forceinline__ void emit(uint32_t cidx, int slot, const fe& d) {
        nonzero |= !fe_is_zero(d);
        fe_acc_mac(res_acc, d, tc[cidx]);
        if ((ACC_MASK >> slot) & 1u) fe_acc_mac(adj_acc[slot], d, tc[nc + cidx]);
        else adj[slot] = fe_add(adj[slot], fe_mul(d, tc[nc + cidx]));
    }
};

// CL, LL: compile-time capacities of the context / loop stack slices (>= a.cl, a.ll).
// SD: compile-time user stack depth, or 0 when the depth is only known at run time (then SLCAP bounds the slice length).
// Only the first `stack_depth` stack constraints are emitted (stack/mod.rs:194), so with SD known the unused slots vanish.
// SECT selects the constraint sections of this launch: bit 0 boundary, 1 op bits, 2 sponge, 7 loop image + context / loop stacks; for the per-operation
// formulation (the instance with SLCAP == 32) also 3 stack: low-degree ops that move items, 5 stack: low-degree arithmetic / selection
// ops, 4 stack: PUSH / CMP / BEGIN / NOOP, 6 stack: RESCR.  The nested-sum instances select stack OPERATIONS with GROUPS (AG_*).
// The combination is linear in the constraints, so a launch that is not FIRST starts from the partial sums (res, adj[6]) left by the
// previous launch and one that is not LAST stores them; the stack constraints themselves are linear in the operation flags, so their
// partial values travel the same way (AF_EV_IN / AF_EV_OUT) and are emitted by the launch that completes them.  Splitting the evaluation
// this way keeps the live state of each launch within the register budget and its code within the instruction cache.
template <int CL, int LL, int SD, int SLCAP, int SECT, uint32_t GROUPS, int FLAGS>
__global__ void __launch_bounds__(AIR_THREADS, air_waves_per_simd(SD, SLCAP, SECT)) air_kernel(AirArgs a) {
    constexpr bool FIRST = (FLAGS & AF_FIRST) != 0, LAST = (FLAGS & AF_LAST) != 0, EV_IN = (FLAGS & AF_EV_IN) != 0, EV_OUT = (FLAGS & AF_EV_OUT) != 0;
    constexpr int SL = SD ? (SD > 8 ? SD : 8) : SLCAP;
    // SLCAP == 12 is the deep instance: any stack depth; slots 0..7 as nested sums from 12 register-resident items of the current
    // row, slots 8.. from memory as seven flag sums times shifted differences (see the stack section)
    constexpr bool DEEP = SD == 0 && SLCAP == 12;
    const int cl = (int)a.cl, ll = (int)a.ll;
    const int sl = SD ? SL : (int)a.sl;
    const int sd = SD ? SD : (int)a.stack_depth;
    const fe* c_sponge_mds = a.consts->sponge_mds; const fe* c_sponge_inv_mds = a.consts->sponge_inv_mds;
    const fe* c_hasher_mds = a.consts->hasher_mds; const fe* c_hasher_inv_mds = a.consts->hasher_inv_mds;
    const size_t k = (size_t)blockIdx.x * AIR_THREADS + threadIdx.x;
    if (k >= a.n) return;
    const uint32_t ql = blockIdx.y;                       // local evaluation coset
    const uint32_t qg = a.q0 + ql;
    const uint32_t jl = ql * a.coset_step;                // local lde coset
    const size_t kn = (k + 1 == a.n) ? 0 : k + 1;
    const fe* cur_p = a.lde + (size_t)jl * a.n + k;
    const fe* nxt_p = a.lde + (size_t)jl * a.n + kn;
    const size_t cs = a.col_stride;
#define CUR(c) cur_p[(size_t)(c) * cs]
#define NXT(c) nxt_p[(size_t)(c) * cs]

    const uint64_t nmask = ((uint64_t)1 << a.log_N) - 1;
    const uint64_t gi = ((uint64_t)k << a.log_b) + (uint64_t)qg * a.coset_step;     // index into the LDE domain, x = w_N^gi
    const uint32_t step = (uint32_t)(((uint64_t)k << 3) + qg);                      // index into the 8n evaluation domain
    const uint64_t n64 = a.n;

    // ---- rows -------------------------------------------------------------------------------------------------------
    const fe c_opc = CUR(0), n_opc = NXT(0);
    fe c_sp[4], n_sp[4], cf[3], ld[5], hd[2], n_cf[3];
#pragma unroll
    for (int i = 0; i < 4; i++) { c_sp[i] = CUR(1 + i); n_sp[i] = NXT(1 + i); }
#pragma unroll
    for (int i = 0; i < 3; i++) { cf[i] = CUR(5 + i); n_cf[i] = NXT(5 + i); }
#pragma unroll
    for (int i = 0; i < 5; i++) ld[i] = CUR(8 + i);
#pragma unroll
    for (int i = 0; i < 2; i++) hd[i] = CUR(13 + i);
    fe c_ctx[CL], n_ctx[CL], c_lp[LL], n_lp[LL], o[SL], nw[SL];
    {
        uint32_t col = 15;
#pragma unroll
        for (int i = 0; i < CL; i++) { bool on = (uint32_t)i < a.ctx_depth; c_ctx[i] = on ? CUR(col + i) : fe_zero(); n_ctx[i] = on ? NXT(col + i) : fe_zero(); }
        col += a.ctx_depth;
#pragma unroll
        for (int i = 0; i < LL; i++) { bool on = (uint32_t)i < a.loop_depth; c_lp[i] = on ? CUR(col + i) : fe_zero(); n_lp[i] = on ? NXT(col + i) : fe_zero(); }
        col += a.loop_depth;
#pragma unroll
        for (int i = 0; i < SL; i++) { bool on = i < sd; o[i] = on ? CUR(col + i) : fe_zero(); nw[i] = on ? NXT(col + i) : fe_zero(); }
    }

    // ---- boundary constraints (evaluator.rs:181-326) -----------------------------------------------------------------
    if constexpr ((SECT & 1) != 0) {
        const fe xp = dpow(a, (gi * (6 * n64 + 2)) & nmask);            // b_degree_adj = 7n + 1 - (n - 1)
        const fe one = fe_one();
#pragma unroll 1
        for (int pass = 0; pass < 2; pass++) {
            const fe* cc = a.coef + pass * 94;
            // two sums of ~25 products each, reduced once (fe_acc)
            fe_acc R, A; fe_acc_zero(R); fe_acc_zero(A);
            auto term = [&](const fe& v, uint32_t idx) { fe_acc_mac(R, v, cc[idx]); fe_acc_mac(A, v, cc[idx + 1]); };
            term(pass ? fe_sub(c_opc, a.op_count) : c_opc, 0);
            if (pass == 0) { for (int i = 0; i < 4; i++) term(c_sp[i], 2 + 2 * i); }
            else { for (int i = 0; i < 2; i++) term(fe_sub(c_sp[i], a.program_hash[i]), 2 + 2 * i); }
            for (int i = 0; i < 3; i++) term(pass ? fe_sub(cf[i], one) : cf[i], 10 + 2 * i);
            for (int i = 0; i < 5; i++) term(pass ? fe_sub(ld[i], one) : ld[i], 16 + 2 * i);
            for (int i = 0; i < 2; i++) term(pass ? fe_sub(hd[i], one) : hd[i], 26 + 2 * i);
            for (int i = 0; i < CL; i++) if (i < cl) term(c_ctx[i], 30 + 2 * i);
            for (int i = 0; i < LL; i++) if (i < ll) term(c_lp[i], 62 + 2 * i);
            const uint32_t nio = pass ? a.num_outputs : a.num_inputs;
#pragma unroll
            for (int i = 0; i < 8; i++) if ((uint32_t)i < nio) term(fe_sub(o[i], pass ? a.outputs[i] : a.inputs[i]), 78 + 2 * i);
            const fe adj = fe_acc_reduce(A);
            fe_acc_mac(R, adj, xp);
            a.out[((size_t)pass * gridDim.y + ql) * a.n + k] = fe_acc_reduce(R);
        }
    }

    AIR_SYNC();
    // ---- op flags (trace_state.rs:281-350) -----------------------------------------------------------------------------
    fe cff[8], hdf[4], begin_flag, noop_flag, n_void, assert_flag;
    // low-degree op flags are products of three small tables and are formed where they are used:
    // ldf(idx) = lo2[idx & 3] * mid[(idx >> 2) & 3] * top[idx >> 4]
    fe lo2[4], mid[4], top[2];
    {
        fe n0 = bnot(cf[0]), n1 = bnot(cf[1]), n2 = bnot(cf[2]);
        fe t0 = fe_mul(n0, n1), t1 = fe_mul(cf[0], n1), t2 = fe_mul(n0, cf[1]), t3 = fe_mul(cf[0], cf[1]);
        cff[0] = fe_mul(t0, n2); cff[1] = fe_mul(t1, n2); cff[2] = fe_mul(t2, n2); cff[3] = fe_mul(t3, n2);
        cff[4] = fe_mul(t0, cf[2]); cff[5] = fe_mul(t1, cf[2]); cff[6] = fe_mul(t2, cf[2]); cff[7] = fe_mul(t3, cf[2]);
        n_void = fe_mul(fe_mul(n_cf[0], n_cf[1]), n_cf[2]);             // next.cf_op_flags()[VOID]
        fe l0 = bnot(ld[0]), l1 = bnot(ld[1]), l2 = bnot(ld[2]), l3 = bnot(ld[3]);
        lo2[0] = fe_mul(l0, l1); lo2[1] = fe_mul(ld[0], l1);
        lo2[2] = fe_mul(l0, cf[1]);                                    // sic: cf_op_bits[1] (trace_state.rs:301)
        lo2[3] = fe_mul(ld[0], ld[1]);
        mid[0] = fe_mul(l2, l3); mid[1] = fe_mul(ld[2], l3); mid[2] = fe_mul(l2, ld[3]); mid[3] = fe_mul(ld[2], ld[3]);
        top[0] = bnot(ld[4]); top[1] = ld[4];
        fe h0 = bnot(hd[0]), h1 = bnot(hd[1]);
        hdf[0] = fe_mul(h0, h1); hdf[1] = fe_mul(hd[0], h1); hdf[2] = fe_mul(h0, hd[1]); hdf[3] = fe_mul(hd[0], hd[1]);
    }
    auto LDF = [&](int idx) { return fe_mul(fe_mul(lo2[idx & 3], mid[(idx >> 2) & 3]), top[idx >> 4]); };
    {
        fe ld0 = LDF(0);
        begin_flag = fe_mul(ld0, hdf[0]);                              // trace_state.rs:329
        noop_flag = fe_mul(LDF(31), hdf[3]);                           // trace_state.rs:335
        hdf[0] = fe_mul(hdf[0], ld[0]);                                // PUSH (trace_state.rs:343)
        assert_flag = fe_mul(ld0, hd[0]);                              // ASSERT (trace_state.rs:346)
    }

    AIR_SYNC();
    const fe* per = a.periodic + (size_t)(step & 127u) * AIR_PERIODIC_STRIDE;
    // specialised instances: every group a launch without op bits emits into; with op bits (five groups) only degree 2, which
    // takes ten of its fifteen constraints
    constexpr uint32_t ACC_MASK = !(SD != 0 || SLCAP == 8 || DEEP) ? 0u : ((SECT & 2) ? 0x01u : 0x3Fu);
    Acc<ACC_MASK> acc;
#pragma unroll
    for (int i = 0; i < 6; i++) if ((ACC_MASK >> i) & 1u) fe_acc_zero(acc.adj_acc[i]);
    fe_acc_zero(acc.res_acc); acc.res = fe_zero(); acc.nonzero = false; acc.tc = a.tc; acc.nc = 20 + cl + ll + 2 + sd;
#pragma unroll
    for (int i = 0; i < 6; i++) acc.adj[i] = fe_zero();
    const size_t pstride = (size_t)gridDim.y * a.n, pidx = (size_t)ql * a.n + k;

    // ---- decoder: op bits (decoder/op_bits.rs:10-79) -------------------------------------------------------------------
    if constexpr ((SECT & 2) != 0) {
        fe cf_sum = fe_add(fe_add(cf[0], cf[1]), cf[2]);
        fe ld_prod = fe_mul(fe_mul(fe_mul(ld[0], ld[1]), fe_mul(ld[2], ld[3])), ld[4]);
        fe hd_prod = fe_mul(hd[0], hd[1]);
#pragma unroll
        for (int i = 0; i < 3; i++) acc.emit(i, 0, is_bin(cf[i]));
#pragma unroll
        for (int i = 0; i < 5; i++) acc.emit(3 + i, 0, is_bin(ld[i]));
#pragma unroll
        for (int i = 0; i < 2; i++) acc.emit(8 + i, 0, is_bin(hd[i]));
        const fe is_hacc = cff[0];
        fe hacc_tr = fe_mul(fe_add(c_opc, fe_one()), is_hacc);
        fe rest_tr = fe_mul(c_opc, bnot(is_hacc));
        acc.emit(10, 1, fe_sub(fe_add(hacc_tr, rest_tr), n_opc));
        acc.emit(11, 5, fe_mul(c_opc, fe_mul(bnot(ld_prod), bnot(hd_prod))));
        acc.emit(12, 5, fe_mul(cf_sum, bnot(fe_mul(ld_prod, hd_prod))));
        acc.emit(13, 3, fe_mul(cff[7], bnot(n_void)));
        fe prefix = fe_add(fe_add(cff[1], cff[4]), fe_add(cff[5], cff[6]));          // BEGIN, LOOP, WRAP, BREAK
        fe v = fe_mul(prefix, per[21]);
        v = fe_add(v, fe_mul(fe_add(cff[2], cff[3]), per[20]));                      // TEND, FEND
        v = fe_add(v, fe_mul(hdf[0], per[22]));                                      // PUSH
        acc.emit(14, 2, v);
    }

    // ---- decoder: sponge, loop image, context and loop stacks (decoder/sponge.rs, decoder/flow_ops.rs) -----------------------
    if constexpr ((SECT & 4) != 0) {
        fe sp[4];
        const fe f_begin = cff[1], f_tend = cff[2], f_fend = cff[3], f_loop = cff[4], f_wrap = cff[5], f_break = cff[6], f_void = cff[7];
        {
            // every sponge constraint is a sum of four flag * value products, reduced once (fe_acc):
            //   HACC (sponge.rs:10-43) | cleared by BEGIN, LOOP, WRAP | copied by BREAK, VOID | TEND / FEND merge hashes
            fe os[4], ns[4];
#pragma unroll
            for (int i = 0; i < 4; i++) os[i] = fe_cube(fe_add(c_sp[i], per[i]));
            matmul<4>(os, c_sponge_mds);
            // op_code = ld0 + 2 ld1 + 4 ld2 + 8 ld3 + 16 ld4 + 32 hd0 + 64 hd1 (sponge.rs:27-33) as a doubling chain: six doublings and six
            // additions (144 instructions) instead of five multiplications by small constants
            fe op_code = hd[1];
            op_code = fe_add(fe_double(op_code), hd[0]);
            op_code = fe_add(fe_double(op_code), ld[4]);
            op_code = fe_add(fe_double(op_code), ld[3]);
            op_code = fe_add(fe_double(op_code), ld[2]);
            op_code = fe_add(fe_double(op_code), ld[1]);
            op_code = fe_add(fe_double(op_code), ld[0]);
            os[0] = fe_add(os[0], op_code);
            os[1] = fe_add(os[1], fe_mul(nw[0], hdf[0]));               // op_value = next.user_stack[0] * push_flag
#pragma unroll
            for (int i = 0; i < 4; i++) ns[i] = n_sp[i];
            matmul<4>(ns, c_sponge_inv_mds);
#pragma unroll
            for (int i = 0; i < 4; i++) ns[i] = fe_sub(fe_cube(ns[i]), per[4 + i]);
            const fe clr = fe_add(fe_add(f_begin, f_loop), f_wrap);
            const fe cpy = fe_add(f_break, f_void);
            const fe tf = fe_add(f_tend, f_fend);
            const fe xf[4] = {tf, f_tend, f_fend, tf};
            const fe xv[4] = {fe_sub(c_ctx[0], n_sp[0]), fe_sub(c_sp[0], n_sp[1]), fe_sub(c_sp[0], n_sp[2]), n_sp[3]};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                fe_acc A; fe_acc_zero(A);
                fe_acc_mac(A, cff[0], fe_sub(os[i], ns[i]));
                fe_acc_mac(A, clr, n_sp[i]);
                fe_acc_mac(A, cpy, fe_sub(c_sp[i], n_sp[i]));
                fe_acc_mac(A, xf[i], xv[i]);
                sp[i] = fe_acc_reduce(A);
            }
        }
        acc.emit(15, 3, sp[0]); acc.emit(16, 4, sp[1]); acc.emit(17, 3, sp[2]); acc.emit(18, 3, sp[3]);
    }
    if constexpr ((SECT & 128) != 0) {
        const fe f_begin = cff[1], f_tend = cff[2], f_fend = cff[3], f_loop = cff[4], f_wrap = cff[5], f_break = cff[6], f_void = cff[7];
        // loop image (WRAP, BREAK)
        acc.emit(19, 2, fe_mul(fe_add(f_wrap, f_break), fe_sub(c_sp[0], c_lp[0])));
        // context stack: BEGIN/LOOP push sponge[0]; TEND/FEND pop; WRAP/BREAK/VOID copy
        // loop stack: BEGIN/TEND/FEND/WRAP/VOID copy; LOOP shifts right (slot 0 unconstrained); BREAK pops
        const fe push = fe_add(f_begin, f_loop), pop = fe_add(f_tend, f_fend), ccpy = fe_add(fe_add(f_wrap, f_break), f_void);
        const fe lcpy = fe_add(fe_add(fe_add(f_begin, f_tend), fe_add(f_fend, f_wrap)), f_void);
        if constexpr (CL <= 2 && LL <= 1) {
            // small shapes: the slices are register arrays with constant indices
#pragma unroll
            for (int i = 0; i < CL; i++) {
                if (i >= cl) break;
                fe_acc A; fe_acc_zero(A);
                fe_acc_mac(A, push, fe_sub(i == 0 ? c_sp[0] : c_ctx[i - 1 < 0 ? 0 : i - 1], n_ctx[i]));
                fe_acc_mac(A, pop, i + 1 < cl ? fe_sub(c_ctx[i + 1 < CL ? i + 1 : 0], n_ctx[i]) : n_ctx[i]);
                fe_acc_mac(A, ccpy, fe_sub(c_ctx[i], n_ctx[i]));
                const fe v = fe_acc_reduce(A);
                acc.emit(20 + i, 2, v);
            }
#pragma unroll
            for (int i = 0; i < LL; i++) {
                if (i >= ll) break;
                fe v = fe_mul(lcpy, fe_sub(c_lp[i], n_lp[i]));
                if (i >= 1) v = fe_add(v, fe_mul(f_loop, fe_sub(c_lp[i - 1 < 0 ? 0 : i - 1], n_lp[i])));
                v = fe_add(v, fe_mul(f_break, i + 1 < ll ? fe_sub(c_lp[i + 1 < LL ? i + 1 : 0], n_lp[i]) : n_lp[i]));
                acc.emit(20 + cl + i, 2, v);
            }
        } else {
            // any depth (up to 16 + 8 registers, known at run time): the rows are read from memory inside run-time loops -- register arrays
            // with run-time indices would live in scratch.  A slice of depth 0 is one zero (trace_state.rs:58-59).
            auto ctx_cur = [&](int i) { return (uint32_t)i < a.ctx_depth ? CUR(15 + i) : fe_zero(); };
            auto ctx_nxt = [&](int i) { return (uint32_t)i < a.ctx_depth ? NXT(15 + i) : fe_zero(); };
            const uint32_t lcol = 15 + a.ctx_depth;
            auto lp_cur = [&](int i) { return (uint32_t)i < a.loop_depth ? CUR(lcol + i) : fe_zero(); };
            auto lp_nxt = [&](int i) { return (uint32_t)i < a.loop_depth ? NXT(lcol + i) : fe_zero(); };
#pragma unroll 1
            for (int i = 0; i < cl; i++) {
                const fe ni = ctx_nxt(i);
                fe_acc A; fe_acc_zero(A);
                fe_acc_mac(A, push, fe_sub(i == 0 ? c_sp[0] : ctx_cur(i - 1), ni));
                fe_acc_mac(A, pop, i + 1 < cl ? fe_sub(ctx_cur(i + 1), ni) : ni);
                fe_acc_mac(A, ccpy, fe_sub(ctx_cur(i), ni));
                acc.emit(20 + i, 2, fe_acc_reduce(A));
            }
#pragma unroll 1
            for (int i = 0; i < ll; i++) {
                const fe ni = lp_nxt(i);
                fe_acc A; fe_acc_zero(A);
                fe_acc_mac(A, lcpy, fe_sub(lp_cur(i), ni));
                if (i >= 1) fe_acc_mac(A, f_loop, fe_sub(lp_cur(i - 1), ni));
                fe_acc_mac(A, f_break, i + 1 < ll ? fe_sub(lp_cur(i + 1), ni) : ni);
                acc.emit(20 + cl + i, 2, fe_acc_reduce(A));
            }
        }
    }

    // ---- stack constraints (constraints/stack/mod.rs:117-195) ---------------------------------------------------------------
    constexpr bool NESTED = SD == 4 || (SD == 0 && SLCAP == 8) || DEEP;      // depth 4 exactly, any depth <= 8 (all 8 slots, `sd` of them emitted), or the first 8 slots of a deeper stack
    static_assert(NESTED ? (SECT & 120) == 0 : GROUPS == 0, "nested-sum instances select stack operations with GROUPS, the per-operation instance with SECT");
    if constexpr (NESTED && (GROUPS & AG_STACK_ALL & ~AG_DEEP) != 0) {
        // low-degree operations as nested sums over the merged terms of each group (st_output), high-degree / flow operations inside the
        // outermost sum; outputs: slots 0..NS-1, ST_AUX0, ST_AUX1
        constexpr int NS = SD == 4 ? 4 : 8;
        StackRows rows;
#pragma unroll
        for (int i = 0; i < 8; i++) { rows.o[i] = o[i]; rows.nw[i] = nw[i]; }
#pragma unroll
        for (int i = 8; i < 12; i++) rows.o[i] = DEEP ? o[i < SL ? i : 0] : fe_zero();
        rows.hd0 = hd[0];
        rows.sl = DEEP ? sl : 8;
        const fe lo0h = fe_mul(lo2[0], hd[0]);                       // ASSERT = LDF(0) * hd[0] (trace_state.rs:346)
        StackHigh high;
        high.push = hdf[0]; high.cmp = hdf[1]; high.rescr = hdf[2]; high.keep = fe_add(begin_flag, noop_flag);
        if constexpr ((GROUPS & AG_RESCR) != 0) {
            // hash.rs:9-35.  Items at and above a known depth are zeros: their round-constant cubes come from the periodic table's
            // values alone, and the inverse-matrix rows are shorter.
            constexpr int NZ = (SD != 0 && SD < 6) ? SD : 6;         // items that can be non-zero
            fe os[6], ns[6];
#pragma unroll
            for (int i = 0; i < 6; i++) os[i] = i < NZ ? fe_cube(fe_add(o[i], per[8 + i])) : per[23 + i];      // an absent item is zero: the constant's cube, from the table
            matmul<6>(os, c_hasher_mds);
            constexpr int NR = NS < 6 ? NS : 6;                      // differences that are used
#pragma unroll
            for (int i = 0; i < NR; i++) {
                fe_acc A; fe_acc_zero(A);
#pragma unroll
                for (int j = 0; j < NZ; j++) fe_acc_mac(A, c_hasher_inv_mds[i * 6 + j], nw[j]);
                ns[i] = fe_acc_reduce(A);
            }
#pragma unroll
            for (int i = 0; i < NR; i++) high.resc[i] = fe_sub(fe_sub(fe_cube(ns[i]), per[14 + i]), os[i]);
        }
        AIR_SYNC();
        const uint32_t sbase = 20 + cl + ll;
        static_for_air<0, NS + 2>([&](auto k_) {
            constexpr int kk = decltype(k_)::value;
            constexpr int I = kk < NS ? kk : (kk == NS ? ST_AUX0 : ST_AUX1);
            constexpr bool touched = st_touches<I, GROUPS>();
            // a launch that neither completes the constraints nor contributes to this one leaves the partial value where it is
            if constexpr (touched || !EV_OUT) {
                if (I >= 8 || I < sd) {
                    fe v = fe_zero();
                    if constexpr (touched) v = st_output<I, GROUPS, SD, DEEP>(rows, lo2, lo0h, mid, top, high);
                    fe* slot = a.ev[I] + pidx;
                    if constexpr (EV_IN) v = touched ? fe_add(v, *slot) : *slot;
                    if constexpr (EV_OUT) *slot = v;
                    else acc.emit(I >= 8 ? sbase + (I - 8) : sbase + 2 + I, 4, v);
                }
            } else if constexpr (!EV_IN) {
                a.ev[I][pidx] = fe_zero();                           // the first stack launch defines every partial value
            }
            AIR_SYNC();
        });
    }
    if constexpr (!NESTED && (SECT & 120) != 0) {
        constexpr int SLR = SL;                  // slots whose constraints are formed in registers
        fe ev[SLR], aux0 = fe_zero(), aux1 = fe_zero();
#pragma unroll
        for (int i = 0; i < SLR; i++) ev[i] = fe_zero();
        auto agg = [&](int i, const fe& f, const fe& v) { if (i < sd) ev[i] = fe_add(ev[i], fe_mul(f, v)); };
        auto copy_from = [&](int from, const fe& f) {
#pragma unroll
            for (int i = 0; i < SLR; i++) if (i >= from && i < sl) agg(i, f, fe_sub(o[i], nw[i]));
        };
        auto rshift = [&](int num, const fe& f) {
#pragma unroll
            for (int i = 0; i < SLR; i++) if (i >= num && i < sl) agg(i, f, fe_sub(o[i - num < 0 ? 0 : i - num], nw[i]));
        };
        auto lshift = [&](int from, int num, const fe& f) {
#pragma unroll
            for (int i = 0; i < SLR; i++) {
                if (i >= from - num && i < sl - num) agg(i, f, fe_sub(o[i + num < SL ? i + num : 0], nw[i]));
                else if (i >= sl - num && i < sl) agg(i, f, nw[i]);
            }
        };
        fe f;
        constexpr bool HD_FUSED = false;
        if constexpr (!NESTED && (SECT & 8) != 0) {
        // flags that only shift / copy are merged before the multiplications
        // right shift by 1: READ (0x10), DUP (0x12)  (PUSH is with the high-degree ops)
        f = LDF(0x12); agg(0, f, fe_sub(nw[0], o[0]));
        rshift(1, fe_add(LDF(0x10), f));
        // right shift by 2: READ2 (0x11), DUP2 (0x13), PAD2 (0x15)
        f = LDF(0x13); agg(0, f, fe_sub(nw[0], o[0])); agg(1, f, fe_sub(nw[1], o[1]));
        f = LDF(0x15); agg(0, f, nw[0]); agg(1, f, nw[1]);
        rshift(2, fe_add(fe_add(LDF(0x11), LDF(0x13)), LDF(0x15)));
        // DUP4 (0x14)
        f = LDF(0x14);
#pragma unroll
        for (int i = 0; i < 4; i++) agg(i, f, fe_sub(nw[i], o[i]));
        rshift(4, f);
        // left shift (1,1): ASSERT (0x00), DROP (0x03)
        lshift(1, 1, fe_add(assert_flag, LDF(0x03)));
        aux0 = fe_add(aux0, fe_mul(assert_flag, fe_sub(fe_one(), o[0])));
        // ASSERTEQ (0x01): left shift (2,2)
        lshift(2, 2, LDF(0x01));
        aux0 = fe_add(aux0, fe_mul(LDF(0x01), fe_sub(o[0], o[1])));
        // DROP4 (0x04)
        lshift(4, 4, LDF(0x04));
        // SWAP (0x18): both constraints land in slot 0 (manipulation.rs:63-64)
        f = LDF(0x18); agg(0, f, fe_sub(nw[0], o[1])); agg(0, f, fe_sub(nw[1], o[0])); copy_from(2, f);
        // SWAP2 (0x19)
        f = LDF(0x19); agg(0, f, fe_sub(nw[0], o[2])); agg(1, f, fe_sub(nw[1], o[3])); agg(2, f, fe_sub(nw[2], o[0])); agg(3, f, fe_sub(nw[3], o[1])); copy_from(4, f);
        // SWAP4 (0x1A)
        f = LDF(0x1A);
#pragma unroll
        for (int i = 0; i < 4; i++) { agg(i, f, fe_sub(nw[i], o[4 + i])); agg(4 + i, f, fe_sub(nw[4 + i], o[i])); }
        copy_from(8, f);
        // ROLL4 (0x1B), ROLL8 (0x1C)
        f = LDF(0x1B); agg(0, f, fe_sub(nw[0], o[3]));
#pragma unroll
        for (int i = 1; i < 4; i++) agg(i, f, fe_sub(nw[i], o[i - 1]));
        copy_from(4, f);
        f = LDF(0x1C); agg(0, f, fe_sub(nw[0], o[7]));
#pragma unroll
        for (int i = 1; i < 8; i++) agg(i, f, fe_sub(nw[i], o[i - 1]));
        copy_from(8, f);
        }   // low-degree ops that only move stack items
        if constexpr (!NESTED && (SECT & 32) != 0) {
        // ADD (0x08), MUL (0x09), AND (0x0A), OR (0x0B): left shift (2,1)
        {
            fe xy = fe_mul(o[0], o[1]);
            agg(0, LDF(0x08), fe_sub(nw[0], fe_add(o[0], o[1])));
            agg(0, fe_add(LDF(0x09), LDF(0x0A)), fe_sub(nw[0], xy));
            agg(0, LDF(0x0B), fe_sub(nw[0], bnot(fe_mul(bnot(o[0]), bnot(o[1])))));
            lshift(2, 1, fe_add(fe_add(LDF(0x08), LDF(0x09)), fe_add(LDF(0x0A), LDF(0x0B))));
            fe b0 = is_bin(o[0]), b1 = is_bin(o[1]);
            fe andor = fe_add(LDF(0x0A), LDF(0x0B));
            aux0 = fe_add(aux0, fe_mul(fe_add(andor, LDF(0x0E)), b0));            // NOT also checks operand 0
            aux1 = fe_add(aux1, fe_mul(andor, b1));
        }
        // INV (0x0C), NEG (0x0D), NOT (0x0E): copy from 1
        agg(0, LDF(0x0C), fe_sub(fe_one(), fe_mul(nw[0], o[0])));
        agg(0, LDF(0x0D), fe_add(nw[0], o[0]));
        agg(0, LDF(0x0E), fe_sub(nw[0], bnot(o[0])));
        copy_from(1, fe_add(fe_add(LDF(0x0C), LDF(0x0D)), LDF(0x0E)));
        // EQ (0x02)
        {
            f = LDF(0x02);
            fe diff = fe_sub(o[1], o[2]);
            agg(0, f, fe_sub(nw[0], bnot(fe_mul(diff, o[0]))));
            lshift(3, 2, f);
            aux0 = fe_add(aux0, fe_mul(f, fe_mul(nw[0], diff)));
        }
        // BINACC (0x1D)
        {
            f = LDF(0x1D);
            agg(0, f, is_bin(nw[0])); agg(1, f, nw[1]);
            agg(2, f, fe_sub(nw[2], fe_double(o[2])));
            agg(3, f, fe_sub(nw[3], fe_add(o[3], fe_mul(nw[0], o[2]))));
            copy_from(4, f);
        }
        // CHOOSE (0x05), CHOOSE2 (0x06), CSWAP2 (0x07)
        {
            f = LDF(0x05);
            fe c = o[2], nc = bnot(c);
            agg(0, f, fe_sub(nw[0], fe_add(fe_mul(c, o[0]), fe_mul(nc, o[1]))));
            lshift(3, 2, f);
            aux0 = fe_add(aux0, fe_mul(f, is_bin(c)));
            c = o[4]; nc = bnot(c);
            fe binc = is_bin(c);
            f = LDF(0x06);
            agg(0, f, fe_sub(nw[0], fe_add(fe_mul(c, o[0]), fe_mul(nc, o[2]))));
            agg(1, f, fe_sub(nw[1], fe_add(fe_mul(c, o[1]), fe_mul(nc, o[3]))));
            lshift(6, 4, f);
            aux0 = fe_add(aux0, fe_mul(f, binc));
            f = LDF(0x07);
            agg(0, f, fe_sub(nw[0], fe_add(fe_mul(c, o[2]), fe_mul(nc, o[0]))));
            agg(1, f, fe_sub(nw[1], fe_add(fe_mul(c, o[3]), fe_mul(nc, o[1]))));
            agg(2, f, fe_sub(nw[2], fe_add(fe_mul(c, o[0]), fe_mul(nc, o[2]))));
            agg(3, f, fe_sub(nw[3], fe_add(fe_mul(c, o[1]), fe_mul(nc, o[3]))));
            lshift(6, 2, f);
            aux0 = fe_add(aux0, fe_mul(f, binc));
        }
        }   // low-degree arithmetic / selection ops
        if constexpr (!HD_FUSED && (SECT & 16) != 0) {
        rshift(1, hdf[0]);                                             // PUSH (input.rs:6)
        // CMP (hd 1) (comparison.rs:64-105)
        {
            f = hdf[1];
            fe x_bit = nw[1], y_bit = nw[2], not_set = nw[3];
            agg(0, f, is_bin(x_bit)); agg(1, f, is_bin(y_bit));
            fe bit_gt = fe_mul(x_bit, bnot(y_bit)), bit_lt = fe_mul(y_bit, bnot(x_bit));
            agg(2, f, fe_sub(nw[4], fe_add(o[4], fe_mul(bit_gt, not_set))));
            agg(3, f, fe_sub(nw[5], fe_add(o[5], fe_mul(bit_lt, not_set))));
            agg(4, f, fe_sub(nw[6], fe_add(o[6], fe_mul(y_bit, o[0]))));
            agg(5, f, fe_sub(nw[7], fe_add(o[7], fe_mul(x_bit, o[0]))));
            agg(6, f, fe_sub(not_set, fe_mul(bnot(o[5]), bnot(o[4]))));
            agg(7, f, fe_sub(fe_double(nw[0]), o[0]));
            copy_from(8, f);
        }
        // BEGIN and NOOP leave the stack untouched
        copy_from(0, fe_add(begin_flag, noop_flag));
        }   // PUSH, CMP, BEGIN / NOOP
        if constexpr (!HD_FUSED && (SECT & 64) != 0) {
        // RESCR (hd 2) (hash.rs:9-35)
        {
            f = hdf[2];
            fe os[6], ns[6];
#pragma unroll
            for (int i = 0; i < 6; i++) os[i] = fe_cube(fe_add(o[i], per[8 + i]));
            matmul<6>(os, c_hasher_mds);
#pragma unroll
            for (int i = 0; i < 6; i++) ns[i] = nw[i];
            matmul<6>(ns, c_hasher_inv_mds);
#pragma unroll
            for (int i = 0; i < 6; i++) ns[i] = fe_sub(fe_cube(ns[i]), per[14 + i]);
#pragma unroll
            for (int i = 0; i < 6; i++) agg(i, f, fe_sub(ns[i], os[i]));
            copy_from(6, f);
        }
        }   // RESCR

        const uint32_t sbase = 20 + cl + ll;
        if constexpr ((SECT & 40) != 0) { acc.emit(sbase, 4, aux0); acc.emit(sbase + 1, 4, aux1); }    // only low-degree ops touch the aux constraints
#pragma unroll
        for (int i = 0; i < SLR; i++) if (i < sd) acc.emit(sbase + 2 + i, 4, ev[i]);
    }
    if constexpr (DEEP && (GROUPS & AG_DEEP) != 0) {
        {
            const uint32_t sbase = 20 + cl + ll;
            // Slots 8 .. sd-1 of a deep stack.  No operation computes anything there: a slot is copied, or takes the item 1, 2 or 4
            // places to its left or right (left shifts zero-fill the end of the slice), constraints/stack/mod.rs:117-195 with the
            // shift helpers :199-262.  So the constraint is a sum of seven products, flag sum * (shifted old item - new item), and
            // the seven flag sums are formed once per point from the three small tables (never the 32 flags):
            //   left by 1: ASSERT DROP ADD MUL AND OR | by 2: ASSERTEQ EQ CHOOSE CSWAP2 | by 4: DROP4 CHOOSE2
            //   right by 1: READ DUP PUSH | by 2: READ2 DUP2 PAD2 | by 4: DUP4
            //   copy: INV NEG NOT SWAP SWAP2 SWAP4 ROLL4 ROLL8 BINACC CMP RESCR BEGIN NOOP
            const fe lo_all = fe_add(fe_add(lo2[0], lo2[1]), fe_add(lo2[2], lo2[3]));
            fe fl1, fl2, fl4, fr1, fr2, fr4, fcp;
            { fe_acc A; fe_acc_zero(A); fe_acc_mac(A, mid[0], lo2[3]); fe_acc_mac(A, mid[2], lo_all); fl1 = fe_add(fe_mul(top[0], fe_acc_reduce(A)), assert_flag); }
            { fe_acc A; fe_acc_zero(A); fe_acc_mac(A, mid[0], fe_add(lo2[1], lo2[2])); fe_acc_mac(A, mid[1], fe_add(lo2[1], lo2[3])); fl2 = fe_mul(top[0], fe_acc_reduce(A)); }
            fl4 = fe_mul(top[0], fe_mul(mid[1], fe_add(lo2[0], lo2[2])));
            fr1 = fe_add(fe_mul(top[1], fe_mul(mid[0], fe_add(lo2[0], lo2[2]))), hdf[0]);
            { fe_acc A; fe_acc_zero(A); fe_acc_mac(A, mid[0], fe_add(lo2[1], lo2[3])); fe_acc_mac(A, mid[1], lo2[1]); fr2 = fe_mul(top[1], fe_acc_reduce(A)); }
            fr4 = fe_mul(top[1], fe_mul(mid[1], lo2[0]));
            {
                fe_acc A; fe_acc_zero(A); fe_acc_mac(A, mid[2], lo_all); fe_acc_mac(A, mid[3], fe_add(lo2[0], lo2[1]));
                fe_acc B; fe_acc_zero(B); fe_acc_mac(B, top[1], fe_acc_reduce(A)); fe_acc_mac(B, top[0], fe_mul(mid[3], fe_add(fe_add(lo2[0], lo2[1]), lo2[2])));
                fcp = fe_add(fe_add(fe_acc_reduce(B), fe_add(hdf[1], hdf[2])), fe_add(begin_flag, noop_flag));
            }
            const uint32_t scol = 15 + a.ctx_depth + a.loop_depth;
#pragma unroll 1
            for (int i = 8; i < sd; i++) {
                const fe ni = NXT(scol + i);
                auto item = [&](int j) { return CUR(scol + j); };              // j in [4, sd) here
                auto left = [&](int num) { return i + num < sl ? fe_sub(item(i + num), ni) : ni; };
                fe_acc A; fe_acc_zero(A);
                fe_acc_mac(A, fcp, fe_sub(item(i), ni));
                fe_acc_mac(A, fr1, fe_sub(item(i - 1), ni)); fe_acc_mac(A, fr2, fe_sub(item(i - 2), ni)); fe_acc_mac(A, fr4, fe_sub(item(i - 4), ni));
                fe_acc_mac(A, fl1, left(1)); fe_acc_mac(A, fl2, left(2)); fe_acc_mac(A, fl4, left(4));
                acc.emit(sbase + 2 + i, 4, fe_acc_reduce(A));
            }
        }
    }

    // ---- combination (evaluator.rs:139-162, 335-358) ---------------------------------------------------------------------------
    if constexpr (SECT == 1) return;           // a boundary-only launch neither reads nor writes the partial sums
    const bool on_trace = (qg == 0);
    if (on_trace && k + 1 != a.n && acc.nonzero) atomicMin(a.bad_step, (unsigned long long)k);      // evaluator.rs:152-158
    // the partial sums of the previous launches join at the end (they are not live during the evaluation); a launch only moves the
    // degree-group sums its sections emit into: op bits {2,3,4,6,8}, sponge / context / loop {4,6,7}, stack {7}
    acc.res = fe_acc_reduce(acc.res_acc);
    constexpr bool EMITS_STACK = (SECT & 120) != 0 || (GROUPS & AG_DEEP) != 0 || ((GROUPS & AG_STACK_ALL) != 0 && !EV_OUT);
    constexpr uint32_t USED = ((SECT & 2) ? 0x2Fu : 0u) | ((SECT & 4) ? 0x18u : 0u) | ((SECT & 128) ? 0x04u : 0u) | (EMITS_STACK ? 0x10u : 0u);
#pragma unroll
    for (int i = 0; i < 6; i++) if (((ACC_MASK & USED) >> i) & 1u) acc.adj[i] = fe_add(acc.adj[i], fe_acc_reduce(acc.adj_acc[i]));
    if constexpr (!FIRST) {
        acc.res = fe_add(acc.res, a.partial[pidx]);
#pragma unroll
        for (int i = 0; i < 6; i++) if (LAST || ((USED >> i) & 1u)) acc.adj[i] = fe_add(acc.adj[i], a.partial[(size_t)(i + 1) * pstride + pidx]);
    }
    if constexpr (!LAST) {
        a.partial[pidx] = acc.res;
#pragma unroll
        for (int i = 0; i < 6; i++) if (FIRST || ((USED >> i) & 1u)) a.partial[(size_t)(i + 1) * pstride + pidx] = acc.adj[i];
        return;
    }
    fe t;
    if (on_trace && k + 1 != a.n) {
        t = fe_zero();
    } else {
        // x^p with p = 8n - 1 - (n - 1) * degree for degrees 2, 3, 4, 6, 7, 8: res + sum_g adj[g] * x^p_g as ONE sum of products
        const uint32_t degs[6] = {2, 3, 4, 6, 7, 8};
        fe_acc T; fe_acc_zero(T);
        fe_acc_add(T, acc.res);
#pragma unroll
        for (int g = 0; g < 6; g++) {
            uint64_t p = 8 * n64 - 1 - (n64 - 1) * degs[g];
            fe_acc_mac(T, acc.adj[g], dpow(a, (gi * p) & nmask));
        }
        t = fe_acc_reduce(T);
    }
    a.out[((size_t)2 * gridDim.y + ql) * a.n + k] = t;
#undef CUR
#undef NXT
}


template <int CL, int LL, int SD, int SLCAP, int SECT, uint32_t GROUPS, int FLAGS>
static void launch_air(dst_ctx* c, const AirArgs& a, uint32_t Q) {
    dim3 g((unsigned)((c->n + AIR_THREADS - 1) / AIR_THREADS), Q);
    static char name[64];          // one name per instantiation, matching the template arguments rocprofv3 prints
    if (!name[0]) snprintf(name, sizeof(name), "air_kernel<%d,%d,%d,%d,%d,%u,%d>", CL, LL, SD, SLCAP, SECT, (unsigned)GROUPS, FLAGS);
    // algorithmic bytes: the W registers of a row, the partial sums that come in and go out, the stack partial values, the result
    constexpr int NS = SD == 4 ? 6 : 10;
    const double units = c->W + ((FLAGS & AF_FIRST) ? 0 : 7) + ((FLAGS & AF_LAST) ? 1 : 7) + ((SECT & 1) ? 2 : 0) + ((FLAGS & AF_EV_IN) ? NS : 0) + ((FLAGS & AF_EV_OUT) ? NS : 0);
    { KScope ks_(c, name, 16.0 * c->n * Q * units, true); hipLaunchKernelGGL((air_kernel<CL, LL, SD, SLCAP, SECT, GROUPS, FLAGS>), g, dim3(AIR_THREADS), 0, c->stream, a); }
}
// instances live in their own translation units (compile time): Fibonacci shape, small stacks, fully generic
void air_launch_sd4(dst_ctx* c, const AirArgs& a, uint32_t Q);      // cl <= 2, ll <= 1, stack_depth == 4
void air_launch_small(dst_ctx* c, const AirArgs& a, uint32_t Q);    // cl <= 2, ll <= 1, stack_depth <= 8
void air_launch_deep(dst_ctx* c, const AirArgs& a, uint32_t Q);     // anything the VM can produce, nested sums + flag sums for slots >= 8
void air_launch_generic(dst_ctx* c, const AirArgs& a, uint32_t Q);  // anything the VM can produce, per-operation formulation (DISTAFF_AIR=generic)
