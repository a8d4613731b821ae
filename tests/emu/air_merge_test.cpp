// TEST INFRASTRUCTURE (host build against the stand-in HIP header of this directory): the merged formulation of the low-degree stack
// operations (air_kernel.h: st_desc / st_merge / st_output -- operations of a group whose terms for a slot coincide up to sign share one
// product, items above the compile-time stack depth are zeros) against the plain sum  sum_op flag(op) * st_term(op, slot)  of the
// per-operation statement it was derived from, on random rows and random (not one-hot) flag factors, for every slot and auxiliary
// constraint, every compile-time stack depth and the run-time-depth instances.  The proof-level tests cannot see a wrong table entry of an
// operation whose flag is zero in the tested programs; this one can.
#include <stdio.h>
#include <stdlib.h>
#include "../../distaff_amd/csrc/air_kernel.h"

static uint64_t rs = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }
static fe rfe() { return fe_make((uint32_t)rnd(), (uint32_t)rnd(), (uint32_t)rnd(), (uint32_t)rnd() & 0x7FFFFFFFu); }   // below the modulus

template <int OP, int I>
static fe plain_term(const StackRows& s) {
    if constexpr (st_has<OP, I>()) return st_term<OP, I>(s);
    else return fe_zero();
}
template <int I>
static fe plain_sum(const StackRows& s, const fe* lo2, const fe& lo0h, const fe* mid, const fe* top) {
    fe total = fe_zero();
    static_for_air<0, 32>([&](auto op_) {
        constexpr int op = decltype(op_)::value;
        const fe cf = op == 0 ? lo0h : lo2[op & 3];
        const fe flag = fe_mul(fe_mul(cf, mid[(op >> 2) & 3]), top[op >> 4]);
        total = fe_add(total, fe_mul(flag, plain_term<op, I>(s)));
    });
    return total;
}
static long bad = 0, checked = 0;
template <int SDK, bool DEEP>
static void run(int sl, const char* what) {
    for (int it = 0; it < 200; it++) {
        StackRows s{};
        for (int j = 0; j < 12; j++) s.o[j] = rfe();
        for (int j = 0; j < 8; j++) s.nw[j] = rfe();
        s.hd0 = rfe(); s.sl = sl;
        if (it % 5 == 0) { s.o[0] = fe_make(it & 1, 0, 0, 0); s.o[1] = fe_make((it >> 1) & 1, 0, 0, 0); s.o[2] = s.o[4] = fe_make((it >> 2) & 1, 0, 0, 0); }   // binary operands now and then
        if (SDK != 0) for (int j = SDK; j < 12; j++) s.o[j] = fe_zero();        // items above the stack depth are zeros (trace_state.rs:58-60 pads the slice)
        if (!DEEP || sl <= 8) for (int j = 8; j < 12; j++) s.o[j] = fe_zero();
        fe lo2[4], mid[4], top[2];
        for (auto& v : lo2) v = rfe();
        for (auto& v : mid) v = rfe();
        for (auto& v : top) v = rfe();
        const fe lo0h = rfe();
        StackHigh h{};
        static_for_air<0, 10>([&](auto i_) {
            constexpr int I = decltype(i_)::value;                  // slots 0..7, ST_AUX0 = 8, ST_AUX1 = 9
            if (SDK != 0 && I < 8 && I >= SDK && !DEEP) return;     // constraints of slots above the depth are never emitted (stack/mod.rs:194)
            const fe got = st_output<I, (AG_LOW0 | AG_LOW1), SDK, DEEP>(s, lo2, lo0h, mid, top, h);
            const fe want = plain_sum<I>(s, lo2, lo0h, mid, top);
            checked++;
            if (!fe_eq(got, want)) { if (bad < 10) printf("MISMATCH %s: slot %d, iteration %d\n", what, I, it); bad++; }
        });
    }
}
int main() {
    run<4, false>(8, "depth 4"); run<5, false>(8, "depth 5"); run<6, false>(8, "depth 6"); run<7, false>(8, "depth 7"); run<8, false>(8, "depth 8");
    run<0, false>(8, "run-time depth, slice of 8");
    run<0, true>(8, "deep instance, slice of 8"); run<0, true>(10, "deep instance, slice of 10"); run<0, true>(12, "deep instance, slice of 12");
    printf("%ld comparisons, %ld mismatches\n", checked, bad);
    return bad ? 1 : 0;
}
