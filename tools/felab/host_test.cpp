// CPU check of the limb-level dataflow of fe2.h against the portable multiplication (itself pinned by the oracle tests)
#include <stdio.h>
#include <stdlib.h>
#define FE_EMULATE_GFX950 1
#include "../../distaff_amd/csrc/fe.h"
static uint64_t s = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static fe special(int k) {
    switch (k % 12) {
        case 0: return fe_zero(); case 1: return fe_one(); case 2: return fe_make(0, 0xFFFFD300u, 0xFFFFFFFFu, 0xFFFFFFFFu); // p - 1
        case 3: return fe_make(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu); case 4: return fe_make(1, 0xFFFFD300u, 0xFFFFFFFFu, 0xFFFFFFFFu); // p
        case 5: return fe_make(0xFFFFFFFFu, 0, 0, 0); case 6: return fe_make(0, 0, 0, 0xFFFFFFFFu); case 7: return fe_make(0xFFFFFFFFu, 0x2CFFu, 0, 0);
        case 8: return fe_make(0, 0, 0xFFFFFFFFu, 0xFFFFFFFFu); case 9: return fe_make(0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0);
        case 10: return fe_make(0xFFFFFFFEu, 0xFFFFD2FFu, 0xFFFFFFFFu, 0xFFFFFFFFu); default: return fe_make(2, 0xFFFFD300u, 0xFFFFFFFFu, 0xFFFFFFFFu);
    }
}
typedef unsigned __int128 u128;
static u128 to128(fe a) { return ((u128)a.v[3] << 96) | ((u128)a.v[2] << 64) | ((u128)a.v[1] << 32) | a.v[0]; }
static fe from128(u128 x) { return fe_make((uint32_t)x, (uint32_t)(x >> 32), (uint32_t)(x >> 64), (uint32_t)(x >> 96)); }
static const u128 PP = (((u128)0xFFFFFFFFFFFFFFFFull) << 64) | 0xFFFFD30000000001ull;
static fe ref_add(fe a, fe b) { u128 x = to128(a), y = to128(b); u128 z = PP - y; return from128(x >= z ? x - z : x + y); }
static fe ref_sub(fe a, fe b) { u128 x = to128(a), y = to128(b); return from128(x >= y ? x - y : x + (PP - y)); }
static fe canon(fe a) { // reduce an arbitrary 128-bit value
    uint32_t t[8] = {a.v[0], a.v[1], a.v[2], a.v[3], 0, 0, 0, 0}; return fe_reduce8(t);
}
int main() {
    long bad = 0;
    for (long it = 0; it < 4000000; it++) {
        fe a, w;
        if (it < 144) { a = special(it / 12); w = canon(special(it % 12)); }
        else {
            a = fe_make(rnd(), rnd(), rnd(), rnd()); w = canon(fe_make(rnd(), rnd(), rnd(), rnd()));
            if ((it & 7) == 1) a.v[3] = a.v[2] = 0xFFFFFFFFu;
            if ((it & 7) == 2) { w.v[3] = w.v[2] = 0xFFFFFFFFu; w = canon(w); }
            if ((it & 15) == 3) { a = fe_make(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu); }
        }
        fe ref = fe_mul_portable(canon(a), w);
        fe q = fe_shift64(w);
        fe r1 = fe_mul_tw(a, w, q);
        fe r2 = fe_mul_wide(a, w);
        fe r3 = fe_mul_wide(a, fe_make(rnd(), rnd(), 0xFFFFFFFFu, 0xFFFFFFFFu));   // non-canonical second operand
        fe ac = canon(a);
        if (!fe_eq(r1, ref) || !fe_eq(r2, ref)) { if (bad < 5) printf("mismatch it=%ld\n", it); bad++; }
        (void)r3;
        if (!fe_eq(fe_add(ac, w), ref_add(ac, w)) || !fe_eq(fe_sub(ac, w), ref_sub(ac, w))) { if (bad < 5) printf("add/sub mismatch it=%ld\n", it); bad++; }
    }
    // non-canonical operands of the general multiplication
    for (long it = 0; it < 1000000; it++) {
        fe a = fe_make(rnd(), rnd(), (it & 1) ? 0xFFFFFFFFu : rnd(), 0xFFFFFFFFu), b = fe_make(rnd(), rnd(), (it & 2) ? 0xFFFFFFFFu : rnd(), (it & 4) ? 0xFFFFFFFFu : rnd());
        if (!fe_eq(fe_mul_wide(a, b), fe_mul_portable(canon(a), canon(b)))) { if (bad < 5) printf("weak mismatch it=%ld\n", it); bad++; }
    }
    // the nine-limb reduction with its overflow limb (sums of products: fe_acc_reduce on the device)
    const fe c128 = fe_make(FE_C0, FE_C1, 0, 0), c256 = fe_mul_portable(c128, c128);
    for (long it = 0; it < 1000000; it++) {
        uint32_t t[9];
        for (int i = 0; i < 8; i++) t[i] = (it & (1 << i)) && (it & 0x300) == 0x300 ? 0xFFFFFFFFu : (uint32_t)rnd();
        t[8] = (uint32_t)(rnd() & 127);
        if ((it & 0xC00) == 0xC00) t[8] = 127;
        fe ref = fe_add(fe_reduce8(t), fe_mul_portable(fe_make(t[8], 0, 0, 0), c256));
        if (!fe_eq(fe_fold9(t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8], true), ref)) { if (bad < 5) printf("fold9 mismatch it=%ld\n", it); bad++; }
        fe x = fe_make(rnd(), rnd(), rnd(), rnd()), y = fe_make(rnd(), rnd(), rnd(), rnd()), sum, dif;
        x = canon(x); y = canon(y);
        if (it & 1) y = special(it >> 1);
        if ((it & 6) == 6) x = special(it >> 3);
        x = canon(x); y = canon(y);
        fe_addsub(x, y, sum, dif);
        if (!fe_eq(sum, ref_add(x, y)) || !fe_eq(dif, ref_sub(x, y))) { if (bad < 5) printf("addsub mismatch it=%ld\n", it); bad++; }
    }
    // sums of products with one reduction: up to 64 terms, operands any 128-bit values
    for (long it = 0; it < 20000; it++) {
        fe_acc A; fe_acc_zero(A);
        fe ref = fe_zero();
        const int terms = 1 + (int)(rnd() % 63);
        for (int i = 0; i < terms; i++) {
            fe x = fe_make(rnd(), rnd(), rnd(), rnd()), y = fe_make(rnd(), rnd(), rnd(), rnd());
            if ((it & 3) == 3) { x = fe_make(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu); y = x; }
            if (i & 1) { fe_acc_mac(A, x, y); ref = fe_add(ref, fe_mul_portable(canon(x), canon(y))); }
            else { fe_acc_mac(A, x, y); fe_acc_add(A, y); ref = fe_add(fe_add(ref, fe_mul_portable(canon(x), canon(y))), canon(y)); }
        }
        if (!fe_eq(fe_acc_reduce(A), ref)) { if (bad < 5) printf("acc mismatch it=%ld terms=%d\n", it, terms); bad++; }
    }
    printf("bad=%ld\n", bad);
    return bad != 0;
}
