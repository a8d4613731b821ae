// TEST INFRASTRUCTURE: prints products, squares and inverse-S-box values computed by the library's host field (distaff_amd/csrc/host_vm.h:
// hf_mul, hf_sqr, inv_alpha4, hf_pow) for edge and seeded random operands; tests/test_host_logic.py compares them with Python integers.
#include <stdio.h>
#include <stdlib.h>
#define HF_COUNT_OPS 1
#include "host_vm.h"
using namespace dsth;
static uint64_t state = 0x9E3779B97F4A7C15ull;
static uint64_t next64() { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return state; }
static u128 rnd() { return (((u128)next64() << 64) | next64()) % FIELD_P; }
static void pr(u128 v) { printf("%016llx%016llx", (unsigned long long)(v >> 64), (unsigned long long)v); }
int main() {
    const u128 edge[] = {0, 1, 2, 3, FIELD_P - 1, FIELD_P - 2, (u128)1 << 127, (~(u128)0) % FIELD_P, (u128)1 << 64, ((u128)1 << 64) - 1,
                         FIELD_P - HF_C, HF_C, (u128)HF_C + 1, ((u128)0xFFFFFFFFFFFFFFFFull << 64), FIELD_P >> 1, (FIELD_P >> 1) + 1};
    const int ne = sizeof(edge) / sizeof(edge[0]);
    for (int i = 0; i < ne; i++) for (int j = 0; j < ne; j++) {
        printf("m "); pr(edge[i]); printf(" "); pr(edge[j]); printf(" "); pr(hf_mul(edge[i], edge[j])); printf(" "); pr(hf_sqr(edge[i])); printf("\n");
    }
    for (int i = 0; i < 4000; i++) {
        u128 a = rnd(), b = rnd();
        printf("m "); pr(a); printf(" "); pr(b); printf(" "); pr(hf_mul(a, b)); printf(" "); pr(hf_sqr(a)); printf("\n");
    }
    for (int i = 0; i < 400; i++) {
        u128 s[4], t[4];
        for (int l = 0; l < 4; l++) t[l] = s[l] = (i * 4 + l) < ne ? edge[i * 4 + l] : rnd();
        inv_alpha4(t);
        for (int l = 0; l < 4; l++) { printf("p "); pr(s[l]); printf(" "); pr(t[l]); printf(" "); pr(hf_pow(s[l], HF_INV_ALPHA)); printf("\n"); }
    }
    { u128 one[4] = {2, 3, 5, 7}; hf_count_sqr = hf_count_mul = 0; inv_alpha4(one); printf("c %x %x\n", hf_count_sqr, hf_count_mul); }      // operations of ONE chain
    return 0;
}
