// GPU laboratory (round 5): two OTHER number representations for the in-LDS butterfly rounds, measured against the product's
// 4 x 32-bit canonical limbs with carries in SGPR lane masks (fe.h: fe_mul_tw 55 VALU instructions of which 18 v_mad_u64_u32, add 12, sub 10).
//
//   (i)  reduced radix, integers: 5 limbs of 26 bits, lazy (no canonical form inside a tile).  A table entry holds w * 2^(26 i) mod p for
//        i < 5 as 5 limbs each, so x * w = sum_i x_i * W_i needs no reduction of high columns: 25 v_mad_u64_u32 into five 64-bit column
//        sums WITHOUT carries, then carry propagation with shifts, the fold of the bits above 2^128 and a second short propagation.
//   (ii) FP64: 3 limbs of 43 bits held in doubles (v_fma_f64 is full rate on CDNA4).  Table entry: w * 2^(43 i) mod p, i < 3, 3 limbs each,
//        pre-scaled; every limb product is split exactly into a multiple of 2^43 (hi) and a remainder (lo) with the magic-constant trick:
//        a_k = fma(x, W, a_(k-1)) rounds at the magic constant's ulp, d = a_(k-1) - a_k is minus the rounded part, fma(x, W, d) the exact
//        remainder.  Additions and subtractions are three v_add_f64 with no carry at all.
//
// Both are verified against the portable multiplication on random and edge inputs (values compared as residues mod p on the host), then
// timed as register-resident butterflies (x0 +- w x1, the rate kernel of felab.hip) at 8 / 4 / 2 waves per SIMD.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 replab.hip -o _build/replab          (ISA: add -save-temps, see tools/felab/replab_isa.py)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../distaff_amd/csrc/fe.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
typedef unsigned __int128 u128;
static const u128 P128 = ((u128)0xFFFFFFFFFFFFFFFFull << 64) | 0xFFFFD30000000001ull;

// ------------------------------------------------------------------------------------------------------------------- (ii) FP64 limbs
struct fd { double l[3]; };              // value = l[0] + l[1] + l[2]; l[k] is an integer multiple of 2^(43 k) (the limb carries its weight)
struct fdw { double w[3][3]; };          // w[i][j] = limb j (weight 2^(43 j)) of (w * 2^(43 i) mod p), times 2^(-43 i): multiplies the weighted limb x.l[i]

#define P2(e) __builtin_ldexp(1.0, (e))
__device__ __forceinline__ fd fd_add(const fd& a, const fd& b) { return fd{{a.l[0] + b.l[0], a.l[1] + b.l[1], a.l[2] + b.l[2]}}; }
__device__ __forceinline__ fd fd_sub(const fd& a, const fd& b) { return fd{{a.l[0] - b.l[0], a.l[1] - b.l[1], a.l[2] - b.l[2]}}; }

// one column: sum_i x.l[i] * W.w[i][j] split exactly into H (multiple of 2^(43 (j + 1))) + L (|L| <= 1.5 * 2^(43 j + 43))
template <int J>
__device__ __forceinline__ void fd_column(const fd& x, const fdw& W, double& H, double& L) {
    const double M = 1.5 * P2(52 + 43 * (J + 1));
    const double a1 = __builtin_fma(x.l[0], W.w[0][J], M);
    const double d1 = M - a1;
    double lo = __builtin_fma(x.l[0], W.w[0][J], d1);
    const double a2 = __builtin_fma(x.l[1], W.w[1][J], a1);
    const double d2 = a1 - a2;
    lo += __builtin_fma(x.l[1], W.w[1][J], d2);
    const double a3 = __builtin_fma(x.l[2], W.w[2][J], a2);
    const double d3 = a2 - a3;
    lo += __builtin_fma(x.l[2], W.w[2][J], d3);
    H = a3 - M;
    L = lo;
}
// x * w mod p, lazy: limbs of the result are bounded by ~2^44 units (limb 0, 1) and ~2^43 (limb 2) in absolute value
__device__ __forceinline__ fd fd_mul_tw(const fd& x, const fdw& W) {
    double H0, L0, H1, L1, H2, L2;
    fd_column<0>(x, W, H0, L0);
    fd_column<1>(x, W, H1, L1);
    fd_column<2>(x, W, H2, L2);
    double r0 = L0, r1 = L1 + H0, r2 = L2 + H1;
    // H2 has weight 2^129: 2^129 = 2 * 2^128 = 45 * 2^41 - 2 (mod p)
    const double u = H2 * P2(-129);
    const double C = 45.0 * P2(41);
    const double Mf = 1.5 * P2(52 + 43);
    const double t = __builtin_fma(u, C, Mf);
    const double hF = t - Mf;
    const double lF = __builtin_fma(u, C, -hF);
    r0 += lF;
    r0 = __builtin_fma(u, -2.0, r0);
    r1 += hF;
    // bits of limb 2 at and above 2^128: k * 2^128 = k * (45 * 2^40 - 1)
    const double M2 = 1.5 * P2(52 + 128);
    const double c2 = (r2 + M2) - M2;
    r2 -= c2;
    r0 = __builtin_fma(c2 * P2(-128), 45.0 * P2(40) - 1.0, r0);
    // carries 0 -> 1 -> 2
    const double M0 = 1.5 * P2(52 + 43);
    const double c0 = (r0 + M0) - M0;
    r0 -= c0; r1 += c0;
    const double M1 = 1.5 * P2(52 + 86);
    const double c1 = (r1 + M1) - M1;
    r1 -= c1; r2 += c1;
    return fd{{r0, r1, r2}};
}

// bounds a lazy element again (what a tile does once per few stages): bits at and above 2^128 folded, carries 0 -> 1 -> 2
__device__ __forceinline__ fd fd_norm(const fd& x) {
    double r0 = x.l[0], r1 = x.l[1], r2 = x.l[2];
    const double M2 = 1.5 * P2(52 + 128);
    const double c2 = (r2 + M2) - M2;
    r2 -= c2;
    r0 = __builtin_fma(c2 * P2(-128), 45.0 * P2(40) - 1.0, r0);
    const double M0 = 1.5 * P2(52 + 43);
    const double c0 = (r0 + M0) - M0;
    r0 -= c0; r1 += c0;
    const double M1 = 1.5 * P2(52 + 86);
    const double c1 = (r1 + M1) - M1;
    r1 -= c1; r2 += c1;
    return fd{{r0, r1, r2}};
}

// ------------------------------------------------------------------------------------------------------------------- (i) 5 x 26-bit limbs
struct fr { int64_t l[5]; };              // value = sum l[k] * 2^(26 k), limbs lazy and signed (|l[k]| < 2^27 between operations)
struct frw { uint32_t w[5][5]; };         // w[i][j] = limb j (26 bits) of (w * 2^(26 i) mod p)
__device__ __forceinline__ fr fr_add(const fr& a, const fr& b) { fr r; for (int k = 0; k < 5; k++) r.l[k] = a.l[k] + b.l[k]; return r; }
__device__ __forceinline__ fr fr_sub(const fr& a, const fr& b) { fr r; for (int k = 0; k < 5; k++) r.l[k] = a.l[k] - b.l[k]; return r; }
__device__ __forceinline__ fr fr_mul_tw(const fr& x, const frw& W) {
    int64_t c[5];
#pragma unroll
    for (int j = 0; j < 5; j++) {
        int64_t s = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) s += (int64_t)(int32_t)x.l[i] * (int64_t)W.w[i][j];      // |x_i| < 2^28: 32 x 32 -> 64 multiply-add
        c[j] = s;
    }
    const int64_t MASK = (1ll << 26) - 1;
    // first propagation (arithmetic shifts: limbs are signed)
    c[1] += c[0] >> 26; c[0] &= MASK;
    c[2] += c[1] >> 26; c[1] &= MASK;
    c[3] += c[2] >> 26; c[2] &= MASK;
    c[4] += c[3] >> 26; c[3] &= MASK;
    // limb 4 has weight 2^104: bits from 24 on are multiples of 2^128 = 45 * 2^40 - 1 = 45 * 2^14 * 2^26 - 1 (mod p)
    const int64_t k = c[4] >> 24; c[4] &= (1ll << 24) - 1;                                    // |k| < 2^34
    c[1] += (k & MASK) * (45ll << 14);
    c[2] += (k >> 26) * (45ll << 14);
    c[0] -= k & MASK;
    c[1] -= k >> 26;
    // second, short propagation
    c[1] += c[0] >> 26; c[0] &= MASK;
    c[2] += c[1] >> 26; c[1] &= MASK;
    c[3] += c[2] >> 26; c[2] &= MASK;
    c[4] += c[3] >> 26; c[3] &= MASK;
    fr r; for (int q = 0; q < 5; q++) r.l[q] = c[q];
    return r;
}

// ------------------------------------------------------------------------------------------------------------------- host helpers
// independent of fe.h: plain 128-bit integer arithmetic (double-and-add multiplication)
static u128 addmod(u128 a, u128 b) { a %= P128; b %= P128; const u128 s = a + b; return (s < a || s >= P128) ? s - P128 : s; }
static u128 mulmod(u128 a, u128 b) { a %= P128; u128 r = 0; for (int i = 127; i >= 0; i--) { r = addmod(r, r); if ((b >> i) & 1) r = addmod(r, a); } return r; }
static u128 of_signed(double v) { const bool neg = v < 0; u128 m = (u128)(neg ? -v : v); m %= P128; return neg && m ? P128 - m : m; }     // an integer-valued double (any weight)
static u128 value_of(const fd& x) { return addmod(addmod(of_signed(x.l[0]), of_signed(x.l[1])), of_signed(x.l[2])); }
static u128 of_i64(int64_t v) { return v < 0 ? P128 - (u128)(-v) : (u128)v; }
static u128 value_of(const fr& x) { u128 r = 0; for (int k = 4; k >= 0; k--) r = addmod(mulmod(r, (u128)1 << 26), of_i64(x.l[k])); return r; }
static fd fd_of(u128 v) { const u128 m = ((u128)1 << 43) - 1; return fd{{(double)(uint64_t)(v & m), __builtin_ldexp((double)(uint64_t)((v >> 43) & m), 43), __builtin_ldexp((double)(uint64_t)(v >> 86), 86)}}; }
static fdw fdw_of(u128 w) {
    fdw t;
    for (int i = 0; i < 3; i++) { const fd f = fd_of(mulmod(w, (u128)1 << (43 * i))); for (int j = 0; j < 3; j++) t.w[i][j] = __builtin_ldexp(f.l[j], -43 * i); }
    return t;
}
static fr fr_of(u128 v) { fr r; for (int k = 0; k < 5; k++) r.l[k] = (int64_t)(uint64_t)((v >> (26 * k)) & ((1u << 26) - 1)); return r; }
static frw frw_of(u128 w) { frw t; for (int i = 0; i < 5; i++) { const fr f = fr_of(mulmod(w, (u128)1 << (26 * i))); for (int j = 0; j < 5; j++) t.w[i][j] = (uint32_t)f.l[j]; } return t; }

// ------------------------------------------------------------------------------------------------------------------- verification kernels
// each lane runs a chain of CHAIN butterflies (a, b) -> (a + w b, a - w b) in the representation and writes the two results; the host repeats
// the chain on residues mod p.  Lazy limbs are exercised exactly as a tile would: several operations between normalisations.
#define CHAIN 24
__global__ void check_fd(const fd* xa, const fd* xb, const fdw* tw, fd* oa, fd* ob, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    fd a = xa[i], b = xb[i]; const fdw W = tw[i];
    for (int k = 0; k < CHAIN; k++) {
        const fd m = fd_mul_tw(b, W); fd s = fd_add(a, m), d = fd_sub(a, m);
        if ((k & 3) == 3) { s = fd_norm(s); d = fd_norm(d); }          // sums grow by one product per step: bounded again every fourth step
        a = s; b = d;
    }
    oa[i] = a; ob[i] = b;
}
__global__ void check_fr(const fr* xa, const fr* xb, const frw* tw, fr* oa, fr* ob, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    fr a = xa[i], b = xb[i]; const frw W = tw[i];
    for (int k = 0; k < CHAIN; k++) {
        const fr m = fr_mul_tw(b, W); fr s = fr_add(a, m), d = fr_sub(a, m);
        // a tile would normalise sums once per round trip: here every other step (limbs stay below 2^28 in absolute value)
        if (k & 1) { const int64_t MASK = (1ll << 26) - 1; for (fr* q : {&s, &d}) { for (int t = 0; t < 4; t++) { q->l[t + 1] += q->l[t] >> 26; q->l[t] &= MASK; } } }
        a = s; b = d;
    }
    oa[i] = a; ob[i] = b;
}

// ------------------------------------------------------------------------------------------------------------------- rate kernels
// two butterflies per iteration on four values, the twiddle constant per lane (register resident): KIND 0 = product representation
template <int KIND>
__global__ void __launch_bounds__(256) rate_kernel(double* out, uint32_t iters, const fdw* twd, const frw* twr) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    double acc = 0;
    if (KIND == 0) {
        fe x0 = fe_make(tid * 2654435761u + 1, tid ^ 0x9E3779B9u, tid * 40503u + 7, 0x12345678u ^ tid), x1 = fe_make(tid + 3, tid * 7 + 1, ~tid, tid * 31 + 5);
        fe x2 = fe_make(tid * 97 + 11, tid + 77, tid * 3, 0x0FEDCBA9u + tid), x3 = fe_make(~tid * 5, tid * 13 + 9, tid + 100, tid * 11);
        const fe w = fe_make(0x6A09E667u + tid, 0xBB67AE85u, 0x3C6EF372u, 0x254FF53Au), q = fe_shift64(w);
        for (uint32_t i = 0; i < iters; i++) {
            fe m1 = fe_mul_tw(x1, w, q), m3 = fe_mul_tw(x3, w, q);
            fe a0, a1, a2, a3; fe_addsub(x0, m1, a0, a1); fe_addsub(x2, m3, a2, a3);
            x0 = a0; x1 = a2; x2 = a1; x3 = a3;
        }
        const fe s = fe_add(fe_add(x0, x1), fe_add(x2, x3)); acc = s.v[0] + s.v[1] + s.v[2] + s.v[3];
    }
    if (KIND == 1) {
        const fdw W = twd[tid & 255];
        fd x0 = {{(double)(tid + 1), P2(43) * (tid + 2), P2(86) * (tid & 1023)}}, x1 = {{(double)(tid * 3 + 1), P2(43) * (tid + 5), P2(86) * 7}};
        fd x2 = {{(double)(tid + 9), P2(43) * 11, P2(86) * (tid & 511)}}, x3 = {{(double)(tid * 5 + 1), P2(43) * (tid + 13), P2(86) * 3}};
        for (uint32_t i = 0; i < iters; i++) {
            const fd m1 = fd_mul_tw(x1, W), m3 = fd_mul_tw(x3, W);
            const fd a0 = fd_add(x0, m1), a1 = fd_sub(x0, m1), a2 = fd_add(x2, m3), a3 = fd_sub(x2, m3);
            x0 = a0; x1 = a2; x2 = a1; x3 = a3;
            if ((i & 3) == 3) { x0 = fd_norm(x0); x1 = fd_norm(x1); x2 = fd_norm(x2); x3 = fd_norm(x3); }     // as in a tile: sums bounded again every fourth stage (+13 operations per element and four stages)
        }
        acc = x0.l[0] + x1.l[1] + x2.l[2] + x3.l[0];
    }
    if (KIND == 2) {
        const frw W = twr[tid & 255];
        fr x0 = {{(int64_t)tid + 1, 2, 3, 4, 5}}, x1 = {{(int64_t)tid * 3 + 1, 7, 1, 2, 3}}, x2 = {{9, (int64_t)tid + 1, 2, 3, 4}}, x3 = {{5, 6, (int64_t)tid + 7, 8, 1}};
        const int64_t MASK = (1ll << 26) - 1;
        for (uint32_t i = 0; i < iters; i++) {
            const fr m1 = fr_mul_tw(x1, W), m3 = fr_mul_tw(x3, W);
            fr a0 = fr_add(x0, m1), a1 = fr_sub(x0, m1), a2 = fr_add(x2, m3), a3 = fr_sub(x2, m3);
            if ((i & 1) == 1) for (fr* q : {&a0, &a1, &a2, &a3}) for (int t = 0; t < 4; t++) { q->l[t + 1] += q->l[t] >> 26; q->l[t] &= MASK; }
            x0 = a0; x1 = a2; x2 = a1; x3 = a3;
        }
        acc = (double)(x0.l[0] + x1.l[1] + x2.l[2] + x3.l[3]);
    }
    out[tid] = acc;
}

template <class F>
static double time_ms(F launch, int reps) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

static uint64_t rs = 0x9E3779B97F4A7C15ull;
static uint64_t xs() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }
static u128 rnd() { u128 v = ((u128)xs() << 64) | xs(); const uint32_t m = (uint32_t)xs() & 31; if (m == 1) v |= (u128)0xFFFFFFFFFFFFFFFFull << 64; if (m == 2) v &= 0xFFFFFFFFull; if (m == 3) v = P128 - 1; if (m == 4) v = 0; return v % P128; }

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs\n", prop.name, prop.multiProcessorCount);
    // ---- correctness
    const size_t n = 1 << 16;
    std::vector<u128> A(n), B(n), Wv(n), EA(n), EB(n);
    for (size_t i = 0; i < n; i++) { A[i] = rnd(); B[i] = rnd(); Wv[i] = rnd(); u128 a = A[i], b = B[i]; for (int k = 0; k < CHAIN; k++) { const u128 m = mulmod(b, Wv[i]); const u128 s = addmod(a, m), d = addmod(a, P128 - m); a = s; b = d; } EA[i] = a; EB[i] = b; }
    {
        std::vector<fd> xa(n), xb(n), oa(n), ob(n); std::vector<fdw> tw(n);
        for (size_t i = 0; i < n; i++) { xa[i] = fd_of(A[i]); xb[i] = fd_of(B[i]); tw[i] = fdw_of(Wv[i]); }
        fd *da, *db, *doa, *dob; fdw* dt;
        CK(hipMalloc(&da, n * sizeof(fd))); CK(hipMalloc(&db, n * sizeof(fd))); CK(hipMalloc(&doa, n * sizeof(fd))); CK(hipMalloc(&dob, n * sizeof(fd))); CK(hipMalloc(&dt, n * sizeof(fdw)));
        CK(hipMemcpy(da, xa.data(), n * sizeof(fd), hipMemcpyHostToDevice)); CK(hipMemcpy(db, xb.data(), n * sizeof(fd), hipMemcpyHostToDevice)); CK(hipMemcpy(dt, tw.data(), n * sizeof(fdw), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(check_fd, dim3((unsigned)(n / 256)), dim3(256), 0, 0, da, db, dt, doa, dob, n);
        CK(hipMemcpy(oa.data(), doa, n * sizeof(fd), hipMemcpyDeviceToHost)); CK(hipMemcpy(ob.data(), dob, n * sizeof(fd), hipMemcpyDeviceToHost));
        size_t bad = 0; double maxl = 0;
        for (size_t i = 0; i < n; i++) { if (value_of(oa[i]) != EA[i] || value_of(ob[i]) != EB[i]) bad++; for (int k = 0; k < 3; k++) { const double v = __builtin_fabs(oa[i].l[k]) * __builtin_ldexp(1.0, -43 * k); if (v > maxl) maxl = v; } }
        printf("(ii) FP64 3 x 43-bit limbs : %zu chains of %d butterflies, %zu mismatches; largest limb after a chain 2^%.1f units\n", n, CHAIN, bad, __builtin_log2(maxl));
        if (bad) return 1;
    }
    {
        std::vector<fr> xa(n), xb(n), oa(n), ob(n); std::vector<frw> tw(n);
        for (size_t i = 0; i < n; i++) { xa[i] = fr_of(A[i]); xb[i] = fr_of(B[i]); tw[i] = frw_of(Wv[i]); }
        fr *da, *db, *doa, *dob; frw* dt;
        CK(hipMalloc(&da, n * sizeof(fr))); CK(hipMalloc(&db, n * sizeof(fr))); CK(hipMalloc(&doa, n * sizeof(fr))); CK(hipMalloc(&dob, n * sizeof(fr))); CK(hipMalloc(&dt, n * sizeof(frw)));
        CK(hipMemcpy(da, xa.data(), n * sizeof(fr), hipMemcpyHostToDevice)); CK(hipMemcpy(db, xb.data(), n * sizeof(fr), hipMemcpyHostToDevice)); CK(hipMemcpy(dt, tw.data(), n * sizeof(frw), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(check_fr, dim3((unsigned)(n / 256)), dim3(256), 0, 0, da, db, dt, doa, dob, n);
        CK(hipMemcpy(oa.data(), doa, n * sizeof(fr), hipMemcpyDeviceToHost)); CK(hipMemcpy(ob.data(), dob, n * sizeof(fr), hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < n; i++) if (value_of(oa[i]) != EA[i] || value_of(ob[i]) != EB[i]) bad++;
        printf("(i)  5 x 26-bit limbs      : %zu chains of %d butterflies, %zu mismatches\n", n, CHAIN, bad);
        if (bad) return 1;
    }
    // ---- rates
    std::vector<fdw> twd(256); std::vector<frw> twr(256);
    for (int i = 0; i < 256; i++) { const u128 w = rnd(); twd[i] = fdw_of(w); twr[i] = frw_of(w); }
    fdw* dtd; frw* dtr; double* out;
    CK(hipMalloc(&dtd, 256 * sizeof(fdw))); CK(hipMalloc(&dtr, 256 * sizeof(frw))); CK(hipMalloc(&out, (size_t)1 << 24));
    CK(hipMemcpy(dtd, twd.data(), 256 * sizeof(fdw), hipMemcpyHostToDevice)); CK(hipMemcpy(dtr, twr.data(), 256 * sizeof(frw), hipMemcpyHostToDevice));
    const int cus = prop.multiProcessorCount;
    const uint32_t iters = 2000;
    printf("register-resident butterflies (x0 +- w x1), twiddle constant per lane; butterflies per second over the whole device\n");
    for (int wps : {8, 4, 2}) {
        const int blocks = cus * wps;
        const double lanes = (double)blocks * 256;
        const double m0 = time_ms([&] { hipLaunchKernelGGL(rate_kernel<0>, dim3(blocks), dim3(256), 0, 0, out, iters, dtd, dtr); }, 3);
        const double m1 = time_ms([&] { hipLaunchKernelGGL(rate_kernel<1>, dim3(blocks), dim3(256), 0, 0, out, iters, dtd, dtr); }, 3);
        const double m2 = time_ms([&] { hipLaunchKernelGGL(rate_kernel<2>, dim3(blocks), dim3(256), 0, 0, out, iters, dtd, dtr); }, 3);
        printf("  %d waves per SIMD: 4 x 32 canonical (product) %.3e   FP64 3 x 43 %.3e (%.2fx)   5 x 26 integers %.3e (%.2fx)\n", wps,
               lanes * iters * 2 / (m0 * 1e-3), lanes * iters * 2 / (m1 * 1e-3), m0 / m1, lanes * iters * 2 / (m2 * 1e-3), m0 / m2);
    }
    return 0;
}
