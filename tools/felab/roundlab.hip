// GPU laboratory: the in-LDS butterfly rounds of the NTT kernels in isolation (no HBM traffic), to separate what the rounds cost
// (LDS round trips, index arithmetic, barriers) from what the memory phases of the passes cost.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 roundlab.hip -o _build/roundlab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../distaff_amd/csrc/ntt_lds.h"

// ---- laboratory only (measured: +2.7 % on the rounds of 1024 x 4 tiles at 4 waves per SIMD, 70 registers -- it cannot run at the 8 waves
// per SIMD that give +8.5 % with the radix-4 rounds, so the product kernels keep radix-4) -------------------------------------------------
// Radix-8 round of the same DIF: THREE radix-2 stages (s, s + 1, s + 2) on eight elements a lane holds in registers -- for a
// 1024-point tile 3 + 3 + 2 + 2 stages = four LDS round trips and barriers instead of five (lds_ntt_dif8 below).  The group of a lane is
// i0 + j * q, j < 8, q = D / 4 (D = distance of stage s), i0 = blk * 2D + pos, pos < q.  Twiddle indices as in lds_ntt_dif: stage s
// w_2D^(pos + j q) (j < 4), stage s + 1 w_D^(pos + j q) (j < 2), stage s + 2 w_{D/2}^pos.
template <class Out>
__device__ __forceinline__ void lds_dif8_group(fe* L, const fe_tw* W, uint32_t log_len, uint32_t log_t, uint32_t s, uint32_t w, bool fin, const Out& out) {
    const uint32_t T = 1u << log_t;
    const uint32_t lq = log_len - s - 2, q = 1u << lq;            // D = 4q
    const uint32_t t = w & (T - 1), g = w >> log_t;
    const uint32_t pos = g & (q - 1), blk = g >> lq;
    const uint32_t i0 = (blk << (lq + 3)) + pos;
    fe x[8];
    static_for<0, 8>([&](auto j_) { constexpr int j = decltype(j_)::value; x[j] = L[lds_slot(i0 + j * q, t, log_t)]; });
    // stage s: (j, j + 4)
    static_for<0, 4>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        fe sum, dif;
        fe_addsub(x[j], x[j + 4], sum, dif);
        x[j] = sum;
        x[j + 4] = (j == 0 && q == 1) ? dif : fe_mul_tw(dif, W[dif_tw_slot((pos + j * q) << (s - 1))]);     // q == 1: pos == 0, w^0 = 1
    });
    // stage s + 1: (j, j + 2) inside each half
    static_for<0, 2>([&](auto h_) {
        constexpr int h = decltype(h_)::value * 4;
        static_for<0, 2>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            fe sum, dif;
            fe_addsub(x[h + j], x[h + j + 2], sum, dif);
            x[h + j] = sum;
            x[h + j + 2] = (j == 0 && q == 1) ? dif : fe_mul_tw(dif, W[dif_tw_slot((pos + j * q) << s)]);
        });
    });
    // stage s + 2: (j, j + 1)
    const bool tw2 = q != 1;
    fe_tw w2; if (tw2) w2 = W[dif_tw_slot(pos << (s + 1))];
    static_for<0, 4>([&](auto j_) {
        constexpr int j = decltype(j_)::value * 2;
        fe sum, dif;
        fe_addsub(x[j], x[j + 1], sum, dif);
        x[j] = sum;
        x[j + 1] = tw2 ? fe_mul_tw(dif, w2) : dif;
    });
    static_for<0, 8>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        if (fin) out.put(i0 + j * q, t, x[j], typename Out::Tok{}); else L[lds_slot(i0 + j * q, t, log_t)] = x[j];
    });
}
// all stages of a transform whose length is at least 2^6: radix-8 rounds while three or more stages remain beyond a radix-4 tail
template <int THREADS, class Out = NttKeepInLds>
__device__ __forceinline__ void lds_ntt_dif8(fe* L, const fe_tw* W, uint32_t log_len, uint32_t log_t, const Out& out = Out()) {
    const uint32_t T = 1u << log_t;
    uint32_t s = 1;
    // stage counts of the rounds: as many threes as leave an even remainder handled by radix-4 rounds (10 = 3 + 3 + 2 + 2, 8 = 3 + 3 + 2)
    uint32_t threes = log_len / 3;
    while (threes && ((log_len - 3 * threes) & 1u)) threes--;
    for (uint32_t r = 0; r < threes; r++, s += 3) {
        const bool fin = Out::active && s + 2 == log_len;
        for (uint32_t w = threadIdx.x; w < ((1u << log_len) >> 3) * T; w += THREADS) lds_dif8_group(L, W, log_len, log_t, s, w, fin, out);
        if (!fin) __syncthreads();
    }
    if (s <= log_len) lds_ntt_dif<THREADS, Out>(L, W, log_len, log_t, s, log_len + 1u, out);
}

// ---- laboratory only (round 4): a lane owns TWO columns of one butterfly position -- the three stage-twiddle pairs of a radix-4 round
// are read once and used twice (LDS operations per element and round 2.75 instead of 3.5), eight elements per lane, 512 lanes per
// 1024 x 4 tile (4 waves per SIMD).  Same arithmetic as lds_dif_round.
template <int THREADS>
__device__ __forceinline__ void lds_dif_round_2col(fe* L, const fe_tw* W, uint32_t log_len, uint32_t log_t, uint32_t s, uint32_t lane) {
    const uint32_t T = 1u << log_t, H = T >> 1;
    const uint32_t ld = log_len - s;
    const uint32_t d = 1u << ld, hd = d >> 1;
    const bool last = (s + 1 == log_len);
    for (uint32_t w = lane; w < ((1u << log_len) >> 2) * H; w += THREADS) {
        const uint32_t q = w >> (log_t - 1), tt = w & (H - 1);
        const uint32_t pos = q & (hd - 1), blk = q >> (ld - 1);
        const uint32_t i0 = (blk << (ld + 1)) + pos;
        fe_tw w0, w1, w2;
        if (hd != 1) w0 = W[dif_tw_slot(pos << (s - 1))];
        w1 = W[dif_tw_slot((pos + hd) << (s - 1))];
        if (!last && hd != 1) w2 = W[dif_tw_slot(pos << s)];
        static_for<0, 2>([&](auto c_) {
            constexpr int c = decltype(c_)::value;
            const uint32_t t = tt + c * H;
            fe* p0 = L + lds_slot(i0, t, log_t); fe* p1 = L + lds_slot(i0 + hd, t, log_t); fe* p2 = L + lds_slot(i0 + d, t, log_t); fe* p3 = L + lds_slot(i0 + d + hd, t, log_t);
            const fe x0 = *p0, x1 = *p1, x2 = *p2, x3 = *p3;
            fe a0, a1, a2, a3;
            fe_addsub(x0, x2, a0, a2);
            fe_addsub(x1, x3, a1, a3);
            if (hd != 1) a2 = fe_mul_tw(a2, w0);
            a3 = fe_mul_tw(a3, w1);
            fe y0, y1, y2, y3;
            fe_addsub(a0, a1, y0, y1);
            fe_addsub(a2, a3, y2, y3);
            if (!last && hd != 1) { y1 = fe_mul_tw(y1, w2); y3 = fe_mul_tw(y3, w2); }
            *p0 = y0; *p1 = y1; *p2 = y2; *p3 = y3;
        });
    }
    __syncthreads();
}
template <int THREADS>
__device__ __forceinline__ void lds_ntt_dif_2col(fe* L, const fe_tw* W, uint32_t log_len, uint32_t log_t) {
    const uint32_t lane = lds_opaque_lane();
    for (uint32_t s = 1; s + 1 <= log_len; s += 2) lds_dif_round_2col<THREADS>(L, W, log_len, log_t, s, lane);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

template <int THREADS, int KIND>      // KIND 0: DIF radix-4 rounds, 1: DIT, 2: DIF with radix-8 rounds
__global__ void __launch_bounds__(THREADS, 4) rounds_kernel(const fe_tw* tw, fe* out, uint32_t log_len, uint32_t log_t, uint32_t reps) {
    fe* L = reinterpret_cast<fe*>(smem);
    const uint32_t len = 1u << log_len, T = 1u << log_t;
    fe_tw* TW = reinterpret_cast<fe_tw*>(L + len * T);
    constexpr bool DIT = KIND == 1;
    for (uint32_t i = threadIdx.x; i < len; i += THREADS) TW[DIT ? i : dif_tw_slot(i)] = tw[i];
    for (uint32_t i = threadIdx.x; i < len * T; i += THREADS) L[i] = fe_make(i * 2654435761u + blockIdx.x, i ^ 0x9E3779B9u, i * 40503u + 7, 0x12345678u ^ i);
    __syncthreads();
    for (uint32_t r = 0; r < reps; r++) {
        if (KIND == 1) lds_ntt_dit<THREADS>(L, TW, log_len, log_t, 1u, log_len + 1u);
        else if (KIND == 2) lds_ntt_dif8<THREADS>(L, TW, log_len, log_t);
        else if (KIND == 3) lds_ntt_dif_2col<THREADS>(L, TW, log_len, log_t);
        else lds_ntt_dif<THREADS>(L, TW, log_len, log_t, 1u, log_len + 1u);
    }
    fe acc = fe_zero();
    for (uint32_t i = threadIdx.x; i < len * T; i += THREADS) acc = fe_add(acc, fe_mul_tw(L[i], tw[(i * 7u + 3u) & 1023u]));      // position-dependent checksum
    out[blockIdx.x * THREADS + threadIdx.x] = acc;
}

template <int THREADS, int DIT>
static fe run(const char* name, const fe_tw* tw, fe* out, uint32_t log_len, uint32_t log_t, size_t lds, int blocks) {
    const uint32_t reps = 20;
    CK(hipFuncSetAttribute((const void*)rounds_kernel<THREADS, DIT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    int nb = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rounds_kernel<THREADS, DIT>, THREADS, lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((rounds_kernel<THREADS, DIT>), dim3(blocks), dim3(THREADS), lds, 0, tw, out, log_len, log_t, 2u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((rounds_kernel<THREADS, DIT>), dim3(blocks), dim3(THREADS), lds, 0, tw, out, log_len, log_t, reps);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bf = (double)blocks * reps * (1u << log_len) * (1u << log_t) * log_len / 2.0;      // radix-2 butterflies
    printf("  %-44s %d WG/CU, %5d blocks: %8.3f ms  %.3e butterflies/s  (%.2f us per tile transform)\n", name, nb, blocks, ms, bf / (ms * 1e-3), ms * 1e3 / reps / (blocks / 256.0 / nb) / nb);
    fe h[64]; CK(hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost));
    fe sum = fe_zero(); for (int i = 0; i < 64; i++) sum = fe_add(sum, h[i]);
    return sum;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    fe_tw* tw; CK(hipMalloc(&tw, 4096 * sizeof(fe_tw)));
    { fe_tw* h = (fe_tw*)malloc(4096 * sizeof(fe_tw)); uint64_t s = 88172645463325252ull; for (int i = 0; i < 4096; i++) { uint32_t v[4]; for (int j = 0; j < 4; j++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v[j] = (uint32_t)s; } v[3] &= 0x7FFFFFFFu; h[i] = fe_tw_make(fe_make(v[0], v[1], v[2], v[3])); } CK(hipMemcpy(tw, h, 4096 * sizeof(fe_tw), hipMemcpyHostToDevice)); free(h); }
    fe* out; CK(hipMalloc(&out, (size_t)cus * 8 * 1024 * sizeof(fe)));
    printf("in-LDS rounds only, 1024-point x 4-column tiles (reference: 4.7e11 butterflies/s for register-resident arithmetic at 4 waves/SIMD)\n");
    fe a = run<512, 0>("DIF, 512 lanes, 80 KiB", tw, out, 10, 2, 65536 + 512 * 32, cus * 2 * 4);
    fe b = run<512, 2>("DIF radix-8 rounds, 512 lanes, 80 KiB", tw, out, 10, 2, 65536 + 512 * 32, cus * 2 * 4);
    printf("  radix-8 rounds %s the radix-4 rounds (checksum of wave 0's results)\n", fe_eq(a, b) ? "agree with" : "DIFFER from");
    run<512, 1>("DIT, 512 lanes, 96 KiB", tw, out, 10, 2, 65536 + 1024 * 32, cus * 2 * 4);
    run<1024, 1>("DIT, 1024 lanes, 96 KiB", tw, out, 10, 2, 65536 + 1024 * 32, cus * 4);
    run<1024, 0>("DIF, 1024 lanes, 80 KiB", tw, out, 10, 2, 65536 + 512 * 32, cus * 4);
    run<1024, 2>("DIF radix-8 rounds, 1024 lanes, 80 KiB", tw, out, 10, 2, 65536 + 512 * 32, cus * 4);
    fe c2 = run<512, 3>("DIF, two columns per lane, 512 lanes, 80 KiB", tw, out, 10, 2, 65536 + 512 * 32, cus * 2 * 4);
    printf("  two-column rounds %s the radix-4 rounds\n", fe_eq(a, c2) ? "agree with" : "DIFFER from");
    a = run<512, 0>("DIF, 512 lanes, 256-point x 16-column", tw, out, 8, 4, 65536 + 256 * 32, cus * 2 * 4);
    b = run<512, 2>("DIF radix-8 rounds, 256-point x 16-column", tw, out, 8, 4, 65536 + 256 * 32, cus * 2 * 4);
    printf("  radix-8 rounds %s the radix-4 rounds\n", fe_eq(a, b) ? "agree with" : "DIFFER from");
    return 0;
}
