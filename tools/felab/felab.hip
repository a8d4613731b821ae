// GPU laboratory for fe2.h: correctness of the candidate formulations against the portable multiplication on random and edge
// inputs, their throughput at 8 and 4 waves per SIMD, and the peak rate of v_mad_u64_u32 (the 32x32+64 multiply-add every
// formulation is built from).  Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 felab.hip -o _build/felab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../distaff_amd/csrc/fe.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

__device__ __forceinline__ uint64_t xs(uint64_t& s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
__device__ __forceinline__ fe canon_dev(fe a) { uint32_t t[8] = {a.v[0], a.v[1], a.v[2], a.v[3], 0, 0, 0, 0}; return fe_reduce8(t); }

// independent references for addition / subtraction (64-bit arithmetic, no carry primitives)
__device__ fe ref_add(const fe& a, const fe& b) {
    uint64_t c = (uint64_t)a.v[0] + b.v[0]; uint32_t s0 = (uint32_t)c;
    c = (uint64_t)a.v[1] + b.v[1] + (c >> 32); uint32_t s1 = (uint32_t)c;
    c = (uint64_t)a.v[2] + b.v[2] + (c >> 32); uint32_t s2 = (uint32_t)c;
    c = (uint64_t)a.v[3] + b.v[3] + (c >> 32); uint32_t s3 = (uint32_t)c;
    uint32_t cs = (uint32_t)(c >> 32);
    uint64_t d = (uint64_t)s0 + FE_C0; uint32_t t0 = (uint32_t)d;
    d = (uint64_t)s1 + FE_C1 + (d >> 32); uint32_t t1 = (uint32_t)d;
    d = (uint64_t)s2 + (d >> 32); uint32_t t2 = (uint32_t)d;
    d = (uint64_t)s3 + (d >> 32); uint32_t t3 = (uint32_t)d;
    return (cs | (uint32_t)(d >> 32)) ? fe_make(t0, t1, t2, t3) : fe_make(s0, s1, s2, s3);
}
__device__ fe ref_sub(const fe& a, const fe& b) { const fe nb = fe_is_zero(b) ? b : fe_make(FE_P0 - b.v[0], 0, 0, 0); (void)nb;
    int64_t c = (int64_t)(uint64_t)a.v[0] - b.v[0]; uint32_t d0 = (uint32_t)c;
    c = (int64_t)(uint64_t)a.v[1] - b.v[1] + (c >> 32); uint32_t d1 = (uint32_t)c;
    c = (int64_t)(uint64_t)a.v[2] - b.v[2] + (c >> 32); uint32_t d2 = (uint32_t)c;
    c = (int64_t)(uint64_t)a.v[3] - b.v[3] + (c >> 32); uint32_t d3 = (uint32_t)c;
    if ((c >> 32) == 0) return fe_make(d0, d1, d2, d3);
    int64_t e = (int64_t)(uint64_t)d0 - FE_C0; uint32_t e0 = (uint32_t)e;
    e = (int64_t)(uint64_t)d1 - FE_C1 + (e >> 32); uint32_t e1 = (uint32_t)e;
    e = (int64_t)(uint64_t)d2 + (e >> 32); uint32_t e2 = (uint32_t)e;
    e = (int64_t)(uint64_t)d3 + (e >> 32); uint32_t e3 = (uint32_t)e;
    return fe_make(e0, e1, e2, e3);
}

__global__ void check_kernel(unsigned long long* bad, uint32_t iters) {
    uint64_t s = 0x9E3779B97F4A7C15ull * (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x + 1);
    unsigned long long b = 0;
    for (uint32_t it = 0; it < iters; it++) {
        fe a = fe_make((uint32_t)xs(s), (uint32_t)xs(s), (uint32_t)xs(s), (uint32_t)xs(s));
        fe w = fe_make((uint32_t)xs(s), (uint32_t)xs(s), (uint32_t)xs(s), (uint32_t)xs(s));
        const uint32_t m = (uint32_t)xs(s);
        if ((m & 7) == 1) a.v[3] = a.v[2] = 0xFFFFFFFFu;
        if ((m & 7) == 2) w.v[3] = w.v[2] = 0xFFFFFFFFu;
        if ((m & 31) == 3) a = fe_make(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        if ((m & 31) == 4) a.v[0] = a.v[1] = 0;
        if ((m & 31) == 5) w.v[1] = w.v[0] = 0xFFFFFFFFu;
        const fe ac = canon_dev(a), wc = canon_dev(w);
        const fe ref = fe_mul_portable(ac, wc);
        if (!fe_eq(fe_mul(ac, wc), ref)) b++;
        if (!fe_eq(fe_mul_tw(a, wc, fe_shift64(wc)), ref)) b += 1ull << 8;
        if (!fe_eq(fe_mul_wide(a, w), ref)) b += 1ull << 16;
        if (!fe_eq(fe_add(ac, wc), ref_add(ac, wc))) b += 1ull << 24;
        if (!fe_eq(fe_sub(ac, wc), ref_sub(ac, wc))) b += 1ull << 32;
    }
    if (b) atomicAdd(bad, b);
}

// KIND 0: portable multiplication, 1: fe_mul, 2: fe_mul_tw, 3: butterfly with a table twiddle, 4: butterfly with a general multiplication, 5: add+sub, 6: add+sub written with 64-bit arithmetic
template <int KIND>
__global__ void __launch_bounds__(256) rate_kernel(fe* out, uint32_t iters) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    fe x0 = fe_make(tid * 2654435761u + 1, tid ^ 0x9E3779B9u, tid * 40503u + 7, 0x12345678u ^ tid), x1 = fe_make(tid + 3, tid * 7 + 1, ~tid, tid * 31 + 5);
    fe x2 = fe_make(tid * 97 + 11, tid + 77, tid * 3, 0x0FEDCBA9u + tid), x3 = fe_make(~tid * 5, tid * 13 + 9, tid + 100, tid * 11);
    const fe w = fe_make(0x6A09E667u + tid, 0xBB67AE85u, 0x3C6EF372u, 0x254FF53Au), q = fe_shift64(w);
    for (uint32_t i = 0; i < iters; i++) {
        if (KIND == 0) { x0 = fe_mul_portable(x0, w); x1 = fe_mul_portable(x1, w); x2 = fe_mul_portable(x2, w); x3 = fe_mul_portable(x3, w); }
        if (KIND == 1) { x0 = fe_mul(x0, w); x1 = fe_mul(x1, w); x2 = fe_mul(x2, w); x3 = fe_mul(x3, w); }
        if (KIND == 2) { x0 = fe_mul_tw(x0, w, q); x1 = fe_mul_tw(x1, w, q); x2 = fe_mul_tw(x2, w, q); x3 = fe_mul_tw(x3, w, q); }
        if (KIND == 3) {
            fe m1 = fe_mul_tw(x1, w, q), m3 = fe_mul_tw(x3, w, q);
            fe a0 = fe_add(x0, m1), a1 = fe_sub(x0, m1), a2 = fe_add(x2, m3), a3 = fe_sub(x2, m3);
            x0 = a0; x1 = a2; x2 = a1; x3 = a3;
        }
        if (KIND == 4) {
            fe m1 = fe_mul(x1, w), m3 = fe_mul(x3, w);
            fe a0 = fe_add(x0, m1), a1 = fe_sub(x0, m1), a2 = fe_add(x2, m3), a3 = fe_sub(x2, m3);
            x0 = a0; x1 = a2; x2 = a1; x3 = a3;
        }
        if (KIND == 5) { fe a0 = fe_add(x0, x1), a1 = fe_sub(x0, x1), a2 = fe_add(x2, x3), a3 = fe_sub(x2, x3); x0 = a0; x1 = a2; x2 = a1; x3 = a3; }
        if (KIND == 6) { fe a0 = ref_add(x0, x1), a1 = ref_sub(x0, x1), a2 = ref_add(x2, x3), a3 = ref_sub(x2, x3); x0 = a0; x1 = a2; x2 = a1; x3 = a3; }
    }
    out[tid] = fe_add(fe_add(x0, x1), fe_add(x2, x3));
}

// peak of v_mad_u64_u32: eight independent accumulators per lane, no carries consumed
__global__ void __launch_bounds__(256) mad_peak_kernel(uint64_t* out, uint32_t iters) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t a0 = tid, a1 = tid + 1, a2 = tid + 2, a3 = tid + 3, a4 = tid + 4, a5 = tid + 5, a6 = tid + 6, a7 = tid + 7;
    uint32_t x = tid * 2654435761u + 12345u, y = tid ^ 0xDEADBEEFu;
    for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            a0 = (uint64_t)x * y + a0; a1 = (uint64_t)x * (uint32_t)a0 + a1; a2 = (uint64_t)y * (uint32_t)a1 + a2; a3 = (uint64_t)x * (uint32_t)a2 + a3;
            a4 = (uint64_t)y * (uint32_t)a3 + a4; a5 = (uint64_t)x * (uint32_t)a4 + a5; a6 = (uint64_t)y * (uint32_t)a5 + a6; a7 = (uint64_t)x * (uint32_t)a6 + a7;
        }
    }
    out[tid] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
// peak of a plain 32-bit VALU op for comparison (v_add_u32 / v_xor chains)
__global__ void __launch_bounds__(256) alu_peak_kernel(uint32_t* out, uint32_t iters) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t a0 = tid, a1 = tid + 1, a2 = tid + 2, a3 = tid + 3, a4 = tid + 4, a5 = tid + 5, a6 = tid + 6, a7 = tid + 7;
    for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            a0 = a0 * 3 + a7; a1 ^= a0 + 1; a2 = a2 + a1 + 5; a3 ^= a2 >> 3; a4 = a4 + a3 + a0; a5 ^= a4 + a1; a6 = a6 + a5 + 9; a7 ^= a6 << 1;
        }
    }
    out[tid] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
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

template <int KIND>
static void rate(const char* name, fe* buf, double ops_per_iter, int blocks) {
    const uint32_t iters = 2000;
    double ms = time_ms([&] { hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, buf, iters); }, 3);
    double lanes = (double)blocks * 256;
    printf("  %-34s %5d blocks: %8.3f ms  %.3e ops/s\n", name, blocks, ms, lanes * iters * ops_per_iter / (ms * 1e-3));
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    unsigned long long* bad; CK(hipMalloc(&bad, 8)); CK(hipMemset(bad, 0, 8));
    hipLaunchKernelGGL(check_kernel, dim3(4096), dim3(256), 0, 0, bad, 2000u);
    unsigned long long hb = 0; CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
    printf("check: %.3e trials per formulation, mismatch word = 0x%llx (0 = all five agree with the portable arithmetic)\n", 4096.0 * 256 * 2000, hb);
    fe* buf; CK(hipMalloc(&buf, (size_t)1 << 26));
    const int cus = prop.multiProcessorCount;
    for (int wps : {8, 4, 2}) {           // waves per SIMD: blocks of 256 lanes = 4 waves = one per SIMD
        const int blocks = cus * wps * 8;  // 8 rounds of full residency
        printf("%d waves per SIMD\n", wps);
        // occupancy is set by the grid only when every block is resident at once; use a grid of exactly cus*wps blocks per round via iters
        rate<0>("mul portable (compiler carries)", buf, 4, cus * wps);
        rate<1>("mul general (fe_mul)", buf, 4, cus * wps);
        rate<2>("mul twiddle pair (fe_mul_tw)", buf, 4, cus * wps);
        rate<3>("DIT butterfly mul_tw/add/sub", buf, 2, cus * wps);
        rate<4>("DIT butterfly mul/add/sub", buf, 2, cus * wps);
        rate<5>("add+sub pair", buf, 2, cus * wps);
        rate<6>("add+sub pair (compiler carries)", buf, 2, cus * wps);
        (void)blocks;
    }
    {
        const uint32_t iters = 4000;
        for (int wps : {8, 4, 2, 1}) {
            double ms = time_ms([&] { hipLaunchKernelGGL(mad_peak_kernel, dim3(cus * wps), dim3(256), 0, 0, (uint64_t*)buf, iters); }, 3);
            printf("v_mad_u64_u32 peak, %d waves/SIMD: %.3f ms  %.3e mad/s\n", wps, ms, (double)cus * wps * 256 * iters * 32 / (ms * 1e-3));
            ms = time_ms([&] { hipLaunchKernelGGL(alu_peak_kernel, dim3(cus * wps), dim3(256), 0, 0, (uint32_t*)buf, iters); }, 3);
            printf("32-bit VALU op chain,  %d waves/SIMD: %.3f ms  (8 statements x 4 per iteration)\n", wps, ms);
        }
    }
    return hb != 0;
}
