// Instruction-cache probe for gfx950: straight-line kernels of 16 ... 256 KiB of code, every wavefront runs its kernel's code exactly
// once -- the shape of the constraint kernels (air_kernel.h), whose largest instance was 172 KB of code and ran three times slower on one
// box of the pool.  A CU pair shares a 64 KiB instruction cache; the question is what a kernel larger than that
// costs, on this box, and whether wavefronts that run the code together (one 512-lane workgroup = all 8 waves of a CU at two waves per
// SIMD, optionally re-aligned by s_barrier) share the fetches.
//
// The body is N repetitions of four independent v_mad_u64_u32 (8 bytes each, the instruction the field arithmetic is made of), written
// with .rept inside ONE asm statement: code size = 32 * N bytes, exactly.  Occupancy is pinned at two waves per SIMD with an LDS
// allocation (40 KiB per 128-lane workgroup, 4 per CU; 160 KiB / 1 for 512 lanes).  The grid is the constraint kernels' at 2^20 steps:
// 8 * 2^20 lanes.
//
//   build:  hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/icache_probe/icache_probe.hip -o tools/icache_probe/_build/icache_probe
//   run:    tools/icache_probe/_build/icache_probe [--json]          (rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES ... around it)
// Prints per kernel: code KiB, lanes per workgroup, barrier period, ms, ns per 1000 instructions per wave slot, ratio to the 16 KiB kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

#define STR_(x) #x
#define STR(x) STR_(x)
// N * 32 bytes of code
#define BODY(N) asm volatile(".rept " STR(N) "\n v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n .endr" \
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y) : "vcc")

// KIB of code; THREADS lanes per workgroup; BAR > 0: an s_barrier after every BAR KiB of code
template <int KIB, int THREADS, int BAR>
__global__ void __launch_bounds__(THREADS) probe(uint64_t* out, uint32_t x, uint32_t y) {
    extern __shared__ char lds[];
    uint64_t a = threadIdx.x, b = blockIdx.x, c = x, d = y;
    constexpr int CHUNK = BAR > 0 ? BAR : KIB;
#pragma unroll
    for (int k = 0; k < KIB / CHUNK; k++) {
        if constexpr (CHUNK == 4) BODY(128);
        else if constexpr (CHUNK == 8) BODY(256);
        else if constexpr (CHUNK == 16) BODY(512);
        else if constexpr (CHUNK == 32) BODY(1024);
        else if constexpr (CHUNK == 48) BODY(1536);
        else if constexpr (CHUNK == 64) BODY(2048);
        else if constexpr (CHUNK == 96) BODY(3072);
        else if constexpr (CHUNK == 128) BODY(4096);
        else if constexpr (CHUNK == 176) BODY(5632);
        else if constexpr (CHUNK == 256) BODY(8192);
        if constexpr (BAR > 0) __builtin_amdgcn_s_barrier();
    }
    if ((a ^ b ^ c ^ d) == 0x1234567 && lds[threadIdx.x]) out[0] = a;        // keeps the chains alive; never true in practice
}

struct Row { int kib, threads, bar; double ms; };
static std::vector<Row> rows;

template <int KIB, int THREADS, int BAR>
static void run(uint64_t* out, hipEvent_t e0, hipEvent_t e1) {
    const size_t lanes = (size_t)8 << 20;
    const unsigned grid = (unsigned)(lanes / THREADS);
    const size_t lds = THREADS == 128 ? 40 * 1024 : THREADS == 256 ? 80 * 1024 : 160 * 1024;     // two waves per SIMD
    CK(hipFuncSetAttribute((const void*)probe<KIB, THREADS, BAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    double best = 1e30;
    for (int rep = 0; rep < 4; rep++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((probe<KIB, THREADS, BAR>), dim3(grid), dim3(THREADS), lds, 0, out, 3u, 5u);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0) best = std::min(best, (double)ms);
    }
    rows.push_back({KIB, THREADS, BAR, best});
}

int main(int argc, char** argv) {
    const bool json = argc > 1 && !strcmp(argv[1], "--json");
    uint64_t* out; CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    run<16, 128, 0>(out, e0, e1);  run<32, 128, 0>(out, e0, e1);  run<48, 128, 0>(out, e0, e1);  run<64, 128, 0>(out, e0, e1);
    run<96, 128, 0>(out, e0, e1);  run<128, 128, 0>(out, e0, e1); run<176, 128, 0>(out, e0, e1); run<256, 128, 0>(out, e0, e1);
    run<64, 512, 0>(out, e0, e1);  run<128, 512, 0>(out, e0, e1); run<176, 512, 0>(out, e0, e1); run<256, 512, 0>(out, e0, e1);
    run<128, 512, 16>(out, e0, e1); run<176, 512, 16>(out, e0, e1); run<256, 512, 16>(out, e0, e1);
    run<128, 256, 0>(out, e0, e1); run<256, 256, 0>(out, e0, e1);
    run<128, 128, 16>(out, e0, e1); run<256, 128, 16>(out, e0, e1);
    // time per 1000 instructions of one wave slot: 2^23 lanes / 64 = 2^17 waves, 8 waves per CU, 256 CUs
    const double waves = (double)((size_t)8 << 20) / 64.0, slots = 256.0 * 8.0;
    double base = 0;
    if (json) printf("{\"icache_probe\": [");
    for (size_t i = 0; i < rows.size(); i++) {
        const Row& r = rows[i];
        const double inst = r.kib * 1024.0 / 8.0;
        const double ns_per_k = r.ms * 1e6 / (waves / slots) / inst * 1000.0;
        if (i == 0) base = ns_per_k;
        if (json) printf("%s{\"code_kib\": %d, \"lanes\": %d, \"barrier_kib\": %d, \"ms\": %.4f, \"ns_per_kinst\": %.1f, \"rel\": %.3f}", i ? ", " : "", r.kib, r.threads, r.bar, r.ms, ns_per_k, ns_per_k / base);
        else printf("code %3d KiB  %3d lanes/workgroup  barrier every %2d KiB   %8.4f ms   %7.1f ns per 1000 instructions   x%.3f\n", r.kib, r.threads, r.bar, r.ms, ns_per_k, ns_per_k / base);
    }
    if (json) printf("]}\n");
    return 0;
}
