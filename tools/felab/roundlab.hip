// GPU laboratory: the in-LDS butterfly rounds of the NTT kernels in isolation (no HBM traffic), to separate what the rounds cost
// (LDS round trips, index arithmetic, barriers) from what the memory phases of the passes cost.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 roundlab.hip -o _build/roundlab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../distaff_amd/csrc/ntt_lds.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

template <int THREADS, bool DIT>
__global__ void __launch_bounds__(THREADS, 4) rounds_kernel(const fe_tw* tw, fe* out, uint32_t log_len, uint32_t log_t, uint32_t reps) {
    fe* L = reinterpret_cast<fe*>(smem);
    const uint32_t len = 1u << log_len, T = 1u << log_t;
    fe_tw* TW = reinterpret_cast<fe_tw*>(L + len * T);
    for (uint32_t i = threadIdx.x; i < len; i += THREADS) TW[DIT ? i : dif_tw_slot(i)] = tw[i];
    for (uint32_t i = threadIdx.x; i < len * T; i += THREADS) L[i] = fe_make(i * 2654435761u + blockIdx.x, i ^ 0x9E3779B9u, i * 40503u + 7, 0x12345678u ^ i);
    __syncthreads();
    for (uint32_t r = 0; r < reps; r++) {
        if (DIT) lds_ntt_dit<THREADS>(L, TW, log_len, log_t, 1u, log_len + 1u); else lds_ntt_dif<THREADS>(L, TW, log_len, log_t, 1u, log_len + 1u);
    }
    fe acc = fe_zero();
    for (uint32_t i = threadIdx.x; i < len * T; i += THREADS) acc = fe_add(acc, L[i]);
    out[blockIdx.x * THREADS + threadIdx.x] = acc;
}

template <int THREADS, bool DIT>
static void run(const char* name, const fe_tw* tw, fe* out, uint32_t log_len, uint32_t log_t, size_t lds, int blocks) {
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
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    fe_tw* tw; CK(hipMalloc(&tw, 4096 * sizeof(fe_tw)));
    { fe_tw* h = (fe_tw*)malloc(4096 * sizeof(fe_tw)); uint64_t s = 88172645463325252ull; for (int i = 0; i < 4096; i++) { uint32_t v[4]; for (int j = 0; j < 4; j++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v[j] = (uint32_t)s; } v[3] &= 0x7FFFFFFFu; h[i] = fe_tw_make(fe_make(v[0], v[1], v[2], v[3])); } CK(hipMemcpy(tw, h, 4096 * sizeof(fe_tw), hipMemcpyHostToDevice)); free(h); }
    fe* out; CK(hipMalloc(&out, (size_t)cus * 8 * 1024 * sizeof(fe)));
    printf("in-LDS rounds only, 1024-point x 4-column tiles (reference: 4.7e11 butterflies/s for register-resident arithmetic at 4 waves/SIMD)\n");
    run<512, false>("DIF, 512 lanes, 80 KiB", tw, out, 10, 2, 65536 + 512 * 32, cus * 2 * 4);
    run<512, true>("DIT, 512 lanes, 96 KiB", tw, out, 10, 2, 65536 + 1024 * 32, cus * 2 * 4);
    run<1024, true>("DIT, 1024 lanes, 96 KiB", tw, out, 10, 2, 65536 + 1024 * 32, cus * 4);
    run<1024, false>("DIF, 1024 lanes, 80 KiB", tw, out, 10, 2, 65536 + 512 * 32, cus * 4);
    run<512, false>("DIF, 512 lanes, 256-point x 16-column", tw, out, 8, 4, 65536 + 256 * 32, cus * 2 * 4);
    return 0;
}
