// Box fingerprint: straight-line code of 16 KiB and of 176 KiB, every wavefront running it once -- the shape of the constraint kernels.
// bench.py reports both times with the arithmetic calibrations (dst_bench_mad, dst_bench_mulmod): on a healthy MI355X the time per
// instruction does not depend on the code size (tools/icache_probe measures the whole curve and the SQC_ICACHE_* counters); a box on
// which code beyond the 64 KiB instruction cache of a CU pair is slow -- one lease in round 2 ran a 172 KB constraint kernel at a third
// of its speed -- shows up as code_ratio >> 1.  No product kernel is that large any more (the build gate, __graft_entry__.py).
#include "ctx.h"
#if DST_TEST_HOOKS            // a laboratory instrument: not part of the product library

#define PROBE_STR_(x) #x
#define PROBE_STR(x) PROBE_STR_(x)
// N x 32 bytes: four independent v_mad_u64_u32 per repetition
#define PROBE_BODY(N) asm volatile(".rept " PROBE_STR(N) "\n v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n .endr" \
                                   : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y) : "vcc")

extern __shared__ __attribute__((aligned(16))) unsigned char ntt_smem[];      // the library's one dynamic-LDS array (kernels_ntt.hip)

// KIB of code; THREADS lanes per workgroup; CONVOY: a workgroup barrier after every 16 KiB of code keeps the wavefronts of a workgroup at one
// program counter, so that they share the instruction-cache lines (tools/icache_probe: misses of 256 KiB of code 352 -> 38 MB per launch)
template <int KIB, int THREADS = 128, bool CONVOY = false>
__global__ void __launch_bounds__(THREADS) code_probe_kernel(uint64_t* out, uint32_t x, uint32_t y) {
    unsigned char* probe_lds = ntt_smem;                      // 40 KiB per 128 lanes: two waves per SIMD, like the constraint kernels
    uint64_t a = threadIdx.x, b = blockIdx.x, c = x, d = y;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (KIB == 16) PROBE_BODY(512);
    else if constexpr (!CONVOY) PROBE_BODY(5632);
    else {
#pragma unroll
        for (int k = 0; k < KIB / 16; k++) { PROBE_BODY(512); __builtin_amdgcn_s_barrier(); }
    }
#endif
    if ((a ^ b ^ c ^ d) == 0x1234567 && probe_lds[threadIdx.x]) out[0] = a;       // keeps the chains alive; never true in practice
}

template <int KIB, int THREADS, bool CONVOY>
static int run_probe(dst_ctx* c, double* ms) {
    const size_t lanes = (size_t)8 << 20;                     // the constraint kernels' grid at 2^20 steps
    const int lds = 40 * 1024 * (THREADS / 128);
    HIP_TRY(c, hipFuncSetAttribute((const void*)code_probe_kernel<KIB, THREADS, CONVOY>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    HIP_TRY(c, hipEventCreate(&e0)); HIP_TRY(c, hipEventCreate(&e1));
    double best = 1e30;
    for (int rep = 0; rep < 3; rep++) {
        HIP_TRY(c, hipEventRecord(e0, c->stream));
        hipLaunchKernelGGL((code_probe_kernel<KIB, THREADS, CONVOY>), dim3((unsigned)(lanes / THREADS)), dim3(THREADS), lds, c->stream, (uint64_t*)c->scratch, 3u, 5u);
        HIP_TRY(c, hipEventRecord(e1, c->stream));
        HIP_TRY(c, hipEventSynchronize(e1));
        float f = 0; HIP_TRY(c, hipEventElapsedTime(&f, e0, e1));
        if (rep > 0 && f < best) best = f;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    *ms = best;
    return DST_OK;
}

// code_kib: 16, 176, or 176 + 1 = the convoy form of the 176 KiB kernel (256 lanes per workgroup, a barrier every 16 KiB)
int k_bench_code(dst_ctx* c, uint32_t code_kib, double* ms) {
    if (code_kib == 16) return run_probe<16, 128, false>(c, ms);
    if (code_kib == 176) return run_probe<176, 128, false>(c, ms);
    if (code_kib == 177) return run_probe<176, 256, true>(c, ms);
    c->err = "dst_bench_code: code size must be 16, 176 or 177 (= 176 KiB in the convoy form)";
    return DST_ERR_ARG;
}
#endif  // DST_TEST_HOOKS
