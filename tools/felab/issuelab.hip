// Two calibrations behind bench.py's roofline section, standalone:
//
//   issuelab issue            how many shader cycles one wave64 instruction of each kind the field arithmetic is made of occupies a SIMD's
//                             VALU issue port: v_add_u32, v_add_co / v_addc_co with SGPR-pair carries, v_cndmask on an SGPR mask, v_xor,
//                             v_alignbit, v_mad_u64_u32, v_mul_lo_u32 -- and v_fma_f32, for which the guide quotes 2 cycles.  Every wave times
//                             itself with s_memtime (tick = shader cycle); 1, 4 and 8 waves per SIMD.  bench.py prices a wave64 VALU
//                             instruction at 4 cycles (VALU_ISSUE_PEAK); this is the measurement under that number.
//   issuelab fetch <pattern>  reads a 1 GiB buffer exactly once in one of the access patterns below; run under
//                             `rocprofv3 --pmc FETCH_SIZE` the counter's value against the known byte count gives the factor for THAT pattern
//                             (the guide's x2 is calibrated for wide streaming reads only):
//                               wide    every lane 16 B, a wave 1 KiB contiguous (the control: FETCH_SIZE should be 1/2 of the bytes)
//                               seg16k  64-byte row segments (4 lanes x 16 B), rows 16 KiB apart, 1024 rows per tile, a workgroup walks
//                                       through 4 adjacent tiles: the first pass of a transform at n = 2^20 (kernels_ntt.hip ntt_pass_a)
//                               seg32k  the same with rows 32 KiB apart and the rows m and m + 1024 of a tile (2048 rows): n = 2^22
//
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 issuelab.hip -o _build/issuelab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef uint64_t cf_t;      // a carry / select mask: one bit per lane in an SGPR pair

enum { K_ADD = 0, K_ADDCO_ADDC = 1, K_CNDMASK = 2, K_XOR = 3, K_ALIGNBIT = 4, K_MAD64 = 5, K_MULLO = 6, K_FMA = 7, K_SUBCO_SUBB = 8,
       K_ADDCO_VCC = 9, K_CNDMASK_VCC = 10, K_PERM = 11, K_ADD3 = 12, K_XOR3 = 13, K_LSHL_ADD = 14, K_LSHLREV = 15, K_MAD_U32_U24 = 16, K_MUL_HI = 17, K_BFE = 18, K_MAD64_VCC = 19,
       K_ADDCO_ONLY = 20, K_ADDC_ONLY = 21, K_COUNT = 22 };
static const char* kind_name[K_COUNT] = {"v_add_u32", "v_add_co_u32 + v_addc_co_u32 (SGPR-pair carries, e64)", "v_cndmask_b32 (SGPR-pair mask, e64)", "v_xor_b32",
                                         "v_alignbit_b32", "v_mad_u64_u32 (carry out to an SGPR pair)", "v_mul_lo_u32", "v_fma_f32", "v_sub_co_u32 + v_subb_co_u32 (SGPR-pair borrows, e64)",
                                         "v_add_co_u32 + v_addc_co_u32 (carries in VCC, e32)", "v_cndmask_b32 (mask in VCC, e32)", "v_perm_b32", "v_add3_u32", "v_xor3_b32", "v_lshl_add_u32",
                                         "v_lshlrev_b32", "v_mad_u32_u24", "v_mul_hi_u32", "v_bfe_u32", "v_mad_u64_u32 (carry out to VCC)",
                                         "v_add_co_u32 alone (SGPR-pair carry out, e64)", "v_addc_co_u32 alone (SGPR-pair carry in and out, e64)"};

// 16 independent accumulators, 2 instructions each per iteration: 32 instructions per iteration, every dependency 16 instructions away
// Residency is PINNED by the launch: the dynamic LDS request lets exactly 1 (>= 81 KiB) or 2 (<= 80 KiB) workgroups share a CU, and every
// wavefront of a workgroup is resident at once -- 256 lanes + 100 KiB = 1 wave per SIMD, 1024 lanes + 100 KiB = 4, 2 x (1024 lanes + 70 KiB) = 8.
extern __shared__ unsigned char issue_smem[];
template <int K>
__global__ void __launch_bounds__(1024) issue_kernel(uint32_t* out, unsigned long long* cycles, uint32_t iters, uint32_t seed) {
    constexpr bool USE_W = (K == K_MAD64 || K == K_MAD64_VCC), USE_F = (K == K_FMA);
    uint32_t x[16], y = seed | 1u, z = threadIdx.x * 2654435761u + 12345u;
    uint64_t w[USE_W ? 16 : 1];
    float f[USE_F ? 16 : 1], g = 1.0000001f, h = 1e-9f;
    cf_t c[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { x[i] = z + i * 977u; c[i] = 0; if constexpr (USE_W) w[i] = ((uint64_t)x[i] << 32) | (z ^ i); if constexpr (USE_F) f[i] = 1.0f + i; }
    cf_t mask;
    asm volatile("v_cmp_gt_u32_e64 %0, %1, %2" : "=s"(mask) : "v"(z), "v"(y));
    if (seed == 0xFFFFFFFFu) issue_smem[threadIdx.x] = 1;          // (keeps the allocation; never true)
    const unsigned long long r0 = wall_clock64();                     // s_memrealtime: constant 100 MHz
    const unsigned long long t0 = __builtin_readcyclecounter();       // s_memtime
    for (uint32_t it = 0; it < iters; it++) {
        // first the 16 "first links", then the 16 "second links": a carry is consumed 16 instructions after it was produced, as in the
        // interleaved chains of fe.h (back to back the compiler has to pad the VALU-writes-SGPR -> VALU-reads-SGPR wait state with an s_nop)
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if constexpr (K == K_ADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
            if constexpr (K == K_ADDCO_ADDC) asm volatile("v_add_co_u32_e64 %0, %1, %0, %2" : "+v"(x[i]), "=s"(c[i]) : "v"(y));
            if constexpr (K == K_SUBCO_SUBB) asm volatile("v_sub_co_u32_e64 %0, %1, %0, %2" : "+v"(x[i]), "=s"(c[i]) : "v"(y));
            if constexpr (K == K_CNDMASK) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "s"(mask));
            if constexpr (K == K_XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
            if constexpr (K == K_ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(x[i]) : "v"(y));
            if constexpr (K == K_MAD64) asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(w[i]), "=s"(c[i]) : "v"(y), "v"(z));
            if constexpr (K == K_MULLO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
            if constexpr (K == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(g), "v"(h));
            if constexpr (K == K_ADDCO_VCC) asm volatile("v_add_co_u32_e32 %0, vcc, %0, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %2, vcc" : "+v"(x[i]) : "v"(y), "v"(z) : "vcc");
            if constexpr (K == K_CNDMASK_VCC) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(y) : );
            if constexpr (K == K_PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));
            if constexpr (K == K_ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));
            if constexpr (K == K_LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x[i]) : "v"(y));
            if constexpr (K == K_LSHLREV) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(x[i]));
            if constexpr (K == K_MAD_U32_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));
            if constexpr (K == K_MUL_HI) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
            if constexpr (K == K_BFE) asm volatile("v_bfe_u32 %0, %0, 3, 17" : "+v"(x[i]));
            if constexpr (K == K_MAD64_VCC) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(y), "v"(z) : "vcc");
            if constexpr (K == K_ADDCO_ONLY) asm volatile("v_add_co_u32_e64 %0, %1, %0, %2" : "+v"(x[i]), "=s"(c[i]) : "v"(y));
            if constexpr (K == K_ADDC_ONLY) asm volatile("v_addc_co_u32_e64 %0, %1, %0, %2, %1" : "+v"(x[i]), "+s"(c[i]) : "v"(y));
        }
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if constexpr (K == K_ADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(z));
            if constexpr (K == K_ADDCO_ADDC) asm volatile("v_addc_co_u32_e64 %0, %1, %0, %2, %1" : "+v"(x[i]), "+s"(c[i]) : "v"(z));
            if constexpr (K == K_SUBCO_SUBB) asm volatile("v_subb_co_u32_e64 %0, %1, %0, %2, %1" : "+v"(x[i]), "+s"(c[i]) : "v"(z));
            if constexpr (K == K_CNDMASK) asm volatile("v_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(x[i]) : "v"(z), "s"(mask));
            if constexpr (K == K_XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(z));
            if constexpr (K == K_ALIGNBIT) asm volatile("v_alignbit_b32 %0, %1, %0, 13" : "+v"(x[i]) : "v"(z));
            if constexpr (K == K_MAD64) asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(w[i]), "=s"(c[i]) : "v"(z), "v"(y));
            if constexpr (K == K_MULLO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(z));
            if constexpr (K == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(g), "v"(h));
            if constexpr (K == K_ADDCO_VCC) asm volatile("v_add_co_u32_e32 %0, vcc, %0, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %2, vcc" : "+v"(x[i]) : "v"(z), "v"(y) : "vcc");
            if constexpr (K == K_CNDMASK_VCC) asm volatile("v_cndmask_b32_e32 %0, %1, %0, vcc" : "+v"(x[i]) : "v"(z) : );
            if constexpr (K == K_PERM) asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(x[i]) : "v"(z), "v"(y));
            if constexpr (K == K_ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(z), "v"(y));
            if constexpr (K == K_LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 5, %1" : "+v"(x[i]) : "v"(z));
            if constexpr (K == K_LSHLREV) asm volatile("v_lshlrev_b32 %0, 5, %0" : "+v"(x[i]));
            if constexpr (K == K_MAD_U32_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x[i]) : "v"(z), "v"(y));
            if constexpr (K == K_MUL_HI) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[i]) : "v"(z));
            if constexpr (K == K_BFE) asm volatile("v_bfe_u32 %0, %0, 1, 27" : "+v"(x[i]));
            if constexpr (K == K_MAD64_VCC) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(z), "v"(y) : "vcc");
            if constexpr (K == K_ADDCO_ONLY) asm volatile("v_add_co_u32_e64 %0, %1, %0, %2" : "+v"(x[i]), "=s"(c[i]) : "v"(z));
            if constexpr (K == K_ADDC_ONLY) asm volatile("v_addc_co_u32_e64 %0, %1, %0, %2, %1" : "+v"(x[i]), "+s"(c[i]) : "v"(z));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = wall_clock64();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) { acc ^= x[i] ^ (uint32_t)c[i]; if constexpr (USE_W) acc ^= (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32); if constexpr (USE_F) acc ^= __float_as_uint(f[i]); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) { const size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64; cycles[2 * w] = t1 - t0; cycles[2 * w + 1] = r1 - r0; }
}

template <int K>
static void run_issue(uint32_t* out, unsigned long long* cyc, int cus, FILE* f) {
    const uint32_t iters = 20000;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(issue_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    for (int wps : {1, 4, 8}) {
        const int threads = wps == 1 ? 256 : 1024, blocks = wps == 8 ? 2 * cus : cus;
        const size_t lds = wps == 8 ? 70 * 1024 : 100 * 1024;
        hipLaunchKernelGGL(issue_kernel<K>, dim3(blocks), dim3(threads), lds, 0, out, cyc, 200u, 3u);      // warm-up (clocks, code)
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(issue_kernel<K>, dim3(blocks), dim3(threads), lds, 0, out, cyc, iters, 3u);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        const size_t waves = (size_t)blocks * threads / 64;
        std::vector<unsigned long long> h(2 * waves);
        CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> tick(waves), real(waves);
        for (size_t w = 0; w < waves; w++) { tick[w] = (double)h[2 * w]; real[w] = (double)h[2 * w + 1]; }
        std::sort(tick.begin(), tick.end()); std::sort(real.begin(), real.end());
        const double med = tick[waves / 2], medr = real[waves / 2], instr = (double)iters * (K == K_ADDCO_VCC ? 64.0 : 32.0);
        const double wave_ms = medr / 100e6 * 1e3;                        // duration of the median wave by the constant 100 MHz counter
        // a SIMD holds `wps` resident waves that share its VALU port: the port spends (wave duration) / (wps * instructions) per wave-instruction
        const double mhz = med / (wave_ms * 1e-3) / 1e6;                  // what s_memtime counted per second of s_memrealtime: the shader clock during this kernel
        const double per_simd = (double)waves * instr / 1024.0 / (ms * 1e-3);      // wave-instructions per second and SIMD over the whole kernel (event time; 1024 SIMDs)
        fprintf(f, "%-58s %d waves/SIMD | kernel %.3f ms: %.3e wave-instr/s/SIMD = %5.2f shader cycles per wave-instruction at the measured %4.0f MHz | median wave %.3f ms (min %.3f max %.3f)\n",
                kind_name[K], wps, ms, per_simd, mhz * 1e6 / per_simd, mhz, wave_ms, real[0] / 100e6 * 1e3, real[waves - 1] / 100e6 * 1e3);
    }
}

// ---- FETCH_SIZE calibration ------------------------------------------------------------------------------------------------------------
struct u32x4 { uint32_t a, b, c, d; };
__global__ void __launch_bounds__(1024) wide_kernel(const u32x4* __restrict__ src, uint32_t* out, size_t elems) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < elems; i += (size_t)gridDim.x * blockDim.x) { const u32x4 v = src[i]; acc ^= v.a ^ v.b ^ v.c ^ v.d; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// columns x [rows][row_elems] arrays; a tile = 4 adjacent elements (64 B) of every row; a workgroup reads `tpb` adjacent tiles one after the other.
// lane index -> (row, element) as ntt_pass_a's fetch: idx = lane + e * 1024, row = idx >> 2, element = idx & 3.  `pre`: rows m and m + 1024.
__global__ void __launch_bounds__(1024) seg_kernel(const u32x4* __restrict__ src, uint32_t* out, uint32_t log_row_elems, uint32_t tpb, uint32_t groups, uint32_t pre) {
    const uint32_t col = blockIdx.x / groups, group = blockIdx.x % groups;
    const uint32_t rows = 1024u << pre;
    const u32x4* base = src + ((size_t)col * rows << log_row_elems);
    uint32_t acc = 0;
    for (uint32_t it = 0; it < tpb; it++) {
        const u32x4* s = base + (size_t)(group * tpb + it) * 4u;
#pragma unroll
        for (uint32_t e = 0; e < 4; e++) {
            const uint32_t idx = threadIdx.x + e * 1024u, row = idx >> 2, t = idx & 3u;
            const u32x4 v = s[((size_t)row << log_row_elems) + t];
            acc ^= v.a ^ v.b ^ v.c ^ v.d;
            if (pre) { const u32x4 u = s[((size_t)(row + 1024u) << log_row_elems) + t]; acc ^= u.a ^ u.b ^ u.c ^ u.d; }
        }
        __syncthreads();                                      // the tiles of a workgroup follow one another as in the pass (LDS hand-over there)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    if (argc >= 2 && !strcmp(argv[1], "issue")) {
        printf("device %s, %d CUs, clockRate %d kHz.  Residency pinned by LDS; wave durations by s_memrealtime (constant 100 MHz); the s_memtime rate is printed, not assumed\n", prop.name, cus, prop.clockRate);
        uint32_t* out; unsigned long long* cyc;
        CK(hipMalloc(&out, (size_t)cus * 2 * 1024 * 4)); CK(hipMalloc(&cyc, (size_t)cus * 2 * 16 * 16));
        run_issue<K_ADD>(out, cyc, cus, stdout);
        run_issue<K_ADDCO_ADDC>(out, cyc, cus, stdout);
        run_issue<K_SUBCO_SUBB>(out, cyc, cus, stdout);
        run_issue<K_CNDMASK>(out, cyc, cus, stdout);
        run_issue<K_XOR>(out, cyc, cus, stdout);
        run_issue<K_ALIGNBIT>(out, cyc, cus, stdout);
        run_issue<K_MAD64>(out, cyc, cus, stdout);
        run_issue<K_MULLO>(out, cyc, cus, stdout);
        run_issue<K_FMA>(out, cyc, cus, stdout);
        run_issue<K_ADDCO_VCC>(out, cyc, cus, stdout);
        run_issue<K_ADDCO_ONLY>(out, cyc, cus, stdout);
        run_issue<K_ADDC_ONLY>(out, cyc, cus, stdout);
        run_issue<K_CNDMASK_VCC>(out, cyc, cus, stdout);
        run_issue<K_MAD64_VCC>(out, cyc, cus, stdout);
        run_issue<K_PERM>(out, cyc, cus, stdout);
        run_issue<K_ADD3>(out, cyc, cus, stdout);
        run_issue<K_LSHL_ADD>(out, cyc, cus, stdout);
        run_issue<K_LSHLREV>(out, cyc, cus, stdout);
        run_issue<K_BFE>(out, cyc, cus, stdout);
        run_issue<K_MAD_U32_U24>(out, cyc, cus, stdout);
        run_issue<K_MUL_HI>(out, cyc, cus, stdout);
        return 0;
    }
    if (argc >= 3 && !strcmp(argv[1], "fetch")) {
        const size_t bytes = (size_t)1 << 30, elems = bytes / 16;
        u32x4* buf; uint32_t* out;
        CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 1, bytes)); CK(hipMalloc(&out, (size_t)1 << 24));
        CK(hipDeviceSynchronize());
        const int reps = 3;
        for (int r = 0; r < reps; r++) {
            if (!strcmp(argv[2], "wide")) hipLaunchKernelGGL(wide_kernel, dim3(cus * 8), dim3(1024), 0, 0, buf, out, elems);
            else if (!strcmp(argv[2], "seg16k") || !strcmp(argv[2], "seg32k")) {
                const uint32_t pre = !strcmp(argv[2], "seg32k") ? 1u : 0u, log_row = 10u + pre, rows = 1024u << pre, tpb = 4;
                const uint32_t tiles = (1u << log_row) / 4u, groups = tiles / tpb;
                const uint32_t cols = (uint32_t)(elems / ((size_t)rows << log_row));                  // 64 arrays of 16 MiB / 16 of 64 MiB: 1 GiB either way
                hipLaunchKernelGGL(seg_kernel, dim3(cols * groups), dim3(1024), 0, 0, buf, out, log_row, tpb, groups, pre);
            } else { printf("unknown pattern %s\n", argv[2]); return 2; }
            CK(hipDeviceSynchronize());
        }
        printf("pattern %s: %d launches, each reads %zu bytes exactly once\n", argv[2], reps, bytes);
        return 0;
    }
    printf("usage: issuelab issue | issuelab fetch wide|seg16k|seg32k\n");
    return 2;
}
