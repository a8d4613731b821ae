// TEST INFRASTRUCTURE -- not a product path, not a fallback.
//
// A stand-in for <hip/hip_runtime.h> that lets g++ compile distaff_amd/csrc/*.hip into tests/emu/_build/libdistaff_emu.so, a
// build of the SAME sources in which every kernel launch is executed on the host: one fiber per work-item, workgroups one after
// another (several OS threads take workgroups in parallel), __syncthreads() = every fiber of the workgroup yields.  It exists
// so that the index arithmetic of the kernels (tiles, twiddle tables, tree levels, sharded layouts, the opening plan) can be
// checked against the oracle where there is no GPU, i.e. in `pytest -m "not gpu"`.  What it does NOT cover: the gfx950 field
// arithmetic (fe.h's __HIP_DEVICE_COMPILE__ branch; the host branch is compiled here), occupancy, timing -- those are the GPU
// tests' and the bench's business.  distaff_amd never loads this library by itself; only tests/test_emulated_kernels.py does,
// in a subprocess, through DISTAFF_HIP_LIB.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <type_traits>
#include <utility>

#define DISTAFF_EMULATED 1

// ---- qualifiers ---------------------------------------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __shared__ thread_local          /* one OS thread runs whole workgroups: thread-local storage IS workgroup-shared storage */
#define __constant__ static

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3 { unsigned x, y, z; };
struct uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

extern thread_local uint3 threadIdx, blockIdx;
extern thread_local dim3 blockDim, gridDim;

// ---- device intrinsics --------------------------------------------------------------------------------------------------------
void emu_syncthreads();
static inline void __syncthreads() { emu_syncthreads(); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}
static inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }
static inline int __clzll(unsigned long long x) { return x ? __builtin_clzll(x) : 64; }
static inline int __ffs(unsigned x) { return __builtin_ffs((int)x); }
static inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline unsigned __brev(unsigned x) {
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    return __builtin_bswap32(x);
}
static inline unsigned long long __brevll(unsigned long long x) { return ((unsigned long long)__brev((unsigned)x) << 32) | __brev((unsigned)(x >> 32)); }
static inline unsigned __funnelshift_r(unsigned lo, unsigned hi, unsigned s) { s &= 31; return s ? (lo >> s) | (hi << (32 - s)) : lo; }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
template <class T> static inline T __ldg(const T* p) { return *p; }
unsigned long long emu_atomic_min_u64(unsigned long long* p, unsigned long long v);
unsigned emu_atomic_add_u32(unsigned* p, unsigned v);
static inline unsigned long long atomicMin(unsigned long long* p, unsigned long long v) { return emu_atomic_min_u64(p, v); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return emu_atomic_add_u32(p, v); }

// ---- runtime API (the subset the library uses) --------------------------------------------------------------------------------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNoDevice = 100, hipErrorNotReady = 600, hipErrorPeerAccessAlreadyEnabled = 704 };
typedef struct emu_stream* hipStream_t;
typedef struct emu_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipDeviceProp_t { char name[256]; size_t totalGlobalMem; int multiProcessorCount; char gcnArchName[256]; size_t sharedMemPerBlock; int clockRate; };

const char* hipGetErrorString(hipError_t e);
hipError_t hipGetLastError();
hipError_t hipGetDeviceCount(int* n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int* d);
inline hipError_t hipDeviceCanAccessPeer(int* can, int, int) { *can = 0; return hipSuccess; }      // one emulated device
inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipErrorInvalidValue; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int d);
hipError_t hipDeviceSynchronize();
hipError_t hipMalloc(void** p, size_t bytes);
template <class T> static inline hipError_t hipMalloc(T** p, size_t bytes) { return hipMalloc((void**)p, bytes); }
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned flags = 0);
template <class T> static inline hipError_t hipHostMalloc(T** p, size_t bytes, unsigned flags = 0) { return hipHostMalloc((void**)p, bytes, flags); }
hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b);
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s = nullptr);
hipError_t hipMemcpy2DAsync(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind, hipStream_t s = nullptr);
hipError_t hipMemset(void* dst, int value, size_t bytes);
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t s = nullptr);
hipError_t hipStreamCreate(hipStream_t* s);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
static inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }      /* every emulated operation is synchronous: a stream is always idle */
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
template <class F> static inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }
template <class F> static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, F, int, size_t) { *n = 1; return hipSuccess; }
#define hipStreamNonBlocking 1u
#define hipHostMallocDefault 0u
#define hipEventDisableTiming 2u
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }     /* every emulated operation is synchronous */
hipError_t hipHostFree(void* p);

// ---- launches -----------------------------------------------------------------------------------------------------------------
struct emu_launch_fn { void (*call)(void*); void* closure; };
void emu_launch(dim3 grid, dim3 block, size_t dynamic_lds, emu_launch_fn fn);

template <class K, class... A> static inline void emu_launch_kernel(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t, A... args) {
    auto body = [&]() { kernel(args...); };
    emu_launch_fn fn{[](void* p) { (*static_cast<decltype(body)*>(p))(); }, &body};
    emu_launch(grid, block, lds, fn);
}
#define HIP_KERNEL_NAME(...) __VA_ARGS__
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) emu_launch_kernel(kernel, dim3(grid), dim3(block), (size_t)(lds), stream, ##__VA_ARGS__)
