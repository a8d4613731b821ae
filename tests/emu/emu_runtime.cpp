// TEST INFRASTRUCTURE (see hip/hip_runtime.h in this directory): host execution of kernel launches for libdistaff_emu.so.
//
// A launch runs its workgroups on a small pool of OS threads; inside a workgroup every work-item is a fiber with its own stack
// and the fibers are resumed round-robin: a fiber runs until it returns or reaches __syncthreads(), so one sweep over the fibers
// is one barrier phase.  "Device memory" is host memory, streams are synchronous, events are wall-clock time stamps.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include <sys/mman.h>

thread_local uint3 threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;

// dynamic LDS of the kernels that declare `extern __shared__` arrays (distaff_amd/csrc/kernels_ntt.hip)
alignas(16) thread_local unsigned char ntt_smem[160 * 1024];
alignas(16) thread_local unsigned char fold_smem[160 * 1024];

// ---- fibers (x86-64 System V: callee-saved registers + stack pointer) ----------------------------------------------------------
extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .globl emu_switch
    .type emu_switch, @function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size emu_switch, .-emu_switch
)");

namespace {
constexpr size_t kStack = 512 * 1024;
constexpr unsigned kMaxItems = 1024;

struct Worker {
    unsigned char* stacks = nullptr;             // kMaxItems stacks, touched lazily
    void* scheduler_sp = nullptr;
    void* item_sp[kMaxItems];
    bool done[kMaxItems];
    unsigned current = 0;
    emu_launch_fn fn{};
};
thread_local Worker tl_worker;

extern "C" void emu_fiber_entry() {
    Worker& w = tl_worker;
    w.fn.call(w.fn.closure);
    w.done[w.current] = true;
    void* dummy;
    emu_switch(&dummy, w.scheduler_sp);          // never resumed
    abort();
}

void run_workgroup(Worker& w, dim3 block) {
    const unsigned items = block.x * block.y * block.z;
    if (items > kMaxItems) { fprintf(stderr, "emu: workgroup of %u items\n", items); abort(); }
    if (!w.stacks) {
        w.stacks = (unsigned char*)mmap(nullptr, kStack * kMaxItems, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (w.stacks == MAP_FAILED) { perror("emu: mmap"); abort(); }
    }
    for (unsigned t = 0; t < items; t++) {
        // initial frame: six callee-saved registers, then the entry point as return address; rsp % 16 == 8 at entry like after a call
        uintptr_t top = (uintptr_t)(w.stacks + (size_t)(t + 1) * kStack);
        top &= ~(uintptr_t)15;
        void** sp = (void**)(top - 8);
        *--sp = (void*)emu_fiber_entry;
        for (int r = 0; r < 6; r++) *--sp = nullptr;
        w.item_sp[t] = sp;
        w.done[t] = false;
    }
    unsigned alive = items;
    while (alive) {
        for (unsigned t = 0; t < items; t++) {
            if (w.done[t]) continue;
            w.current = t;
            threadIdx.x = t % block.x; threadIdx.y = (t / block.x) % block.y; threadIdx.z = t / (block.x * block.y);
            emu_switch(&w.scheduler_sp, w.item_sp[t]);
            if (w.done[t]) alive--;
        }
    }
}
}  // namespace

void emu_syncthreads() {
    Worker& w = tl_worker;
    emu_switch(&w.item_sp[w.current], w.scheduler_sp);
}

namespace {
// persistent pool: launches are serialised (one at a time, like one stream), their workgroups are spread over the pool
struct Pool {
    std::mutex launch_mu, mu;
    std::condition_variable cv_start, cv_done;
    uint64_t generation = 0;
    unsigned pending = 0;
    std::function<void()> job;
    std::vector<std::thread> threads;
};
Pool* pool() {
    static Pool* p = [] {
        Pool* q = new Pool;                       // never destroyed: the workers may still wait on it at process exit
        const char* e = getenv("DISTAFF_EMU_THREADS");
        unsigned n = e ? (unsigned)atoi(e) : std::thread::hardware_concurrency();
        if (n == 0) n = 1;
        for (unsigned i = 1; i < n; i++) {
            q->threads.emplace_back([q] {
                uint64_t seen = 0;
                for (;;) {
                    std::function<void()> job;
                    {
                        std::unique_lock<std::mutex> lk(q->mu);
                        q->cv_start.wait(lk, [&] { return q->generation != seen; });
                        seen = q->generation;
                        job = q->job;
                    }
                    job();
                    {
                        std::lock_guard<std::mutex> lk(q->mu);
                        if (--q->pending == 0) q->cv_done.notify_all();
                    }
                }
            });
            q->threads.back().detach();
        }
        return q;
    }();
    return p;
}
}  // namespace

void emu_launch(dim3 grid, dim3 block, size_t dynamic_lds, emu_launch_fn fn) {
    if (dynamic_lds > sizeof(ntt_smem)) { fprintf(stderr, "emu: %zu bytes of dynamic LDS\n", dynamic_lds); abort(); }
    const size_t groups = (size_t)grid.x * grid.y * grid.z;
    if (groups == 0 || block.x * block.y * block.z == 0) { fprintf(stderr, "emu: launch with an empty grid or block (the device would reject it)\n"); abort(); }
    std::atomic<size_t> next{0};
    auto work = [&]() {
        Worker& w = tl_worker;
        w.fn = fn;
        blockDim = block; gridDim = grid;
        for (;;) {
            const size_t g = next.fetch_add(1);
            if (g >= groups) break;
            blockIdx.x = (unsigned)(g % grid.x); blockIdx.y = (unsigned)((g / grid.x) % grid.y); blockIdx.z = (unsigned)(g / ((size_t)grid.x * grid.y));
            run_workgroup(w, block);
        }
    };
    Pool* p = pool();
    std::lock_guard<std::mutex> launch(p->launch_mu);
    if (groups < 4 || p->threads.empty()) { work(); return; }
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->job = work;
        p->pending = (unsigned)p->threads.size();
        p->generation++;
    }
    p->cv_start.notify_all();
    work();
    std::unique_lock<std::mutex> lk(p->mu);
    p->cv_done.wait(lk, [&] { return p->pending == 0; });
}

unsigned long long emu_atomic_min_u64(unsigned long long* p, unsigned long long v) {
    auto* a = reinterpret_cast<std::atomic<unsigned long long>*>(p);
    unsigned long long old = a->load();
    while (v < old && !a->compare_exchange_weak(old, v)) {}
    return old;
}
unsigned emu_atomic_add_u32(unsigned* p, unsigned v) { return reinterpret_cast<std::atomic<unsigned>*>(p)->fetch_add(v); }

// ---- runtime API ---------------------------------------------------------------------------------------------------------------
struct emu_stream { int unused; };
struct emu_event { std::chrono::steady_clock::time_point t; };

const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorOutOfMemory ? "out of memory (emulated)" : "error (emulated)"; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d >= 0 && d < 64 ? hipSuccess : hipErrorInvalidValue; }    // one host stands in for every ordinal
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    memset(p, 0, sizeof(*p));
    strcpy(p->name, "host emulation"); strcpy(p->gcnArchName, "emulated");
    p->totalGlobalMem = (size_t)8 << 30; p->multiProcessorCount = 1; p->sharedMemPerBlock = sizeof(ntt_smem);
    return hipSuccess;
}
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipMalloc(void** p, size_t bytes) {
    *p = nullptr;
    if (posix_memalign(p, 256, bytes ? bytes : 256)) return hipErrorOutOfMemory;
    return hipSuccess;
}
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) { return hipMalloc(p, bytes); }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = *t = (size_t)8 << 30; return hipSuccess; }
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind) { memmove(dst, src, bytes); return hipSuccess; }
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind, hipStream_t) { memmove(dst, src, bytes); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t) {
    for (size_t r = 0; r < height; r++) memmove((char*)dst + r * dpitch, (const char*)src + r * spitch, width);
    return hipSuccess;
}
hipError_t hipMemset(void* dst, int value, size_t bytes) { memset(dst, value, bytes); return hipSuccess; }
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t) { memset(dst, value, bytes); return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t* s) { *s = new emu_stream{0}; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new emu_event{}; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }
