// Transports of dst_comm (comm.h): RCCL bound at run time, and ranks-as-threads of one process.
#include <dlfcn.h>
#include <functional>
#include <condition_variable>
#include <chrono>
#include <mutex>
#include <thread>
#include <vector>
#include "comm.h"

// ---- RCCL, bound with dlopen -----------------------------------------------------------------------------------------------------------
// The prototypes are the public RCCL / NCCL C API (rccl.h); they are declared here because the library must load without librccl.so.
namespace {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0, ncclInProgress = 7, ncclUint8 = 1 };
struct RcclApi {
    void* handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*CommCount)(ncclComm_t, int*) = nullptr;
    int (*CommCuDevice)(ncclComm_t, int*) = nullptr;
    int (*CommUserRank)(ncclComm_t, int*) = nullptr;
    int (*GetVersion)(int*) = nullptr;
    int (*CommAbort)(ncclComm_t) = nullptr;
    int (*CommGetAsyncError)(ncclComm_t, int*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string error;
};
RcclApi* rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // a process that already holds an RCCL (PyTorch-ROCm ships one) keeps using that copy: same soname
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* nm : names) { api.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL); if (api.handle) break; }
        if (!api.handle) { const char* e = dlerror(); api.error = std::string("librccl.so could not be loaded: ") + (e ? e : "?"); return; }      // dlerror() clears the state: call it once
        auto sym = [&](const char* s) { void* p = dlsym(api.handle, s); if (!p && api.error.empty()) api.error = std::string("librccl.so lacks ") + s; return p; };
        api.GetUniqueId = (int (*)(ncclUniqueId*))sym("ncclGetUniqueId");
        api.CommInitRank = (int (*)(ncclComm_t*, int, ncclUniqueId, int))sym("ncclCommInitRank");
        api.CommDestroy = (int (*)(ncclComm_t))sym("ncclCommDestroy");
        api.AllGather = (int (*)(const void*, void*, size_t, int, ncclComm_t, hipStream_t))sym("ncclAllGather");
        api.Send = (int (*)(const void*, size_t, int, int, ncclComm_t, hipStream_t))sym("ncclSend");
        api.Recv = (int (*)(void*, size_t, int, int, ncclComm_t, hipStream_t))sym("ncclRecv");
        api.GroupStart = (int (*)())sym("ncclGroupStart");
        api.GroupEnd = (int (*)())sym("ncclGroupEnd");
        api.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
        // containment: without these two a stalled collective could not be ended, so they are required like the collectives themselves
        api.CommAbort = (int (*)(ncclComm_t))sym("ncclCommAbort");
        api.CommGetAsyncError = (int (*)(ncclComm_t, int*))sym("ncclCommGetAsyncError");
        // diagnostics only (dst_comm_describe): a librccl without one of them keeps the transport, the field stays 0
        auto opt = [&](const char* s) { return dlsym(api.handle, s); };
        api.CommCount = (int (*)(ncclComm_t, int*))opt("ncclCommCount");
        api.CommCuDevice = (int (*)(ncclComm_t, int*))opt("ncclCommCuDevice");
        api.CommUserRank = (int (*)(ncclComm_t, int*))opt("ncclCommUserRank");
        api.GetVersion = (int (*)(int*))opt("ncclGetVersion");
    });
    return &api;
}

struct RcclComm : dst_comm {
    RcclApi* api = nullptr;
    ncclComm_t comm = nullptr;
    int device = 0;
    hipStream_t own_stream = nullptr;      // host-value gathers
    uint8_t* staging = nullptr; size_t staging_bytes = 0;
    uint8_t* h_land = nullptr; size_t h_land_bytes = 0;          // page-locked landing area of the host-value gathers
    int fail(int r, const char* what) { err = std::string(what) + ": " + (api && api->GetErrorString ? api->GetErrorString(r) : "RCCL error"); return DST_ERR_COMM; }
    ~RcclComm() override {
        if (comm && api) api->CommDestroy(comm);
        if (staging) hipFree(staging);
        if (h_land) hipHostFree(h_land);
        if (own_stream) hipStreamDestroy(own_stream);
    }
    bool stream_ordered() const override { return true; }
    int transport_kind() const override { return DST_COMM_RCCL; }
    void fill_info(dst_comm_info* o) const override {
        int v = 0;
        // what RCCL itself says about this communicator: the number of ranks it connected, this rank's index and device
        if (api->GetVersion && api->GetVersion(&v) == ncclSuccess) o->rccl_version = (uint32_t)v;
        if (!comm) return;                                  // aborted
        if (api->CommCount && api->CommCount(comm, &v) == ncclSuccess) o->rccl_ranks = (uint32_t)v;
        if (api->CommUserRank && api->CommUserRank(comm, &v) == ncclSuccess) o->rccl_rank = (uint32_t)v;
        if (api->CommCuDevice && api->CommCuDevice(comm, &v) == ncclSuccess) o->device = v;
    }
    // what RCCL's own watchdog knows (a peer process that died, a failed transport): looked at while the host polls a stream
    int poll_async() override {
        int state = ncclSuccess;
        if (!comm || api->CommGetAsyncError(comm, &state) != ncclSuccess || state == ncclSuccess || state == ncclInProgress) return DST_OK;
        err = std::string("RCCL reported an asynchronous error: ") + (api->GetErrorString ? api->GetErrorString(state) : "?");
        return DST_ERR_COMM;
    }
    // ncclCommAbort ends the communicator's kernels wherever they wait and frees it: the streams drain, nothing of this rank is left behind
    void abort_impl() override { if (comm) { api->CommAbort(comm); comm = nullptr; } }
    int all_gather_impl(const void* send, void* recv, size_t bytes, hipStream_t stream) override {
        int r = api->AllGather(send, recv, bytes, ncclUint8, comm, stream);          // in place when send == recv + rank * bytes
        if (r != ncclSuccess) return fail(r, "ncclAllGather");
        return DST_OK;
    }
    int all_to_all_impl(const void* send, void* recv, size_t chunk, hipStream_t stream) override {
        int r = api->GroupStart();
        if (r != ncclSuccess) return fail(r, "ncclGroupStart");
        for (uint32_t p = 0; p < world; p++) {
            if ((r = api->Send((const uint8_t*)send + (size_t)p * chunk, chunk, ncclUint8, (int)p, comm, stream)) != ncclSuccess) return fail(r, "ncclSend");
            if ((r = api->Recv((uint8_t*)recv + (size_t)p * chunk, chunk, ncclUint8, (int)p, comm, stream)) != ncclSuccess) return fail(r, "ncclRecv");
        }
        if ((r = api->GroupEnd()) != ncclSuccess) return fail(r, "ncclGroupEnd");
        return DST_OK;
    }
    int all_gather_host_impl(const void* send, void* recv, size_t bytes, hipStream_t stream) override {
        hipStream_t own_stream = stream ? stream : this->own_stream;       // the prover's stream; a caller without one gets the communicator's own
        const size_t need = bytes * (world + 1);
        if (need > staging_bytes) {
            if (staging) hipFree(staging);
            staging_bytes = need < 65536 ? 65536 : need;
            if (hipMalloc((void**)&staging, staging_bytes) != hipSuccess) { staging = nullptr; staging_bytes = 0; err = "all_gather_host: out of device memory"; return DST_ERR_HIP; }
        }
        // both directions go through PAGE-LOCKED memory of the communicator ([0, bytes): this rank's values, behind them the gathered ones): a copy
        // from or into the caller's pageable buffer may block inside hipMemcpyAsync until the stream has run that far -- the all-gather included,
        // i.e. for ever if a peer is missing -- in front of the bounded wait below
        if (need > h_land_bytes) {
            if (h_land) hipHostFree(h_land);
            h_land_bytes = need < ((size_t)1 << 20) ? ((size_t)1 << 20) : need;
            if (hipHostMalloc((void**)&h_land, h_land_bytes, hipHostMallocDefault) != hipSuccess) { h_land = nullptr; h_land_bytes = 0; err = "all_gather_host: out of page-locked memory"; return DST_ERR_HIP; }
        }
        memcpy(h_land, send, bytes);
        if (hipMemcpyAsync(staging, h_land, bytes, hipMemcpyHostToDevice, own_stream) != hipSuccess) { err = "all_gather_host: upload failed"; return DST_ERR_HIP; }
        int r = api->AllGather(staging, staging + bytes, bytes, ncclUint8, comm, own_stream);
        if (r != ncclSuccess) return fail(r, "ncclAllGather");
        if (hipMemcpyAsync(h_land + bytes, staging + bytes, bytes * world, hipMemcpyDeviceToHost, own_stream) != hipSuccess) { err = "all_gather_host: download failed"; return DST_ERR_HIP; }
        const int w = wait_stream(own_stream, "the all-gather of host values");
        if (w == DST_OK) memcpy(recv, h_land + bytes, bytes * world);
        return w;
    }
};

// ---- ranks as threads of one process ----------------------------------------------------------------------------------------------------
struct LocalShared {
    uint32_t world;
    std::mutex mu; std::condition_variable cv;
    uint32_t arrived = 0; uint64_t generation = 0; bool aborted = false;
    std::vector<const void*> ptr;
    std::vector<int> device;                          // device of every rank's thread (-1 until its first device collective)
    uint32_t refs;
    // stream-ordered form (LocalComm::exchange_ordered): collective number k of every rank uses slot k % RING -- the rank's send pointer, an event
    // recorded on its stream when that buffer is ready, one recorded when the rank has queued its copies out of the peers' buffers; `posted` /
    // `copied`: how many collectives a rank has published / has queued the copies of (host side, under `mu`)
    static constexpr uint32_t RING = 4;
    std::vector<const void*> sptr;                    // [RING][world]
    std::vector<hipEvent_t> ready, done;              // [RING][world], created by the owning rank's thread on its device
    std::vector<uint64_t> posted, copied;             // [world]
    explicit LocalShared(uint32_t w) : world(w), ptr(w, nullptr), device(w, -1), refs(w), sptr((size_t)RING * w, nullptr),
                                       ready((size_t)RING * w, nullptr), done((size_t)RING * w, nullptr), posted(w, 0), copied(w, 0) {}
    ~LocalShared() { for (hipEvent_t e : ready) if (e) hipEventDestroy(e); for (hipEvent_t e : done) if (e) hipEventDestroy(e); }
    // host-side rendezvous WITHOUT touching the device: returns when every rank's counter has reached `value` (true), when the group was
    // aborted (false) or after `limit_s` seconds (false, *timed_out; the group is marked aborted so that every peer leaves as well)
    bool wait_counters(const std::vector<uint64_t>& counter, uint64_t value, double limit_s, bool* timed_out) {
        std::unique_lock<std::mutex> lk(mu);
        *timed_out = false;
        auto all = [&] { if (aborted) return true; for (uint64_t c : counter) if (c < value) return false; return true; };
        if (limit_s > 0) {
            if (!cv.wait_for(lk, std::chrono::duration<double>(limit_s), all)) { *timed_out = true; aborted = true; cv.notify_all(); return false; }
        } else cv.wait(lk, all);
        return !aborted;
    }
    void publish(std::vector<uint64_t>& counter, uint32_t rank, uint64_t value) { { std::lock_guard<std::mutex> lk(mu); counter[rank] = value; } cv.notify_all(); }
    // false: another rank gave up (dst_comm_destroy / abort while peers wait) or did not arrive within `limit_s` seconds (<= 0: no limit);
    // *timed_out tells the two apart.  Whoever times out marks the group aborted: every peer leaves its barrier with an error too.
    bool barrier(double limit_s, bool* timed_out) {
        std::unique_lock<std::mutex> lk(mu);
        *timed_out = false;
        if (aborted) return false;
        const uint64_t gen = generation;
        if (++arrived == world) { arrived = 0; generation++; cv.notify_all(); return true; }
        auto done = [&] { return generation != gen || aborted; };
        if (limit_s > 0) {
            if (!cv.wait_for(lk, std::chrono::duration<double>(limit_s), done)) { *timed_out = true; aborted = true; cv.notify_all(); return false; }
        } else cv.wait(lk, done);
        return generation != gen;                      // the generation moved on: the barrier completed (even if someone aborted right after)
    }
    void abort() { { std::lock_guard<std::mutex> lk(mu); aborted = true; } cv.notify_all(); }
};
struct LocalComm : dst_comm {
    LocalShared* sh = nullptr;
    ~LocalComm() override {
        bool last;
        { std::lock_guard<std::mutex> lk(sh->mu); sh->aborted = true; last = --sh->refs == 0; }
        sh->cv.notify_all();
        if (last) delete sh;
    }
    int broken(bool timed_out) {
        dead = true;
        err = timed_out ? "local communicator: a peer rank did not reach " + last_collective() + " within " + std::to_string(timeout_s) + " s; the group was aborted"
                        : "local communicator: a peer rank left the group (aborted or destroyed) at " + last_collective();
        return DST_ERR_COMM;
    }
    void abort_impl() override { sh->abort(); }
    void wake_impl() override { sh->abort(); }         // LocalShared::abort takes the group's mutex: safe from any thread
    int transport_kind() const override { return DST_COMM_LOCAL; }
    // Stream-ordered form (the default; DISTAFF_LOCAL_TRANSPORT=blocking keeps the older one): a collective is only ENQUEUED.  The ranks' host
    // threads meet twice per collective, but only to hand over pointers and events -- nobody waits for a stream, so the devices keep working
    // through the exchanges exactly as under RCCL (dst_prove_sharded runs its two-stream choreography over this transport as well):
    //   record "my send buffer is ready" on my stream | publish | wait until every peer has published | on my stream: wait for each peer's event,
    //   copy its piece (device to device, peer access on a multi-GPU node) | record "I have read" | publish | wait until every peer has |
    //   on my stream: wait for the peers' "have read" events, so that work queued behind the collective may overwrite the send buffer.
    // The blocking form drained every rank's stream twice per collective: measured with 8 thread-ranks sharing one GPU it cost 3.2 - 3.4 ms per
    // rank and proof at config 3 beyond the repeated kernels.
    bool ordered = true;
    uint64_t seq = 0;                                  // device collectives this rank has issued in ordered form
    bool events_made = false;
    bool stream_ordered() const override { return ordered; }
    hipEvent_t& ev(std::vector<hipEvent_t>& v, uint32_t slot, uint32_t r) { return v[(size_t)slot * world + r]; }
    int ensure_events() {
        if (events_made) return DST_OK;
        for (uint32_t s2 = 0; s2 < LocalShared::RING; s2++) {
            if (hipEventCreateWithFlags(&ev(sh->ready, s2, rank), hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ev(sh->done, s2, rank), hipEventDisableTiming) != hipSuccess) {
                (void)hipGetLastError(); err = "local collective: event creation failed"; return DST_ERR_HIP;
            }
        }
        events_made = true;
        return DST_OK;
    }
    int exchange_ordered(const void* send, hipStream_t stream, const std::function<hipError_t(uint32_t peer, const void* peer_send)>& take) {
        const uint64_t k = seq++;
        const uint32_t slot = (uint32_t)(k % LocalShared::RING);
        if (my_device < 0) { if (hipGetDevice(&my_device) != hipSuccess) my_device = -1; sh->device[rank] = my_device; }
        int r = ensure_events();
        hipError_t e = hipSuccess;
        if (r == DST_OK && (e = hipEventRecord(ev(sh->ready, slot, rank), stream)) != hipSuccess) { err = std::string("local collective: ") + hipGetErrorString(e); r = DST_ERR_HIP; }
        if (r != DST_OK) { const std::string why = err; abort(why); return r; }       // this rank cannot take part: its peers must not wait for it
        { std::lock_guard<std::mutex> lk(sh->mu); sh->sptr[(size_t)slot * world + rank] = send; }
        sh->publish(sh->posted, rank, k + 1);
        bool late = false;
        if (!sh->wait_counters(sh->posted, k + 1, timeout_s, &late)) return broken(late);
        if (!peers_checked && my_device >= 0) enable_peer_access();                  // every rank has published its device before its first collective
        for (uint32_t p = 0; p < world && e == hipSuccess; p++) if (p != rank) e = hipStreamWaitEvent(stream, ev(sh->ready, slot, p), 0);
        for (uint32_t p = 0; p < world && e == hipSuccess; p++) e = take(p, sh->sptr[(size_t)slot * world + p]);
        if (e == hipSuccess) e = hipEventRecord(ev(sh->done, slot, rank), stream);
        if (e != hipSuccess) { abort(std::string("local collective: ") + hipGetErrorString(e)); return DST_ERR_HIP; }
        sh->publish(sh->copied, rank, k + 1);
        if (!sh->wait_counters(sh->copied, k + 1, timeout_s, &late)) return broken(late);
        for (uint32_t p = 0; p < world && e == hipSuccess; p++) if (p != rank) e = hipStreamWaitEvent(stream, ev(sh->done, slot, p), 0);
        if (e != hipSuccess) { abort(std::string("local collective: ") + hipGetErrorString(e)); return DST_ERR_HIP; }
        return DST_OK;
    }
    int my_device = -1;
    bool peers_checked = false;
    uint32_t peers_enabled = 0, peers_other_device = 0;
    void fill_info(dst_comm_info* o) const override { o->device = my_device; o->peers_other_device = peers_other_device; o->peers_enabled = peers_enabled; }
    // The ranks' buffers live on different devices when a single-process host drives all GPUs of the node (dst_prove_sharded_local):
    // without peer access a device-to-device hipMemcpy between them is staged through the host instead of crossing xGMI.  Enabled once,
    // from every rank's own thread towards every other rank's device; a pair that cannot be enabled stays on the staged path silently.
    void enable_peer_access() {
        peers_checked = true;
        for (uint32_t p = 0; p < world; p++) {
            const int pd = sh->device[p];
            if (p == rank || pd < 0 || pd == my_device) continue;
            peers_other_device++;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, my_device, pd) != hipSuccess || !can) { (void)hipGetLastError(); continue; }
            const hipError_t e = hipDeviceEnablePeerAccess(pd, 0);
            if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) peers_enabled++;
            (void)hipGetLastError();                   // "already enabled" is not an error of this call site
        }
    }
    int exchange(const void* send, hipStream_t stream, bool host, const std::function<hipError_t(uint32_t peer, const void* peer_send)>& take) {
        // any failure of this rank ends the group: its peers are waiting in the barriers below
        if (!host) { const int w = wait_stream(stream, "the work queued before a collective of the in-process transport"); if (w) { const std::string why = err; abort(why); return w; } }
        if (!host && my_device < 0) { if (hipGetDevice(&my_device) != hipSuccess) my_device = -1; sh->device[rank] = my_device; }
        sh->ptr[rank] = send;
        bool late = false;
        if (!sh->barrier(timeout_s, &late)) return broken(late);
        if (!host && !peers_checked && my_device >= 0) enable_peer_access();      // every rank has published its device before the barrier
        hipError_t e = hipSuccess;
        for (uint32_t p = 0; p < world && e == hipSuccess; p++) e = take(p, sh->ptr[p]);
        if (e != hipSuccess) { abort(std::string("local collective: ") + hipGetErrorString(e)); return DST_ERR_HIP; }
        if (!host) { const int w = wait_stream(stream, "the copies of a collective of the in-process transport"); if (w) { const std::string why = err; abort(why); return w; } }
        if (!sh->barrier(timeout_s, &late)) return broken(late);           // every rank has read every send buffer
        return DST_OK;
    }
    int all_gather_impl(const void* send, void* recv, size_t bytes, hipStream_t stream) override {
        auto take = [&](uint32_t p, const void* src) {
            uint8_t* dst = (uint8_t*)recv + (size_t)p * bytes;
            return dst == src ? hipSuccess : hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, stream);          // in place: the own piece is already there
        };
        return ordered ? exchange_ordered(send, stream, take) : exchange(send, stream, false, take);
    }
    int all_to_all_impl(const void* send, void* recv, size_t chunk, hipStream_t stream) override {
        auto take = [&](uint32_t p, const void* src) { return hipMemcpyAsync((uint8_t*)recv + (size_t)p * chunk, (const uint8_t*)src + (size_t)rank * chunk, chunk, hipMemcpyDefault, stream); };
        return ordered ? exchange_ordered(send, stream, take) : exchange(send, stream, false, take);
    }
    int all_gather_host_impl(const void* send, void* recv, size_t bytes, hipStream_t) override {
        return exchange(send, nullptr, true, [&](uint32_t p, const void* src) { memcpy((uint8_t*)recv + (size_t)p * bytes, src, bytes); return hipSuccess; });
    }
};
// ---- the host's own transport -------------------------------------------------------------------------------------------------------
// kind 0: all-gather of `bytes` per rank (device buffers), 1: all-to-all with chunks of `bytes` (device buffers), 2: all-gather of host values
typedef int (*dst_comm_fn)(void* user, int kind, const void* send, void* recv, size_t bytes);
struct CallbackComm : dst_comm {
    dst_comm_fn fn = nullptr; void* user = nullptr;
    // The host's channel is the host's to bound (a torch.distributed group has its timeout, MPI its own): a callback that reports failure --
    // a peer that stopped answering included -- kills this communicator, and the rank returns DST_ERR_COMM from dst_prove_sharded.
    int call(int kind, const void* send, void* recv, size_t bytes, hipStream_t stream) {
        if (kind != 2) { const int w = wait_stream(stream, "the work queued before a callback collective"); if (w) return w; }
        const int r = fn(user, kind, send, recv, bytes);
        if (r) { dead = true; err = "the host's collective callback returned " + std::to_string(r) + " at " + last_collective(); return DST_ERR_COMM; }
        return DST_OK;
    }
    int transport_kind() const override { return DST_COMM_CALLBACKS; }
    int all_gather_impl(const void* send, void* recv, size_t bytes, hipStream_t stream) override { return call(0, send, recv, bytes, stream); }
    int all_to_all_impl(const void* send, void* recv, size_t chunk, hipStream_t stream) override { return call(1, send, recv, chunk, stream); }
    int all_gather_host_impl(const void* send, void* recv, size_t bytes, hipStream_t) override { return call(2, send, recv, bytes, nullptr); }
};
}  // namespace

#if defined(DISTAFF_TEST_HOOKS) && defined(__HIPCC__)
// one wavefront that holds its stream until the host releases it (or 30 s have passed: the device is never left hanging)
__global__ void comm_stall_kernel(uint32_t* release, unsigned long long max_ticks) {
    const unsigned long long t0 = wall_clock64();                  // constant-rate counter (100 MHz)
    while (__hip_atomic_load(release, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u && wall_clock64() - t0 < max_ticks) __builtin_amdgcn_s_sleep(64);
}
void dst_comm::test_stall(hipStream_t stream) {
    if (stall_at < 0 || (int64_t)issued != stall_at) return;
    if (!stall_flag) { if (hipHostMalloc((void**)&stall_flag, 64, hipHostMallocDefault) != hipSuccess) { stall_flag = nullptr; return; } *stall_flag = 0u; }
    hipLaunchKernelGGL(comm_stall_kernel, dim3(1), dim3(64), 0, stream, stall_flag, 30ull * 100000000ull);
}
// "k" (every rank of the process) or "k@r" (rank r only: thread-ranks share the environment)
static int64_t comm_stall_env(uint32_t rank) {
    const char* e = getenv("DISTAFF_TEST_STALL_COLLECTIVE");
    if (!e || !e[0]) return -1;
    const char* at = strchr(e, '@');
    if (at && (uint32_t)atoll(at + 1) != rank) return -1;
    return atoll(e);
}
#else
void dst_comm::test_stall(hipStream_t) {}
static int64_t comm_stall_env(uint32_t) { return -1; }
#endif

std::string dst_comm::last_collective() const {
    if (!issued) return "the first collective (none issued yet)";
    const char* k = last_kind == 'G' ? "all-gather" : last_kind == 'A' ? "all-to-all" : "all-gather of host values";
    return "collective #" + std::to_string(issued - 1) + " (" + k + ", " + std::to_string(last_bytes) + " bytes per rank)";
}

// Bounded wait for everything queued on `stream` so far.  hipStreamSynchronize cannot be interrupted; the host polls instead (the runtime's
// own wait spins too), and every 64 queries looks at the transport's asynchronous error state and at the clock.
int dst_comm::wait_stream(hipStream_t stream, const char* what) {
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 1;; spins++) {
        const hipError_t e = hipStreamQuery(stream);
        if (e == hipSuccess) return DST_OK;
        (void)hipGetLastError();                           // "not ready" is not an error of this call site
        if (e != hipErrorNotReady) { err = std::string("waiting for ") + what + ": " + hipGetErrorString(e); return DST_ERR_HIP; }
        if (spins & 63u) continue;
        if (check_requested()) return DST_ERR_COMM;        // dst_comm_abort from another thread
        if (poll_async() != DST_OK) { const std::string why = err + " (while waiting for " + what + ", last issued: " + last_collective() + ")"; abort(why); return DST_ERR_COMM; }
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (timeout_s > 0 && waited > timeout_s) {
            char lim[32]; snprintf(lim, sizeof lim, "%.1f", timeout_s);
            abort(std::string("no completion within ") + lim + " s while waiting for " + what + "; last issued: " + last_collective() + "; the communicator was aborted");
            // the abort ends the transport's kernels: give the stream a moment to drain so that the caller's buffers are quiet when it returns
            for (int i = 0; i < 2000 && hipStreamQuery(stream) == hipErrorNotReady; i++) std::this_thread::sleep_for(std::chrono::milliseconds(1));
            (void)hipGetLastError();
            return DST_ERR_COMM;
        }
        if (waited > 0.002) std::this_thread::yield();      // a long wait (a peer is behind): let the host's other threads run
    }
}

int ctx_sync(dst_ctx* c, const char* what) {
    if (c->wait_comm) {
        const int r = c->wait_comm->wait_stream(c->stream, what);
        if (r) c->err = std::string("waiting for ") + what + ": " + c->wait_comm->err;
        return r;
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DST_OK;
}

static double comm_timeout_env() { const char* e = getenv("DISTAFF_COMM_TIMEOUT_S"); if (!e || !e[0]) return 60.0; char* end = nullptr; const double v = strtod(e, &end); return end == e ? 60.0 : v; }
static bool local_blocking_env() { const char* e = getenv("DISTAFF_LOCAL_TRANSPORT"); return e && !strcmp(e, "blocking"); }
static bool shard_debug_env() { const char* e = getenv("DISTAFF_SHARD_DEBUG"); return e && e[0] && e[0] != '0'; }
static thread_local std::string g_comm_error;      // errors before a communicator exists, per calling thread (ranks may be threads)

extern "C" {

const char* dst_comm_last_error(const dst_comm* comm) { return comm ? comm->err.c_str() : g_comm_error.c_str(); }

int dst_comm_unique_id(uint8_t id[128]) {
    if (!id) return DST_ERR_ARG;
    RcclApi* api = rccl_api();
    if (!api->error.empty()) { g_comm_error = api->error; return DST_ERR_HIP; }
    ncclUniqueId u;
    int r = api->GetUniqueId(&u);
    if (r != ncclSuccess) { g_comm_error = std::string("ncclGetUniqueId: ") + api->GetErrorString(r); return DST_ERR_HIP; }
    memcpy(id, u.internal, 128);
    return DST_OK;
}

int dst_comm_init(const uint8_t id[128], uint32_t rank, uint32_t world, int device, dst_comm** out) {
    if (!id || !out || world == 0 || rank >= world) { g_comm_error = "dst_comm_init: bad arguments"; return DST_ERR_ARG; }
    RcclApi* api = rccl_api();
    if (!api->error.empty()) { g_comm_error = api->error; return DST_ERR_HIP; }
    if (hipSetDevice(device) != hipSuccess) { g_comm_error = "dst_comm_init: hipSetDevice failed"; return DST_ERR_HIP; }
    RcclComm* c = new RcclComm();
    c->api = api; c->rank = rank; c->world = world; c->device = device;
    ncclUniqueId u; memcpy(u.internal, id, 128);
    int r = api->CommInitRank(&c->comm, (int)world, u, (int)rank);
    if (r != ncclSuccess) { g_comm_error = std::string("ncclCommInitRank: ") + api->GetErrorString(r); c->comm = nullptr; delete c; return DST_ERR_HIP; }
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) { g_comm_error = "dst_comm_init: stream creation failed"; delete c; return DST_ERR_HIP; }
    // the staging area of the host-value all-gathers, NOW: hipMalloc synchronises the device, and in the middle of a proof that would be an
    // unbounded wait behind whatever collectives are in flight (1 MiB holds the opening blobs of 8 ranks several times over; it still grows on demand)
    c->staging_bytes = (size_t)1 << 20;
    if (hipMalloc((void**)&c->staging, c->staging_bytes) != hipSuccess) { (void)hipGetLastError(); c->staging = nullptr; c->staging_bytes = 0; }
    c->h_land_bytes = (size_t)1 << 20;
    if (hipHostMalloc((void**)&c->h_land, c->h_land_bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); c->h_land = nullptr; c->h_land_bytes = 0; }
    c->tracing = shard_debug_env(); c->timeout_s = comm_timeout_env(); c->stall_at = comm_stall_env(c->rank);
    *out = c;
    return DST_OK;
}

int dst_comm_init_local(uint32_t world, dst_comm** out) {
    if (!out || world == 0 || world > 64) { g_comm_error = "dst_comm_init_local: bad arguments"; return DST_ERR_ARG; }
    LocalShared* sh = new LocalShared(world);
    for (uint32_t r = 0; r < world; r++) { LocalComm* c = new LocalComm(); c->rank = r; c->world = world; c->sh = sh; c->tracing = shard_debug_env(); c->timeout_s = comm_timeout_env(); c->stall_at = comm_stall_env(c->rank); c->ordered = !local_blocking_env(); out[r] = c; }
    return DST_OK;
}

int dst_comm_init_callbacks(uint32_t rank, uint32_t world, dst_comm_fn fn, void* user, dst_comm** out) {
    if (!out || !fn || world == 0 || rank >= world) { g_comm_error = "dst_comm_init_callbacks: bad arguments"; return DST_ERR_ARG; }
    CallbackComm* c = new CallbackComm();
    c->rank = rank; c->world = world; c->fn = fn; c->user = user;
    c->tracing = shard_debug_env(); c->timeout_s = comm_timeout_env(); c->stall_at = comm_stall_env(c->rank);
    *out = c;
    return DST_OK;
}

void dst_comm_destroy(dst_comm* comm) { delete comm; }

// limit of every host wait behind a collective of this communicator, seconds (default 60, or DISTAFF_COMM_TIMEOUT_S at creation); <= 0: none
int dst_comm_set_timeout(dst_comm* comm, double seconds) { if (!comm) return DST_ERR_ARG; comm->timeout_s = seconds; return DST_OK; }
// Gives the communicator up from the host's side (a watchdog that learnt of a dead peer, a shutdown): RCCL's kernels end, in-process peers
// leave their barriers, every later collective of this handle returns DST_ERR_COMM.  The handle still has to be destroyed.
// May be called from another thread than the one inside dst_prove_sharded: it only marks the handle (and wakes in-process peers); the rank's
// own thread tears the transport down at its next poll or collective.
int dst_comm_abort(dst_comm* comm) { if (!comm) return DST_ERR_ARG; comm->request_abort(); return DST_OK; }

int dst_comm_describe(const dst_comm* comm, dst_comm_info* out) {
    if (!comm || !out) return DST_ERR_ARG;
    memset(out, 0, sizeof(*out));
    out->transport = (uint32_t)comm->transport_kind(); out->rank = comm->rank; out->world = comm->world; out->device = -1;
    comm->fill_info(out);
    return DST_OK;
}

// enable: 1 start (clears the record), 0 stop, -1 leave as is.  Writes the record so far as text, one collective per line:
// "<kind> <bytes per rank> <stream>" with kind G all-gather, A all-to-all, H all-gather of host values; stream = index of the stream
// among those this communicator has queued collectives on, in order of first use ('-' for host values).
int dst_comm_trace(dst_comm* comm, int enable, char* out, size_t cap, size_t* len) {
    if (!comm) return DST_ERR_ARG;
    std::string text;
    for (const auto& r : comm->trace) {
        text += r.kind; text += ' '; text += std::to_string(r.bytes); text += ' ';
        text += r.stream == 255 ? std::string("-") : std::to_string((unsigned)r.stream); text += '\n';
    }
    if (len) *len = text.size();
    if (out && cap) { const size_t k = text.size() < cap - 1 ? text.size() : cap - 1; memcpy(out, text.data(), k); out[k] = 0; }
    if (enable == 1) { comm->tracing = true; comm->trace.clear(); comm->seen_streams.clear(); }
    else if (enable == 0) comm->tracing = false;
    return DST_OK;
}

// For hosts whose own transport (dst_comm_init_callbacks) stages through memory the library did not allocate: one synchronous copy
// between any two of {host memory, memory of the current device}; the direction follows from the pointers.
int dst_comm_copy(void* dst, const void* src, size_t bytes) {
    if ((!dst || !src) && bytes) { g_comm_error = "dst_comm_copy: null pointer"; return DST_ERR_ARG; }
    if (bytes == 0 || dst == src) return DST_OK;
    const hipError_t e = hipMemcpy(dst, src, bytes, hipMemcpyDefault);
    if (e != hipSuccess) { g_comm_error = std::string("dst_comm_copy: ") + hipGetErrorString(e); return DST_ERR_HIP; }
    return DST_OK;
}

}  // extern "C"
