// Transports of dst_comm (comm.h): RCCL bound at run time, and ranks-as-threads of one process.
#include <dlfcn.h>
#include <functional>
#include <condition_variable>
#include <mutex>
#include <vector>
#include "comm.h"

// ---- RCCL, bound with dlopen -----------------------------------------------------------------------------------------------------------
// The prototypes are the public RCCL / NCCL C API (rccl.h); they are declared here because the library must load without librccl.so.
namespace {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0, ncclUint8 = 1 };
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
        api.CommCount = (int (*)(ncclComm_t, int*))sym("ncclCommCount");
        api.CommCuDevice = (int (*)(ncclComm_t, int*))sym("ncclCommCuDevice");
        api.CommUserRank = (int (*)(ncclComm_t, int*))sym("ncclCommUserRank");
        api.GetVersion = (int (*)(int*))sym("ncclGetVersion");
    });
    return &api;
}

struct RcclComm : dst_comm {
    RcclApi* api = nullptr;
    ncclComm_t comm = nullptr;
    int device = 0;
    hipStream_t own_stream = nullptr;      // host-value gathers
    uint8_t* staging = nullptr; size_t staging_bytes = 0;
    int fail(int r, const char* what) { err = std::string(what) + ": " + (api && api->GetErrorString ? api->GetErrorString(r) : "RCCL error"); return DST_ERR_HIP; }
    ~RcclComm() override {
        if (comm && api) api->CommDestroy(comm);
        if (staging) hipFree(staging);
        if (own_stream) hipStreamDestroy(own_stream);
    }
    bool stream_ordered() const override { return true; }
    int transport_kind() const override { return DST_COMM_RCCL; }
    void fill_info(dst_comm_info* o) const override {
        int v = 0;
        // what RCCL itself says about this communicator: the number of ranks it connected, this rank's index and device
        if (api->CommCount(comm, &v) == ncclSuccess) o->rccl_ranks = (uint32_t)v;
        if (api->CommUserRank(comm, &v) == ncclSuccess) o->rccl_rank = (uint32_t)v;
        if (api->CommCuDevice(comm, &v) == ncclSuccess) o->device = v;
        if (api->GetVersion && api->GetVersion(&v) == ncclSuccess) o->rccl_version = (uint32_t)v;
    }
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
        if (hipMemcpyAsync(staging, send, bytes, hipMemcpyHostToDevice, own_stream) != hipSuccess) { err = "all_gather_host: upload failed"; return DST_ERR_HIP; }
        int r = api->AllGather(staging, staging + bytes, bytes, ncclUint8, comm, own_stream);
        if (r != ncclSuccess) return fail(r, "ncclAllGather");
        if (hipMemcpyAsync(recv, staging + bytes, bytes * world, hipMemcpyDeviceToHost, own_stream) != hipSuccess || hipStreamSynchronize(own_stream) != hipSuccess) {
            err = "all_gather_host: download failed"; return DST_ERR_HIP;
        }
        return DST_OK;
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
    explicit LocalShared(uint32_t w) : world(w), ptr(w, nullptr), device(w, -1), refs(w) {}
    bool barrier() {                                   // false: another rank gave up (dst_comm_destroy while peers wait)
        std::unique_lock<std::mutex> lk(mu);
        if (aborted) return false;
        const uint64_t gen = generation;
        if (++arrived == world) { arrived = 0; generation++; cv.notify_all(); return true; }
        cv.wait(lk, [&] { return generation != gen || aborted; });
        return !aborted;
    }
};
struct LocalComm : dst_comm {
    LocalShared* sh = nullptr;
    ~LocalComm() override {
        bool last;
        { std::lock_guard<std::mutex> lk(sh->mu); sh->aborted = true; last = --sh->refs == 0; }
        sh->cv.notify_all();
        if (last) delete sh;
    }
    int broken() { err = "local communicator: a peer rank left the collective"; return DST_ERR_STATE; }
    int transport_kind() const override { return DST_COMM_LOCAL; }
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
        if (!host && hipStreamSynchronize(stream) != hipSuccess) { err = "local collective: stream synchronisation failed"; return DST_ERR_HIP; }
        if (!host && my_device < 0) { if (hipGetDevice(&my_device) != hipSuccess) my_device = -1; sh->device[rank] = my_device; }
        sh->ptr[rank] = send;
        if (!sh->barrier()) return broken();
        if (!host && !peers_checked && my_device >= 0) enable_peer_access();      // every rank has published its device before the barrier
        hipError_t e = hipSuccess;
        for (uint32_t p = 0; p < world && e == hipSuccess; p++) e = take(p, sh->ptr[p]);
        if (!host && e == hipSuccess) e = hipStreamSynchronize(stream);
        if (!sh->barrier()) return broken();           // every rank has read every send buffer
        if (e != hipSuccess) { err = std::string("local collective: ") + hipGetErrorString(e); return DST_ERR_HIP; }
        return DST_OK;
    }
    int all_gather_impl(const void* send, void* recv, size_t bytes, hipStream_t stream) override {
        return exchange(send, stream, false, [&](uint32_t p, const void* src) {
            uint8_t* dst = (uint8_t*)recv + (size_t)p * bytes;
            return dst == src ? hipSuccess : hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, stream);          // in place: the own piece is already there
        });
    }
    int all_to_all_impl(const void* send, void* recv, size_t chunk, hipStream_t stream) override {
        return exchange(send, stream, false, [&](uint32_t p, const void* src) { return hipMemcpyAsync((uint8_t*)recv + (size_t)p * chunk, (const uint8_t*)src + (size_t)rank * chunk, chunk, hipMemcpyDefault, stream); });
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
    int call(int kind, const void* send, void* recv, size_t bytes, hipStream_t stream) {
        if (kind != 2 && hipStreamSynchronize(stream) != hipSuccess) { err = "callback collective: stream synchronisation failed"; return DST_ERR_HIP; }
        const int r = fn(user, kind, send, recv, bytes);
        if (r) { err = "the host's collective callback returned " + std::to_string(r); return DST_ERR_STATE; }
        return DST_OK;
    }
    int transport_kind() const override { return DST_COMM_CALLBACKS; }
    int all_gather_impl(const void* send, void* recv, size_t bytes, hipStream_t stream) override { return call(0, send, recv, bytes, stream); }
    int all_to_all_impl(const void* send, void* recv, size_t chunk, hipStream_t stream) override { return call(1, send, recv, chunk, stream); }
    int all_gather_host_impl(const void* send, void* recv, size_t bytes, hipStream_t) override { return call(2, send, recv, bytes, nullptr); }
};
}  // namespace

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
    c->tracing = shard_debug_env();
    *out = c;
    return DST_OK;
}

int dst_comm_init_local(uint32_t world, dst_comm** out) {
    if (!out || world == 0 || world > 64) { g_comm_error = "dst_comm_init_local: bad arguments"; return DST_ERR_ARG; }
    LocalShared* sh = new LocalShared(world);
    for (uint32_t r = 0; r < world; r++) { LocalComm* c = new LocalComm(); c->rank = r; c->world = world; c->sh = sh; c->tracing = shard_debug_env(); out[r] = c; }
    return DST_OK;
}

int dst_comm_init_callbacks(uint32_t rank, uint32_t world, dst_comm_fn fn, void* user, dst_comm** out) {
    if (!out || !fn || world == 0 || rank >= world) { g_comm_error = "dst_comm_init_callbacks: bad arguments"; return DST_ERR_ARG; }
    CallbackComm* c = new CallbackComm();
    c->rank = rank; c->world = world; c->fn = fn; c->user = user;
    c->tracing = shard_debug_env();
    *out = c;
    return DST_OK;
}

void dst_comm_destroy(dst_comm* comm) { delete comm; }

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
