// Collectives of the sharded prover behind the C-ABI (dst_comm_* / dst_prove_sharded in include/distaff_hip.h).
//
// One communicator handle per rank.  Two transports implement the same three operations:
//   * RCCL over xGMI (one process per GPU): librccl.so is bound at run time (dlopen) when the first RCCL communicator is created, so
//     a single-GPU host never needs the library; the rank-0 host creates the unique id (dst_comm_unique_id) and hands it to the
//     other ranks through whatever channel it has (the Rust host's own process launcher, a file, MPI, torch.distributed ...);
//   * "local": the ranks are threads of ONE process (tests on one GPU or on the host emulation, and a single-process multi-GPU
//     host); device buffers are exchanged with device-to-device copies between two barriers.
// The reference has no multi-device path (src/math/polynom.rs:36-37 fixes one thread); the exchange points are SURVEY.md 8(e).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <atomic>
#include <string>
#include <vector>
#include "ctx.h"

struct dst_comm {
    uint32_t rank = 0, world = 1;
    std::string err;
    // Containment (the reference panics, src/lib.rs:32,49,56; the C-ABI promises an error code on every rank instead).  A collective whose
    // peer never arrives cannot fail by itself -- the stream it was queued on simply stops -- so no host wait behind a collective is
    // unbounded: wait_stream polls the stream, asks the transport for asynchronous errors, and after `timeout_s` without completion ABORTS the
    // communicator (RCCL: ncclCommAbort, which also ends the kernels stuck on the stream; in-process: wakes every peer out of its barrier).
    // A dead communicator refuses every further collective at once, so the rank leaves dst_prove_sharded through its ordinary error path.
    double timeout_s = 60.0;               // DISTAFF_COMM_TIMEOUT_S at creation, dst_comm_set_timeout afterwards; <= 0: wait without limit
    bool dead = false;
    std::atomic<bool> abort_requested{false};     // dst_comm_abort from ANOTHER thread: the rank's own thread performs the abort at its next check
    uint64_t issued = 0;                   // collectives issued so far (counted whether or not the record below is on)
    char last_kind = 0; uint64_t last_bytes = 0;
    virtual ~dst_comm() { if (stall_flag) { *(volatile uint32_t*)stall_flag = 1u; (void)hipHostFree(stall_flag); } }
    int wait_stream(hipStream_t stream, const char* what);      // comm.hip: bounded wait for everything queued on `stream`
    void abort(const std::string& why) { if (stall_flag) *(volatile uint32_t*)stall_flag = 1u; if (!dead) { dead = true; abort_impl(); } err = why; }
    // the host's side (dst_comm_abort; any thread): mark, and wake whoever can be woken without touching the transport's state
    void request_abort() { abort_requested.store(true); if (stall_flag) *(volatile uint32_t*)stall_flag = 1u; wake_impl(); }
    bool check_requested() { if (!dead && abort_requested.load()) abort("the host aborted the communicator (dst_comm_abort)"); return dead; }
    std::string last_collective() const;   // "collective #k (all-to-all, 1048576 bytes per rank)"
    // Fault injection of the TEST build (DISTAFF_TEST_STALL_COLLECTIVE=k at creation; nothing in the product library): before this rank's
    // device collective number k a kernel is queued on the collective's stream that waits for a host flag only abort() sets -- the stream
    // stops exactly as it does under a collective whose peer never arrives, on one GPU.  (The kernel gives up by itself after 30 s.)
    int64_t stall_at = -1;
    uint32_t* stall_flag = nullptr;        // page-locked
    void test_stall(hipStream_t stream);
    // device buffers.  Work queued on `stream` before the call is ordered before the exchange, work queued after it sees `recv` complete
    // and may overwrite `send`.  A transport that is stream_ordered() only ENQUEUES the exchange (RCCL: the host does not wait, so
    // collectives overlap with kernels of other streams); the others return when the exchange is complete.  The host synchronises the
    // stream before it reads results.  all_gather may be in place: send == recv + rank * bytes_per_rank.
    virtual bool stream_ordered() const { return false; }
    int all_gather(const void* send, void* recv, size_t bytes_per_rank, hipStream_t stream) { if (check_requested()) return refused(); test_stall(stream); note('G', bytes_per_rank, stream); return all_gather_impl(send, recv, bytes_per_rank, stream); }   // recv = [world][bytes]
    int all_to_all(const void* send, void* recv, size_t chunk_bytes, hipStream_t stream) { if (check_requested()) return refused(); test_stall(stream); note('A', chunk_bytes, stream); return all_to_all_impl(send, recv, chunk_bytes, stream); }           // chunk g of send -> rank g; chunk r of recv <- rank r
    // small host values (status words, lengths, opening blobs); complete on return.  `stream`: the caller's stream -- a transport that moves
    // the values through the device (RCCL) queues them THERE, so that a communicator sees its collectives on the streams of the prover only
    int all_gather_host(const void* send, void* recv, size_t bytes_per_rank, hipStream_t stream) { if (check_requested()) return refused(); note('H', bytes_per_rank, nullptr); return all_gather_host_impl(send, recv, bytes_per_rank, stream); }
    int refused() { if (err.empty()) err = "the communicator was aborted"; return DST_ERR_COMM; }       // err keeps the reason of the abort

    // Issue-order record (dst_comm_trace; DISTAFF_SHARD_DEBUG=1 switches it on at creation): one entry per collective -- kind, bytes per
    // rank, and which of the streams this communicator has seen it was queued on (index by first appearance; '-' = host values).  RCCL
    // requires every rank to issue the collectives of a communicator in the same order; a transport that forgives a mismatch (gloo, the
    // in-process one) hides it, so the multi-process tests compare these records across ranks.
    struct CollRec { char kind; uint8_t stream; uint64_t bytes; };
    bool tracing = false;
    std::vector<CollRec> trace;
    std::vector<hipStream_t> seen_streams;
    void note(char kind, size_t bytes, hipStream_t s) {
        issued++; last_kind = kind; last_bytes = (uint64_t)bytes;
        if (!tracing) return;
        uint8_t tag = 255;
        if (kind != 'H') {
            size_t i = 0;
            while (i < seen_streams.size() && seen_streams[i] != s) i++;
            if (i == seen_streams.size()) seen_streams.push_back(s);
            tag = (uint8_t)i;
        }
        if (trace.size() < 65536) trace.push_back(CollRec{kind, tag, (uint64_t)bytes});      // a record for tests / diagnosis, not a log: bounded
    }
    // what dst_comm_describe reports
    virtual int transport_kind() const = 0;                      // DST_COMM_RCCL / DST_COMM_LOCAL / DST_COMM_CALLBACKS
    virtual void fill_info(dst_comm_info* out) const {}
protected:
    virtual int poll_async() { return DST_OK; }                   // RCCL: ncclCommGetAsyncError; != DST_OK with `err` set when the transport has failed
    virtual void abort_impl() {}                                  // transport-specific part of abort() (the rank's own thread)
    virtual void wake_impl() {}                                   // request_abort(): thread-safe wake-up of ranks blocked in this transport
    virtual int all_gather_impl(const void* send, void* recv, size_t bytes_per_rank, hipStream_t stream) = 0;
    virtual int all_to_all_impl(const void* send, void* recv, size_t chunk_bytes, hipStream_t stream) = 0;
    virtual int all_gather_host_impl(const void* send, void* recv, size_t bytes_per_rank, hipStream_t stream) = 0;
};
