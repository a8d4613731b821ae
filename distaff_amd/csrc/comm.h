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
#include <string>
#include "ctx.h"

struct dst_comm {
    uint32_t rank = 0, world = 1;
    std::string err;
    virtual ~dst_comm() {}
    // device buffers.  Work queued on `stream` before the call is ordered before the exchange, work queued after it sees `recv` complete
    // and may overwrite `send`.  A transport that is stream_ordered() only ENQUEUES the exchange (RCCL: the host does not wait, so
    // collectives overlap with kernels of other streams); the others return when the exchange is complete.  The host synchronises the
    // stream before it reads results.  all_gather may be in place: send == recv + rank * bytes_per_rank.
    virtual bool stream_ordered() const { return false; }
    virtual int all_gather(const void* send, void* recv, size_t bytes_per_rank, hipStream_t stream) = 0;      // recv = [world][bytes]
    virtual int all_to_all(const void* send, void* recv, size_t chunk_bytes, hipStream_t stream) = 0;         // chunk g of send -> rank g; chunk r of recv <- rank r
    // small host values (status words, lengths, opening blobs)
    virtual int all_gather_host(const void* send, void* recv, size_t bytes_per_rank) = 0;
};
