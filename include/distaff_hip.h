/* libdistaff_hip.so -- C-ABI of the MI355X-native STARK prover backend for Distaff.
 *
 * The reference (GuildOfWeavers/distaff v0.5.1, pure Rust) has no FFI layer; the drop-in seam is the body of
 * `stark::prove` (/root/reference/src/stark/prover.rs:17), which is called from exactly one place
 * (/root/reference/src/lib.rs:62).  A Rust maintainer replaces that body with calls to the entry points below
 * (INTEGRATION.md shows the `extern "C"` block); Fiat-Shamir dependencies between the nine prover steps force the
 * boundary to be phase-wise, and `dst_prove` chains the phases with the library's own restatement of the
 * reference's challenge derivation.
 *
 * Conventions
 *   - a field element is 16 bytes, little-endian, canonical (< p = 2^128 - 45*2^40 + 1); this is the memory image of
 *     the reference's `u128` (/root/reference/src/utils/mod.rs:35-41).  Pointers need only byte alignment.
 *   - a digest is 32 bytes (BLAKE3, the only serialisable `HashFunction`: src/stark/options.rs:97-120).
 *   - every function returns DST_OK (0) or a negative error code and never aborts the process (the reference
 *     panics instead: src/lib.rs:32,49,56, src/stark/constraints/evaluator.rs:155); `dst_last_error` gives the text.
 *   - a context owns all device memory and one HIP stream; it is not thread-safe; calls are synchronous (they
 *     return after the stream has drained) unless stated otherwise.  Host buffers are caller-owned and are only
 *     read or written during the call.
 */
#ifndef DISTAFF_HIP_H
#define DISTAFF_HIP_H
#include <stddef.h>
#include <stdint.h>

/* exported entry points (the shared library is built with hidden visibility: nothing else leaves it) */
#if defined(__GNUC__)
#define DST_API __attribute__((visibility("default")))
#else
#define DST_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define DST_OK 0
#define DST_ERR_ARG (-1)      /* invalid argument / call order */
#define DST_ERR_HIP (-2)      /* HIP runtime error (no device, out of memory, launch failure) */
#define DST_ERR_AIR (-3)      /* transition constraints not satisfied by the trace (reference panic: evaluator.rs:155) */
#define DST_ERR_STATE (-4)    /* phase called out of order */
#define DST_ERR_COMM (-5)     /* a collective of dst_prove_sharded failed or did not complete within the communicator's limit (a peer that died or
                                 never arrived); the communicator has been aborted and refuses further collectives -- destroy it, create a new one */

typedef struct dst_ctx dst_ctx;

/* Shape of one proving job; mirrors TraceTable::new (src/stark/trace/trace_table.rs:23-58) + ProofOptions (options.rs:16-27). */
typedef struct dst_params {
    uint32_t log_trace_length;   /* n = 2^log_trace_length rows, >= 4 (TraceTable asserts a power of two; MIN_TRACE_LENGTH = 16) */
    uint32_t log_blowup;         /* extension_factor = 2^log_blowup, 16 <= factor <= 256 (options.rs:39-41) */
    uint32_t width;              /* number of registers W < 128 (lib.rs:83) */
    uint32_t ctx_depth;          /* context stack registers (<= 16) */
    uint32_t loop_depth;         /* loop stack registers (<= 8) */
    uint32_t num_queries;        /* 1..128 (options.rs:43-44) */
    uint32_t grinding_factor;    /* <= 32 (options.rs:46) */
    int32_t  device;             /* HIP device ordinal */
    /* multi-GPU sharding of the LDE domain by cosets (DESIGN.md "Multi-GPU"): this context owns cosets
       [rank * B / world, (rank + 1) * B / world); world = 1 means the whole job; world is a power of two <= min(8, B / 2). */
    uint32_t rank, world;
} dst_params;

/* Public inputs / outputs of the boundary constraints (src/stark/constraints/evaluator.rs:35-79). */
typedef struct dst_public {
    uint32_t num_inputs, num_outputs;       /* <= 8 each (lib.rs:136-137) */
    uint8_t inputs[8][16];
    uint8_t outputs[8][16];
} dst_public;

/* ---- lifetime ---------------------------------------------------------------------------------------------------- */
DST_API int dst_ctx_create(const dst_params* params, dst_ctx** out);
DST_API void dst_ctx_destroy(dst_ctx* ctx);
DST_API const char* dst_last_error(const dst_ctx* ctx);      /* ctx may be NULL: returns the creation-time error */
/* milliseconds spent in each of the reference's nine prover steps during the last proof (prover.rs:28,36,66,74,87,103,112,134,167) */
DST_API int dst_phase_ms(const dst_ctx* ctx, double out_ms[9]);

/* ---- step 0: trace upload (host -> HBM).  cols[c] points to n*16 bytes of register c, i.e. the reference's
 *      column-major `Vec<Vec<u128>>` (trace_table.rs:10).  Not part of the timed prover region. ------------------------ */
DST_API int dst_trace_upload(dst_ctx* ctx, const uint8_t* const* cols);
/* same, from one contiguous [W][n] buffer */
DST_API int dst_trace_upload_contiguous(dst_ctx* ctx, const uint8_t* cols);
/* Sharded contexts (world > 1): uploads only the registers this rank interpolates, r = rank (mod world); cols[r] of the other registers is
 * not read.  dst_prove_sharded all-gathers the coefficient vectors ("trace columns shard across the GPUs"), so every GPU
 * receives 1/world of the trace from its host.  Only dst_prove_sharded accepts a context in this state. */
DST_API int dst_trace_upload_owned(dst_ctx* ctx, const uint8_t* const* cols);
/* Asynchronous form for a host-resident trace (what stark::prove receives: prover.rs:17, trace_table.rs:10).  Starts the copies of
 * the W columns on a copy stream and returns at once; the next dst_commit_trace / dst_prove interpolates and extends the registers
 * group by group as their copies land, so that all but the first group's transfer overlaps with the extension.  The host buffers
 * must stay valid until that call returns; for real overlap they must be page-locked (dst_pinned_alloc or the host's own pinning). */
DST_API int dst_trace_upload_async(dst_ctx* ctx, const uint8_t* const* cols);
DST_API int dst_pinned_alloc(size_t bytes, void** out);
DST_API int dst_pinned_free(void* p);

/* ---- steps 1-2: TraceTable::extend + build_merkle_tree (prover.rs:22-35; trace_table.rs:143,174) ---------------------- */
DST_API int dst_commit_trace(dst_ctx* ctx, uint8_t trace_root[32]);

/* ---- steps 3-5: constraint evaluation, combination, constraint LDE + Merkle tree (prover.rs:43-86) ----------------------
 * coeffs = the 344 draws of ConstraintCoefficients::new(trace_root) in draw order (utils/coefficients.rs:66).
 * On DST_ERR_AIR *bad_step receives the first trace step whose transition constraints do not vanish.  The verdict of that check
 * (the reference panics at evaluator.rs:155) travels back with the constraint root: a failing trace is reported AFTER the combination,
 * the constraint LDE and its tree were queued behind the evaluation, so the context's constraint buffers (dst_read_buffer CPOLY, CEVALS,
 * CNODES) hold values computed from a non-vanishing evaluation then; the context stays in the "trace committed" state. */
DST_API int dst_eval_constraints(dst_ctx* ctx, const dst_public* pub, const uint8_t* coeffs /* 344*16 */, uint8_t constraint_root[32], int64_t* bad_step);

/* ---- step 6: DEEP composition (prover.rs:94-101, 189-201).  draws = the 516 draws of prng_vector(constraint_root):
 * draws[0] = z, then CompositionCoefficients (coefficients.rs:80-104).  Outputs the DeepValues (proof.rs:24-28). --------- */
DST_API int dst_compose(dst_ctx* ctx, const uint8_t* draws /* 516*16 */, uint8_t* trace_at_z1 /* W*16 */, uint8_t* trace_at_z2 /* W*16 */);

/* ---- step 7: FRI commit phase (fri/prover.rs:11-53).  Call dst_fri_commit_layer (root of the current layer), then
 * dst_fri_fold with special_x = prng(root); repeat while dst_fri_commit_layer reports more = 1.  The last committed
 * layer (<= 256 evaluations) is the remainder. -------------------------------------------------------------------------- */
DST_API int dst_fri_commit_layer(dst_ctx* ctx, uint8_t layer_root[32], int* more);
DST_API int dst_fri_fold(dst_ctx* ctx, const uint8_t special_x[16]);

/* ---- step 8: proof of work (utils/proof_of_work.rs:4-32): smallest nonce >= 1 whose digest has >= grinding_factor
 * trailing zero bits in its first little-endian u64. ------------------------------------------------------------------- */
DST_API int dst_pow_grind(dst_ctx* ctx, const uint8_t seed[32], uint32_t grinding_factor, uint8_t out_seed[32], uint64_t* nonce);

/* ---- step 9: openings (prover.rs:143-165).  Serialises the whole StarkProof in the reference's wire format
 * (bincode, src/main.rs:44) for the given query positions.  Two-call protocol: pass out = NULL to get the size. --------- */
DST_API int dst_build_proof(dst_ctx* ctx, const uint64_t* positions, uint32_t num_positions, uint64_t pow_nonce,
                    uint8_t* out, size_t cap, size_t* out_len);

/* ---- the whole of stark::prove (prover.rs:17-168) on a trace already uploaded with dst_trace_upload ------------------- */
DST_API int dst_prove(dst_ctx* ctx, const dst_public* pub, uint8_t* proof_out, size_t cap, size_t* proof_len);

/* ---- host-side helpers that the Rust host would otherwise take from `rand` (they run on the CPU) -------------------- */
DST_API void dst_prng_vector(const uint8_t seed[32], uint32_t count, uint8_t* out /* count*16 */);          /* field.rs:271 */
DST_API int dst_query_positions(const uint8_t seed[32], uint64_t domain_size, uint32_t blowup, uint32_t num_queries, uint64_t* out); /* utils/mod.rs:25 */
DST_API void dst_blake3(const uint8_t* in, size_t len, uint8_t out[32]);                                     /* crypto/hash.rs:205 (host) */

/* ---- benchmark inputs: the Fibonacci example trace (src/examples/fibonacci.rs:32-47) filling exactly 2^log_n rows
 * (W = 20, ctx_depth 1, loop_depth 0).  cols = [20][n] elements; outputs the program hash and the result. -------------- */
DST_API int dst_fibonacci_trace(uint32_t log_n, uint8_t* cols, uint8_t program_hash[32], uint8_t result[16]);

/* ---- coset-sharded proving on several GPUs (one context per GPU, params.rank / params.world) ----------
 * The phases mirror the single-GPU ones; collectives are issued by the host between them.  `what`: 0 trace-tree boundary
 * nodes, 1 constraint-tree boundary nodes, 2 FRI-tree boundary nodes of (sharded) layer `arg`, 3 combined constraint evaluations,
 * 4 last FRI layer (remainder: all of it, natural order, after the commit phase), 5 size query only: the largest item
 * dst_shard_fri_begin hands out.  dst_shard_import takes the all-gathered items of all ranks (rank-major); for trees it
 * finishes the replicated upper levels and returns the root.  *_is_device: the pointer is device memory (possibly owned
 * by another HIP runtime instance in the process, e.g. a torch tensor) instead of host memory. */
DST_API int dst_shard_commit_trace(dst_ctx* ctx);
DST_API int dst_shard_eval_constraints(dst_ctx* ctx, const dst_public* pub, const uint8_t* coeffs /* 344*16 */, int64_t* bad_step);
DST_API int dst_shard_combine(dst_ctx* ctx);
DST_API int dst_shard_fri_layer(dst_ctx* ctx, int* more);
DST_API int dst_shard_fri_fold(dst_ctx* ctx, const uint8_t special_x[16]);
DST_API int dst_shard_export_size(dst_ctx* ctx, uint32_t what, uint32_t arg, size_t* bytes);
DST_API int dst_shard_export(dst_ctx* ctx, uint32_t what, uint32_t arg, void* dst, int dst_is_device);
DST_API int dst_shard_import(dst_ctx* ctx, uint32_t what, uint32_t arg, const void* src, int src_is_device, uint8_t root_out[32]);
/* openings: `count` items by LOCAL index from buffer 0 trace leaves, 1 trace local nodes, 2 trace upper nodes, 3 constraint
 * evaluations (elements), 4 constraint local nodes, 5 constraint upper nodes, 6 FRI layer `arg` evaluations (elements), 7 FRI leaves,
 * 8 FRI local nodes, 9 FRI upper nodes, 10 LDE rows (idx = natural positions owned by this rank, W elements each), 11 transition
 * evaluations of the last dst_eval_constraints / dst_shard_eval_constraints (elements; this rank's cosets of the 8n-point domain,
 * coset-major: index q * n + k is the point 8k + q on one GPU).  The element buffers 3, 6 (sharded layers) and 11 are coset-major. */
DST_API int dst_shard_read(dst_ctx* ctx, uint32_t buffer, uint32_t arg, const uint64_t* idx, uint32_t count, uint8_t* out);
/* The FRI commit phase (fri/prover.rs:11-53): call dst_shard_fri_begin, all-gather the *bytes it wrote into `send` (same size
 * on every rank), call dst_shard_fri_end with the gathered bytes (rank-major); repeat while *more.  Large layers stay sharded:
 * begin = leaves + rank-local tree levels + export of the boundary nodes, end = upper tree, root, fold at field::prng(root).
 * From the first layer of at most 2^17 elements (or with fewer than one 4-element row per coset) begin hands out the rank's
 * cosets of the layer's EVALUATIONS instead (*more = 0) and end commits that layer and all following ones on every rank by
 * itself (replicated tail: no further exchanges).  dst_shard_fri_roots then returns the roots of all layers. */
DST_API int dst_shard_fri_begin(dst_ctx* ctx, void* send, int send_is_device, size_t cap, size_t* bytes, int* more);
DST_API int dst_shard_fri_end(dst_ctx* ctx, const void* gathered, int src_is_device, uint8_t root_out[32]);
DST_API int dst_shard_fri_roots(dst_ctx* ctx, uint8_t* roots /* 32 per layer, or NULL */, size_t cap, uint32_t* num_layers, uint32_t* replicated_from);

/* Step 9 across ranks (prover.rs:143-165; merkle.rs:64-124 prove_batch; fri/prover.rs:55-96 build_proof).  Every rank derives the
 * same ordered list of openings from the query positions.  dst_shard_open returns the items THIS rank owns, concatenated in that
 * order (blob == NULL: only the sizes; all_lens, if not NULL, receives every rank's blob length, `world` entries);
 * dst_shard_assemble takes the blobs of all ranks back to back (blob_lens[g] bytes each) and writes the serialised StarkProof
 * (proof.rs:11-77, bincode as src/main.rs:44) -- identical bytes on every rank. */
DST_API int dst_shard_open(dst_ctx* ctx, const uint64_t* positions, uint32_t num_positions, uint8_t* blob, size_t cap, size_t* blob_len, uint64_t* all_lens);
DST_API int dst_shard_assemble(dst_ctx* ctx, const uint64_t* positions, uint32_t num_positions, uint64_t pow_nonce, const uint8_t* blobs,
                       const uint64_t* blob_lens, uint8_t* out, size_t cap, size_t* out_len);
DST_API int dst_shard_info(dst_ctx* ctx, uint64_t* op_count, uint32_t* num_fri_layers, uint32_t* stack_depth);

/* ---- inspection (tests and profiling): copies an internal device buffer to the host.  `what` ids are listed in
 * distaff_amd/csrc/ctx.h (DST_BUF_*).  Two-call protocol: out = NULL returns the size through *len. ------------------- */
DST_API int dst_read_buffer(dst_ctx* ctx, uint32_t what, uint32_t arg, uint8_t* out, size_t cap, size_t* len);
/* 1: this is the test / bench build (libdistaff_hip_hooks.so: calibration kernels, the alternative formulations behind the test-only
 * DISTAFF_* switches of INTEGRATION.md section 6); 0: the product library, which has neither */
DST_API int dst_test_hooks(void);
/* micro-benchmark hook used by bench.py's roofline section: runs `iters` dependent modular multiplications per lane
 * on `lanes` lanes and returns the elapsed milliseconds.  (Test / bench build only: the product library returns DST_ERR_STATE.) */
DST_API int dst_bench_mulmod(dst_ctx* ctx, uint64_t lanes, uint32_t iters, double* ms);
/* peak of the 32x32+64 multiply-add (v_mad_u64_u32) on this device: `iters` iterations of 32 independent-enough mads per lane on
 * `lanes` lanes; returns the elapsed milliseconds (rate = lanes * iters * 32 / time).  The integer-multiplier roofline of the path.
 * (Test / bench build only: the product library returns DST_ERR_STATE.) */
DST_API int dst_bench_mad(dst_ctx* ctx, uint64_t lanes, uint32_t iters, double* ms);
/* the shader clock (MHz) the device sustains under this path's arithmetic: `iters` iterations of four dependent modular multiplications per lane
 * on `lanes` lanes, every wavefront timing itself with s_memtime against the constant 100 MHz s_memrealtime; the median.  The power management
 * does not hold the nominal clock under this load (profiles/r6_power_clock.md).  (Test / bench build only: the product library returns DST_ERR_STATE.) */
DST_API int dst_bench_clock(dst_ctx* ctx, uint64_t lanes, uint32_t iters, double* mhz);
/* box fingerprint: milliseconds for 2^23 lanes to run `code_kib` (16 or 176) KiB of straight-line multiply-adds once each.  The ratio
 * of the two times per instruction is 0.9 on a healthy device; a device on which code beyond the instruction cache is slow shows it here.
 * code_kib = 177: the 176 KiB kernel in its convoy form (256 lanes per workgroup, a workgroup barrier every 16 KiB: the wavefronts share
 * their instruction-cache lines) -- whether that form would help on the device at hand.
 * (Test / bench build only: the product library returns DST_ERR_STATE.) */
DST_API int dst_bench_code(dst_ctx* ctx, uint32_t code_kib, double* ms);
/* element-wise device field arithmetic on caller data (tests): op 0 add, 1 sub, 2 mul, 3 mul (portable formulation), 4 inv(a), 5 a^b */
DST_API int dst_field_op(dst_ctx* ctx, int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t count);
/* per-kernel timing with HIP events on the context's stream.  level 0: off; 1: every kernel launch is bracketed (costs ~5 % of a
 * proof: ~300 launches lose their back-to-back issue); 2: only the heavy kernels (NTT passes, constraint kernel, leaf hashing:
 * ~35 launches, ~90 % of the device time, no measurable cost).  dst_kernel_stats drains the events and writes a JSON object
 * {"kernel": {"launches": k, "ms": total, "bytes": algorithmic}, ...}. */
DST_API int dst_set_profiling(dst_ctx* ctx, int level);
DST_API int dst_kernel_stats(dst_ctx* ctx, char* json_out, size_t cap, int reset);

/* ---- one proof over several GPUs, collectives behind the C-ABI -------------------------------------------------------------------
 * The reference has no multi-device path (src/math/polynom.rs:36-37); the partitioning follows SURVEY.md 8(e): rank g of `world` owns
 * the cosets [g*B/world, (g+1)*B/world) of every extension, Merkle trees are finished from an all-to-all of boundary nodes (rank g
 * builds the subtree over the trace indices k in [g*n/world, (g+1)*n/world)), an all-gather of the `world` subtree roots and the top
 * log2(world) levels; constraint evaluations and the first small FRI layer are all-gathered.  No all-reduce anywhere.
 *
 * One communicator handle per rank.  RCCL transport (one process per GPU, xGMI): the rank-0 host calls dst_comm_unique_id and hands
 * the 128 bytes to the other ranks through its own channel; every rank then calls dst_comm_init with the device it created its
 * context on.  librccl.so is bound at run time, a single-GPU host never needs it.  In-process transport (dst_comm_init_local fills
 * `world` handles): the ranks are threads of one process, e.g. a host that drives all GPUs of a node itself, or tests; stream-ordered like
 * RCCL -- a collective is enqueued on the ranks' streams (events between them, device-to-device copies over peer access), the host threads
 * only hand over pointers and never wait for a stream (DISTAFF_LOCAL_TRANSPORT=blocking: the older form that drains the streams around
 * every collective).
 * dst_prove_sharded is stark::prove on context `ctx` (created with rank / world in dst_params, trace uploaded on EVERY rank): all ranks
 * call it, all ranks receive the same proof bytes; when any rank fails every rank returns an error.  dst_prove_sharded_local is the
 * one-call form for a single-process host: `world` contexts, one thread each. */
typedef struct dst_comm dst_comm;
DST_API int dst_comm_unique_id(uint8_t id[128]);
DST_API int dst_comm_init(const uint8_t id[128], uint32_t rank, uint32_t world, int device, dst_comm** out);
DST_API int dst_comm_init_local(uint32_t world, dst_comm** out /* [world] */);
/* the host's own transport (MPI, its process launcher's channel, ...): `fn` is called for every collective with kind 0 = all-gather of
 * `bytes` per rank, 1 = all-to-all with chunks of `bytes` (both on DEVICE buffers, complete when fn returns), 2 = all-gather of host values;
 * it returns 0 on success */
typedef int (*dst_comm_fn)(void* user, int kind, const void* send, void* recv, size_t bytes);
DST_API int dst_comm_init_callbacks(uint32_t rank, uint32_t world, dst_comm_fn fn, void* user, dst_comm** out);
DST_API void dst_comm_destroy(dst_comm* comm);
/* Containment.  The reference panics (src/lib.rs:32,49,56); dst_prove_sharded returns an error on every rank instead, also when a peer
 * never arrives: every host wait behind a collective is a bounded poll of the stream (plus ncclCommGetAsyncError on the RCCL transport).
 * After `seconds` without completion (default 60, DISTAFF_COMM_TIMEOUT_S at creation; <= 0: no limit) the rank aborts its communicator --
 * ncclCommAbort on RCCL, which also ends the kernels stuck on the stream; the in-process transport wakes every peer out of its barrier --
 * and returns DST_ERR_COMM; dst_comm_last_error then names the wait and the index, kind and size of the last collective this rank issued
 * (with dst_comm_trace the whole issue order).  dst_comm_abort does the same from the host's side (a watchdog that learnt of a dead
 * peer).  A callback transport's channel is the host's to bound: a callback that returns non-zero ends the rank the same way. */
DST_API int dst_comm_set_timeout(dst_comm* comm, double seconds);
DST_API int dst_comm_abort(dst_comm* comm);
/* What a communicator is and what its transport says about itself.  For the RCCL transport rccl_ranks / rccl_rank / device come from
 * ncclCommCount / ncclCommUserRank / ncclCommCuDevice on the live communicator (the proof that RCCL connected `world` ranks);
 * for the in-process transport peers_other_device = ranks whose buffers live on another device than this rank's and peers_enabled = how many
 * of those this rank reaches with peer access (hipDeviceEnablePeerAccess; the others are staged through the host), both known after the
 * first device collective. */
enum { DST_COMM_RCCL = 0, DST_COMM_LOCAL = 1, DST_COMM_CALLBACKS = 2 };
typedef struct dst_comm_info {
    uint32_t transport, rank, world;
    int32_t device;                     /* -1 when the transport does not know it */
    uint32_t rccl_ranks, rccl_rank, rccl_version;
    uint32_t peers_other_device, peers_enabled;
} dst_comm_info;
DST_API int dst_comm_describe(const dst_comm* comm, dst_comm_info* out);
/* Issue-order record of the collectives of this rank (see comm.h): enable = 1 start / 0 stop / -1 unchanged; `out` receives the record so
 * far as text ("<G|A|H> <bytes per rank> <stream index|->" per line), *len its full length.  DISTAFF_SHARD_DEBUG=1 starts it at creation. */
DST_API int dst_comm_trace(dst_comm* comm, int enable, char* out, size_t cap, size_t* len);
/* helper for callback transports that stage through their own buffers: one synchronous copy of `bytes` between host memory and memory of
 * the calling thread's current device, in either direction or device to device (the direction follows from the pointers) */
DST_API int dst_comm_copy(void* dst, const void* src, size_t bytes);
DST_API const char* dst_comm_last_error(const dst_comm* comm);   /* comm may be NULL: the creation-time error */
DST_API int dst_prove_sharded(dst_ctx* ctx, dst_comm* comm, const dst_public* pub, uint8_t* proof, size_t cap, size_t* len);
/* host-side view of the last dst_prove_sharded on this rank: out[0] = milliseconds inside the transport's calls (enqueue time on RCCL, the
 * whole exchange on a blocking transport), out[1] = milliseconds waiting for tree roots (the only host waits of the protocol), out[2] =
 * number of tree exchanges */
DST_API int dst_shard_stage_ms(const dst_ctx* ctx, double out[3]);
/* device-side view of the collectives of the last dst_prove_sharded on this rank: milliseconds from enqueue to completion (events around
 * every collective on the stream it was queued on; on RCCL that includes the wait for the slowest peer), summed per kind --
 * out[0] coefficient all-gathers, [1] tree all-to-alls (boundary nodes), [2] tree all-gathers (subtree roots + status records, or boundary
 * nodes in the all-gather-only form), [3] the all-gather of the constraint evaluations, [4] the all-gather of the first replicated FRI
 * layer, [5] all-gathers of host values (openings; host wall time), [6] number of collectives, [7] of which timed by events */
DST_API int dst_shard_exchange_ms(const dst_ctx* ctx, double out[8]);
DST_API int dst_prove_sharded_local(dst_ctx** ctxs, uint32_t world, const dst_public* pub, uint8_t* proof, size_t cap, size_t* len);

#ifdef __cplusplus
}
#endif
#endif
