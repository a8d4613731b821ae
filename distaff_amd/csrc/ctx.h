// Internal definitions shared by the kernel translation units and the C-ABI layer of libdistaff_hip.so.
//
// HBM layout (all field elements are 16-byte `fe`, digests 32-byte `digest`):
//   trace   [W][n]        register traces as uploaded (column-major, the reference's Vec<Vec<u128>>)
//   polys   [W][n]        the same registers in coefficient form (TraceTable.polys, trace_table.rs:11)
//   lde     [W][Bc][n]    "coset-major" low-degree extension: lde[c][j][k] = T_c(w_N^(B*k + j0 + j)), i.e. the value the
//                         reference keeps at registers[c][B*k + j0 + j].  A coset is a contiguous size-n array, so the
//                         size-n NTT that produces it, the constraint kernel (rows i and i+B are (j,k),(j,k+1)) and the
//                         FRI fold (i, i+N/4, ... share j) all stream with unit stride.
//   Merkle leaves / nodes are kept in the reference's natural leaf order (leaf i = B*k + j), heap-indexed
//   (nodes[1] = root, merkle.rs:269-294); kernels that read coset-major data and emit natural-order digests go
//   through an LDS tile transpose so that both sides are coalesced.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <memory>
#include <string>
#include <map>
#include <vector>
#include "fe.h"
#include "../../include/distaff_hip.h"

struct __attribute__((aligned(16))) digest { uint32_t w[8]; };
#define AIR_PERIODIC_STRIDE 29         // 8 sponge constants, 12 hasher constants, 3 cycle masks, the cubes of hasher constants 0..5

// Four-step twiddles are streamed from HBM once per element of every first pass: as plain elements (16 bytes, general multiplication,
// 71 VALU instructions) or as table pairs (32 bytes, fe_mul_tw, 55).  Measured on the 2^20 proof: first passes 13.1 ms with plain
// elements, 13.4 ms with pairs (the table is re-fetched per register: the four workgroups that share it drift apart by more than the
// 4 MiB L2 of their XCD holds), and half the table memory: plain elements are the default.
#ifndef NTT_TW4_PAIRS
#define NTT_TW4_PAIRS 0
#endif
#if NTT_TW4_PAIRS
typedef fe_tw tw4_t;
#else
typedef fe tw4_t;
#endif

#define DST_MAX_FRI_LAYERS 24

// Every DISTAFF_* variable the library looks at, in one place (INTEGRATION.md section 6 prints this table; a test keeps both in step).
// They are read ONCE per context, when it is created (dst_ctx_create -> dst_ctx::sw), never per call.  `product` switches are operational
// choices every build honours; the others select alternative formulations that exist for the tests to compare with the default ones and are
// honoured only by the test / bench build (libdistaff_hip_hooks.so, -DDISTAFF_TEST_HOOKS): the product library does not even read them.
// (DISTAFF_SHARD_DEBUG is also looked at by dst_comm_init*, which has no context: once, when the communicator is created; DISTAFF_COMM_TIMEOUT_S
// DISTAFF_LOCAL_TRANSPORT and the test build's DISTAFF_TEST_STALL_COLLECTIVE are communicator settings read at the same moment and nowhere else.)
struct dst_switch_def { const char* name; bool product; const char* values; const char* what; };
static const dst_switch_def DST_SWITCHES[] = {
    {"DISTAFF_SHARD_DEBUG",        true,  "1",               "stderr line per tree exchange on rank 0; communicators record the order of their collectives (dst_comm_trace)"},
    {"DISTAFF_SHARD_TREE_GATHER",  true,  "1",               "sharded Merkle trees: all-gather of all boundary nodes + upper levels repeated on every rank (BASELINE north_star's all-gather-only form) instead of the k-range all-to-all"},
    {"DISTAFF_SHARD_NO_OVERLAP",   true,  "1",               "sharded prover: every collective on the context's main stream (no second stream / events) also on a stream-ordered transport"},
    {"DISTAFF_COMM_TIMEOUT_S",     true,  "seconds",         "communicators: limit of every host wait behind a collective (default 60; <= 0: none); on expiry the rank aborts the communicator and returns DST_ERR_COMM (dst_comm_set_timeout changes it per handle)"},
    {"DISTAFF_LOCAL_TRANSPORT",    true,  "blocking",        "in-process transport (dst_comm_init_local, dst_prove_sharded_local): drain every rank's stream around each collective instead of the stream-ordered form (events between the ranks' streams, nothing waited for)"},
    {"DISTAFF_TMP_REGS",           true,  "4..W",            "registers per transform launch = size of the staging array (default: as many as 12 GiB hold, at most W)"},
    {"DISTAFF_AIR",                false, "small|deep|generic", "force a more general constraint-kernel instance set than the trace shape needs (generic = per-operation formulation)"},
    {"DISTAFF_BOUNDARY",           false, "eval",            "boundary combinations by evaluation on the 8n domain (the reference's route) instead of coefficient form"},
    {"DISTAFF_COMBINE",            false, "steps",           "combine_polys / DEEP composition as the reference's sequence of whole-array steps instead of the fused passes"},
    {"DISTAFF_NTT",                false, "pre|3pass|reg|lds", "force a transform plan family"},
    {"DISTAFF_NTT_SHAPE",          false, "a,b",             "three-pass plans: log2 of the first two pass lengths"},
    {"DISTAFF_NTT_ORDER",          false, "0",               "first pass of an extension in coset-slow block order"},
    {"DISTAFF_NTT_WAVES",          false, "4|8",             "force the 512- or 1024-lane instances of the LDS passes"},
    {"DISTAFF_NTT_FIXED",          false, "0",               "use the any-shape instances where a shape-compiled one exists"},
    {"DISTAFF_NTT_DIF",            false, "0|1|2",           "first pass of an extension: coset DIT (0), pre-scale + DIF (1), DIT with last-stage twiddles from global memory (2)"},
    {"DISTAFF_NTT_DEBUG",          false, "1",               "print the occupancy of every transform launch shape once"},
    {"DISTAFF_LDE_BATCH",          false, "cols,cosets",     "registers x cosets per transform launch"},
    {"DISTAFF_FOLD8_DFT",          false, "0",               "8n-coefficient extensions without the fused fold + 8-point step"},
    {"DISTAFF_TRACE_BUFFER",       false, "1",               "give the trace its own buffer instead of coset 0 of the extension"},
    {"DISTAFF_MERKLE_LEVEL2_LOG",  false, "k",               "build two tree levels per launch from 2^k nodes on (default 2^15)"},
    {"DISTAFF_MERKLE_LEVELS",      false, "1",               "one launch per tree level"},
    {"DISTAFF_SYN_DIV_TABLES",     false, "1",               "synthetic division by power tables + scan instead of the blocked form"},
    {"DISTAFF_FRI_TAIL",           false, "0|k",             "0: no single-launch tail; k: tail from layers of 2^k elements on"},
    {"DISTAFF_FRI_CHAIN",          false, "0",               "FRI commit phase with a root read-back and a host draw per layer"},
    {"DISTAFF_FRI_REPLICATE_LOG",  false, "k",               "sharded prover: replicate FRI layers from 2^k elements on (default 2^17)"},
    {"DISTAFF_SHARD_FORCE_OVERLAP", false, "1",              "sharded prover: the two-stream choreography of a stream-ordered transport over a blocking one"},
    {"DISTAFF_TEST_STALL_COLLECTIVE", false, "k|k@r",        "communicators: fault injection -- before device collective number k (of rank r only) a kernel holds the stream until the communicator is aborted"},
};
// Layout of the page-locked staging area dst_ctx::h_stage (HS_TOTAL bytes): every small upload / read-back that must be QUEUED rather than
// waited for has its own region, so that the lifetime of the host side of an asynchronous copy is explicit (a pageable source would be
// staged by the runtime -- which makes the host wait for the stream -- or go out of scope).  Regions are reused per proof: the host
// synchronises the stream at least once between two uses of the same region.
enum : size_t {
    HS_WEIGHTS = 0,                    // boundary weights: 4 W + 4 elements                                   (api.hip dst_internal_boundary_polys)
    HS_WEIGHTS_BYTES = (4 * 128 + 4) * 16,
    HS_DEEP = 32768,                   // read-back of T(z) at +0 and T(z g) at +HS_DEEP_HALF, W elements each   (api.hip compose_impl)
    HS_DEEP_HALF = 2048,
    HS_FRI_ROOTS = 40960,              // read-back of the FRI roots: 32 bytes per layer                          (api.hip / shard.hip FRI commit)
    HS_FRI_SLOTS = 45056,              // sharded FRI layers: deferred tree records, HS_FRI_SLOT_BYTES per layer  (shard.hip dst_prove_sharded)
    HS_FRI_SLOT_BYTES = 800,           //   G records of 96 bytes + the 32-byte root, G <= 8
    HS_AIR_FLAG = 65536 - 64,          // read-back of the failing step of the constraint check                   (kernels_air.hip)
    HS_DRAWS = 65536,                  // upload of the 344 constraint coefficients + the compacted transition coefficients (<= 156)
    HS_DRAWS_BYTES = (344 + 160) * 16,
    HS_COMPOSE = 65536 + 8192,         // upload of the 516 composition draws
    HS_COMPOSE_BYTES = 516 * 16,
    HS_STATUS = 65536 + 8192 + 8320,   // upload of this rank's status record of a tree exchange: a ring of HS_STATUS_SLOTS records of 64 bytes
    HS_STATUS_SLOTS = 32,
    // read-backs that sit BEHIND collectives of the sharded prover: into page-locked memory, because a device-to-host copy into pageable
    // memory blocks the host inside hipMemcpyAsync -- an unbounded wait in front of the bounded one (comm.h wait_stream)
    HS_TREE_READBACK = 65536 + 8192 + 8320 + 2048,     // tree exchange: G records of 96 bytes + the 32-byte root               (shard.hip tree_exchange)
    HS_TAIL_ROOTS = HS_TREE_READBACK + 1024,           // roots of the FRI layers committed in one launch, 32 bytes per layer     (kernels_poly.hip k_fri_tail)
    HS_POW = HS_TAIL_ROOTS + 1024,                     // the nonce found by the proof-of-work search                            (kernels_poly.hip k_pow)
    HS_TOTAL = 131072
};
static_assert(HS_WEIGHTS + HS_WEIGHTS_BYTES <= HS_DEEP, "boundary weights overlap the DEEP values");
static_assert(HS_DEEP_HALF >= 128 * 16 && HS_DEEP + 2 * HS_DEEP_HALF <= HS_FRI_ROOTS, "DEEP values: W <= 128 registers");
static_assert(HS_FRI_ROOTS + DST_MAX_FRI_LAYERS * 32 <= HS_FRI_SLOTS, "FRI roots overlap the layer slots");
static_assert(8 * 96 + 32 <= HS_FRI_SLOT_BYTES && HS_FRI_SLOTS + DST_MAX_FRI_LAYERS * HS_FRI_SLOT_BYTES <= HS_AIR_FLAG, "FRI layer slots: G <= 8 ranks, DST_MAX_FRI_LAYERS layers");
static_assert(HS_DRAWS + HS_DRAWS_BYTES <= HS_COMPOSE && HS_COMPOSE + HS_COMPOSE_BYTES <= HS_STATUS && HS_STATUS + HS_STATUS_SLOTS * 64 <= HS_TOTAL, "upload regions overlap");
static_assert(HS_STATUS_SLOTS >= 2 + DST_MAX_FRI_LAYERS, "one status slot per tree exchange of a proof");
static_assert(HS_STATUS + HS_STATUS_SLOTS * 64 <= HS_TREE_READBACK && 8 * 96 + 32 <= 1024 && DST_MAX_FRI_LAYERS * 32 <= 1024 && HS_POW + 64 <= HS_TOTAL, "read-back regions overlap");

#ifdef DISTAFF_TEST_HOOKS
#define DST_TEST_HOOKS 1
#else
#define DST_TEST_HOOKS 0
#endif

// ids for dst_read_buffer
enum {
    DST_BUF_POLYS = 0,        // [W][n]
    DST_BUF_LDE = 1,          // arg = register: natural order [N] (converted from coset-major on the device)
    DST_BUF_TRACE_LEAVES = 2, // [N] x 32
    DST_BUF_TRACE_NODES = 3,  // [N] x 32 (heap)
    DST_BUF_CEVAL_I = 4,      // combined boundary (first step) evaluations over the 8n domain, natural order
    DST_BUF_CEVAL_F = 5,
    DST_BUF_CEVAL_T = 6,
    DST_BUF_CPOLY = 7,        // constraint polynomial, 8n coefficients
    DST_BUF_CEVALS = 8,       // constraint polynomial over the LDE domain, natural order [N]
    DST_BUF_CNODES = 9,       // constraint tree nodes [N/2] x 32 (heap)
    DST_BUF_COMP_POLY = 10,   // composition polynomial, 8n coefficients
    DST_BUF_COMP_EVALS = 11,  // composition evaluations, natural order [N]
    DST_BUF_FRI_EVALS = 12,   // arg = layer: evaluations of that layer, natural order
    DST_BUF_FRI_NODES = 13,   // arg = layer: tree nodes (heap)
    DST_BUF_FRI_LEAVES = 14,  // arg = layer: hashed rows
};

struct NttPlan {
    uint32_t log_n = 0, log_n1 = 0, log_n2 = 0;      // n = n1 * n2: first pass length, rest
    uint32_t log_n3 = 0;                             // 0: two passes (second pass length n2); else three passes n = n1 * (n2/n3) * n3
    uint32_t tile_a = 1, tile_b = 1;                 // columns per workgroup tile in pass A / pass B
    uint32_t tile_m = 1;                             // three-pass plans: columns per tile of the middle pass
    bool reg_a = false, reg_b = false;               // per pass: register-radix kernel (tile lengths 2^6 .. 2^12) instead of the LDS radix-2 one
    uint32_t pre_a = 0, pre_b = 0;                   // two-pass plans: register pre-stage of the pass (its LDS tiles hold 2^(log_n1 - pre_a) / 2^(log_n2 - pre_b) points), see NttArgs
};

struct dst_ctx {
    dst_params prm{};
    std::string err;
    // the DISTAFF_* switches as they were when the context was created (see DST_SWITCHES); sw() returns nullptr for an unset one
    std::map<std::string, std::string> sw_values;
    const char* sw(const char* name) const { auto it = sw_values.find(name); return it == sw_values.end() ? nullptr : it->second.c_str(); }
    bool sw_is(const char* name, const char* value) const { const char* v = sw(name); return v && !strcmp(v, value); }
    bool sw_flag(const char* name) const { const char* v = sw(name); return v && v[0] && v[0] != '0'; }
    void read_switches() {
        sw_values.clear();
        for (const dst_switch_def& d : DST_SWITCHES)
            if (d.product || DST_TEST_HOOKS) if (const char* v = getenv(d.name)) sw_values[d.name] = v;
    }
    int device = 0;
    hipStream_t stream = nullptr;

    // derived sizes
    uint32_t log_n = 0, log_b = 0, log_N = 0;
    size_t n = 0, B = 0, N = 0, W = 0;
    size_t Bc = 0, j0 = 0;              // local cosets
    size_t stack_depth = 0;
    NttPlan plan;
    std::vector<fe> shard_draws;       // shard.hip: the 344 constraint coefficients of the current proof
    std::shared_ptr<void> open_plan;   // shard.hip: plan of the last dst_shard_open, reused by dst_shard_assemble for the same positions
    std::vector<uint64_t> open_plan_positions;
    uint8_t open_plan_root[32] = {0};   // trace root the cached plan was built for
    int fri_rep_from = 0;               // sharded phases: FRI layers >= this one are replicated (natural order, full heaps on every rank), see shard.hip
    bool fri_tail_pending = false;      // dst_shard_fri_begin exported the evaluations of layer fri_rep_from; dst_shard_fri_end finishes the commit phase
    fe* fri_nat0 = nullptr;             // natural-order layer 0 when fri_rep_from == 0 (fri_e[0] is the rank's coset-major piece)
    bool sharded_layout = false;        // FRI layers >= 1 coset-major with per-rank tree heaps (dst_shard_* phases) instead of natural order / full heaps

    // tables (device)
    fe *tw_lo = nullptr, *tw_hi = nullptr;       // w_N^t, two-level: e = (hi << lo_bits) | lo
    fe *itw_lo = nullptr, *itw_hi = nullptr;     // w_N^-t
    uint32_t tw_lo_bits = 0;
    // every twiddle of the LDS-family transforms is a table pair (w, w * 2^64 mod p), see fe_mul_tw (fe.h)
    fe_tw *w1f = nullptr, *w2f = nullptr, *w1i = nullptr, *w2i = nullptr;   // stage twiddles w_{n1}^t, w_{n2}^t and inverses (of the LDS transform lengths)
    fe_tw *w1pf = nullptr, *w1pi = nullptr, *w2pf = nullptr, *w2pi = nullptr;   // register pre-stages: w_{n1}^m, m < n1 / 2 (and inverse), the same for n2
    fe_tw *prescale = nullptr;                   // w_{B*n1}^t, t < B*n1
    fe_tw *dit_last = nullptr;                   // [B][R][len/2], len = n1 / R, R = 2^pre_a: last-stage twiddles of every coset's (half-length) DIT, w_{B*n1}^(j + B*(h + R*k)) (the pre-scale table regrouped)
    // four-step twiddles of pass A as full tables in output order [k1][m2] (one multiplication per element instead of a two-level
    // lookup + two; the extra 16 B/element read is free: the pass runs at a tenth of the HBM bandwidth)
    tw4_t *tw4_lde = nullptr;                    // [Bc][n]: w_N^(m2 * (B*k1 + j)), local cosets j
    tw4_t *tw4_fwd = nullptr, *tw4_inv = nullptr;   // [n]: w_n^(m2*k1) and its inverse
    tw4_t *tw4_row_fwd = nullptr, *tw4_row_inv = nullptr;   // three-pass plans: [n2] twiddles w_{n2}^(k2*m3) of the middle pass and inverse
    fe_tw *w3f = nullptr, *w3i = nullptr;        // three-pass plans: stage twiddles of the last pass (length n3)
    fe *tmp2 = nullptr;                          // three-pass plans: second staging buffer
    fe *periodic = nullptr;                      // [128][AIR_PERIODIC_STRIDE] extended Rescue round constants + cycle masks + cubes of six of them
    void *air_consts = nullptr;                  // AirConsts (Rescue MDS matrices) in device memory
    fe c16f[8], c16i[8];                         // w_16^j and w_16^-j, j < 8 (passed to the NTT kernels by value)
    fe_tw n_inv_tw{};                            // 1/n as a table pair
    fe n_inv{}, eight_inv{}, four_inv{}, iota{}, g_trace{}, x_last{};   // 1/n, 1/8, 1/4, w_N^(N/4), w_n, w_n^(n-1)

    // data (device)
    fe *trace = nullptr, *polys = nullptr, *lde = nullptr, *tmp = nullptr;
    size_t tmp_regs = 4;                // tmp (and tmp2) hold tmp_regs x Bc arrays of n elements
    // register r of the trace starts at trace + r * trace_stride.  A context that owns coset 0 of the extension keeps the trace IN the
    // coset-0 slots of `lde` (trace == lde, trace_stride == Bc * n): coset 0 of the extension is the trace itself, nothing is copied
    size_t trace_stride = 0;
    digest *trace_leaves = nullptr, *trace_nodes = nullptr;
    fe *ceval = nullptr;                // [3][8c][n] combined constraint evaluations (i, f, t), coset-major over the 8n domain
    fe *cwork = nullptr;                // scratch, 3 * 8n
    fe *cpoly = nullptr;                // [8n]
    fe *cevals = nullptr;               // [Bc][n]
    digest *cnodes = nullptr;           // [N/2]
    fe *comp_poly = nullptr;            // [8n]
    fe *comp = nullptr;                 // [Bc][n]
    fe *scratch = nullptr;              // misc scan / reduction scratch
    size_t scratch_elems = 0;
    int num_fri_layers = 0, fri_committed = 0, fri_folded = 0;
    fe *fri_e[DST_MAX_FRI_LAYERS] = {nullptr};       // evaluations of layer d (d = 0 aliases comp, coset-major)
    digest *fri_leaves[DST_MAX_FRI_LAYERS] = {nullptr};
    digest *fri_nodes[DST_MAX_FRI_LAYERS] = {nullptr};
    size_t fri_size[DST_MAX_FRI_LAYERS] = {0};       // N_d
    // coset-sharded mode (world > 1): replicated upper parts of the trees and the gather landing buffer
    digest *trace_upper = nullptr, *c_upper = nullptr, *fri_upper[DST_MAX_FRI_LAYERS] = {nullptr};
    uint8_t *gather_buf = nullptr; size_t gather_bytes = 0;
    // dst_prove_sharded: trees whose upper part is partitioned by k-ranges (rank g finishes the subtree over k in [g*K/G, (g+1)*K/G) from
    // an all-to-all of boundary nodes; `*_upper` then holds the replicated top heap [0, 2G) and this rank's subtree heap behind it)
    // instead of being rebuilt from an all-gather on every rank (the host-orchestrated dst_shard_import).  Index 0 trace, 1 constraint, 2 + d FRI layer d.
    bool tree_krange[2 + DST_MAX_FRI_LAYERS] = {false};
    uint64_t *d_u64 = nullptr;                        // small device scalars (pow result, AIR failure flag)
    uint8_t *d_fri_chain = nullptr;                   // dst_prove's FRI commit phase without host round trips: [24] roots (32 B) then [24] draws (16 B)
    uint8_t *h_stage = nullptr;                       // page-locked host staging (HS_TOTAL bytes, layout HS_* above): small uploads / read-backs that must not make the host wait
    unsigned long long air_flag_host = ~0ull;         // read-back of the AIR failure flag (deferred check)
    hipEvent_t ph_ev[6] = {nullptr};                  // phase boundaries on the stream (phase times without host waits)
    uint8_t *d_stage = nullptr;                       // staging buffer for gathers
    size_t stage_bytes = 0;

    // host copies kept for proof assembly
    dst_public pub{};
    uint8_t trace_root[32] = {0}, constraint_root[32] = {0};
    std::vector<uint8_t> deep_z1, deep_z2;
    bool deep_pending = false;          // the DEEP values of the last composition are still in the page-locked staging area
    std::vector<std::vector<uint8_t>> fri_roots;
    uint64_t op_count = 0;
    fe program_hash[2] = {};
    bool have_trace = false, committed = false, constraints_done = false, composed = false;
    // asynchronous upload (dst_trace_upload_async): column group g is complete when upload_done[g] has fired on upload_stream
    hipStream_t upload_stream = nullptr;
    std::vector<hipEvent_t> upload_done;
    std::vector<size_t> upload_bounds;  // group g holds registers [upload_bounds[g], upload_bounds[g + 1])
    bool upload_pending = false;
    // dst_trace_upload_owned: only the registers r = rank (mod world) of the trace are on this device (the sharded prover interpolates
    // exactly those and all-gathers the coefficient vectors)
    bool trace_owned_only = false;
    // dst_prove_sharded: the gathered transition evaluations in `ceval` are already inverse-transformed per coset (every rank did its own
    // cosets before the exchange); dst_shard_combine then only runs the 8-point step across cosets
    bool ceval_inverted = false;
    // dst_prove_sharded: collectives that overlap with compute run on their own stream, ordered by events; status records of all ranks
    hipStream_t comm_stream = nullptr;
    std::vector<hipEvent_t> comm_events;
    uint8_t* d_status = nullptr;
    double phase_ms[9] = {0};
    double shard_ms[2] = {0, 0};           // dst_prove_sharded: host milliseconds inside the transport's calls / waiting for tree roots
    uint32_t shard_trees = 0;              // tree exchanges of the last sharded proof
    // dst_prove_sharded: the communicator whose collectives may sit on this context's streams.  While it is set, every host wait of the
    // prover is the communicator's bounded poll (ctx_sync below, comm.h) instead of hipStreamSynchronize
    struct dst_comm* wait_comm = nullptr;
    hipEvent_t sh_ev[8] = {nullptr};       // phase boundaries of dst_prove_sharded on the stream (created on first use)
    struct CollEv { hipEvent_t e0, e1; int kind; };
    std::vector<CollEv> coll_ev;           // events around the collectives of the last sharded proof (the first coll_used of them)
    size_t coll_used = 0;
    double exchange_ms[8] = {0};           // dst_shard_exchange_ms

    // optional per-kernel timing with HIP events recorded on `stream` (dst_set_profiling / dst_kernel_stats)
    int profile = 0;                       // dst_set_profiling: 0 off, 1 every kernel launch, 2 only the heavy kernels (NTT passes, constraint kernel, leaf hashing)
    struct KEvent { hipEvent_t e0, e1; std::string name; double bytes, mads; };
    std::vector<KEvent> kpending;
    std::vector<hipEvent_t> event_pool;    // recycled profiling events
    struct KStat { uint64_t launches = 0; double ms = 0, bytes = 0, mads = 0; };
    std::map<std::string, KStat> kstats;
};

// brackets one kernel launch with events when profiling is on; `bytes` = algorithmic HBM bytes of that launch, `mads` = 32x32+64
// multiply-adds (v_mad_u64_u32) the launch executes per the kernel's arithmetic (0 where not counted)
struct KScope {
    dst_ctx* c; bool on;
    dst_ctx::KEvent ev;
    KScope(dst_ctx* ctx, const char* name, double bytes, bool heavy = false, double mads = 0.0) : c(ctx), on(ctx->profile == 1 || (ctx->profile == 2 && heavy)) {
        if (!on) return;
        ev.name = name; ev.bytes = bytes; ev.mads = mads;
        // events come from a pool that dst_kernel_stats refills: creating two per launch would be host time inside the timed region
        if (c->event_pool.size() >= 2) { ev.e0 = c->event_pool.back(); c->event_pool.pop_back(); ev.e1 = c->event_pool.back(); c->event_pool.pop_back(); }
        else if (hipEventCreate(&ev.e0) != hipSuccess || hipEventCreate(&ev.e1) != hipSuccess) { on = false; return; }
        (void)hipEventRecord(ev.e0, c->stream);
    }
    ~KScope() { if (on) { (void)hipEventRecord(ev.e1, c->stream); c->kpending.push_back(ev); } }
};

#define HIP_TRY(ctx, expr)                                                                          \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) {                                                                     \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                          \
            return DST_ERR_HIP;                                                                     \
        }                                                                                           \
    } while (0)

// host wait for the context's stream: bounded while a communicator is attached (comm.hip), hipStreamSynchronize otherwise
int ctx_sync(dst_ctx* c, const char* what);
#define CTX_SYNC(ctx, what) do { const int _r = ctx_sync((ctx), (what)); if (_r) return _r; } while (0)

// ---- kernel launchers (kernels_*.hip) ------------------------------------------------------------------------------------
// NTT / LDE
extern "C" bool dst_internal_boundary_by_evaluation(const dst_ctx* c);                             // api.hip: DISTAFF_BOUNDARY=eval (test build only)
extern "C" int dst_internal_boundary_polys(dst_ctx* c, const fe* draws344, fe* ip, fe* fp, fe* o0 = nullptr, fe* o1 = nullptr, fe* o2 = nullptr, fe* o3 = nullptr);   // api.hip: boundary combinations in coefficient form (ip / fp: 8n coefficients each; or the four n-coefficient pieces)
extern "C" bool dst_internal_combine_by_steps(const dst_ctx* c);                                   // api.hip: DISTAFF_COMBINE=steps (the reference's sequence of whole-array steps; test build only)
extern "C" int dst_internal_boundary_quotients(dst_ctx* c, const fe* draws344, fe* q4, size_t stride);   // api.hip: the n-coefficient quotients the fused combination reads
int k_build_twiddle_tables(dst_ctx* c);                                                 // fills tw4_lde / tw4_fwd / tw4_inv (context creation)
void k_intt_columns(dst_ctx* c, const fe* src, size_t src_stride, fe* dst, size_t ncols); // size-n inverse NTT of ncols columns src_stride apart -> contiguous columns
void k_lde_columns(dst_ctx* c, const fe* polys, fe* lde, size_t ncols);                 // n coefficients -> coset-major [Bc][n] per column
void k_lde_fold8(dst_ctx* c, const fe* poly8n, fe* out);                                // 8n coefficients -> coset-major [Bc][n]
void k_intt8_cosets(dst_ctx* c, fe* vals /* [8][n] coset-major, in place scratch */, fe* out8n, fe* work);
void k_coset_to_natural(dst_ctx* c, const fe* src, size_t cosets, fe* dst);             // [cosets][n] -> natural [n*cosets]
void k_coset_to_natural_len(dst_ctx* c, const fe* src, size_t cosets, size_t len, fe* dst);
void k_fri_leaves_at(dst_ctx* c, const fe* e, digest* leaves, size_t R);
void k_fri_fold_at(dst_ctx* c, const fe* e, fe* out, size_t R, int layer, fe special_x, const fe* alpha_dev = nullptr);   // alpha_dev: x read from device memory instead
// hashing
void k_trace_leaves(dst_ctx* c);
void k_merkle_levels(dst_ctx* c, const digest* leaves, digest* nodes, size_t num_leaves);
void k_constraint_tree(dst_ctx* c);
void k_fri_leaves_layer0(dst_ctx* c);
void k_fri_leaves(dst_ctx* c, int layer);
// AIR
int k_eval_constraints(dst_ctx* c, const fe* coeffs_dev, const fe* tc_dev, int64_t* bad_step, bool defer_check = false);   // defer_check: no host wait; k_constraint_check reads the flag later
int k_constraint_check(dst_ctx* c, int64_t* bad_step);                                  // after a stream synchronisation: the failing step recorded by the last evaluation, if any
// polynomial helpers
void k_syn_div(dst_ctx* c, fe* a, size_t len, fe b);                                    // polynom.rs:190 semantics, in place
void k_syn_div_batch(dst_ctx* c, fe* const* a, const fe* b, int count, size_t len);                 // a[k] <- a[k] / (x - b[k]) in place, count <= 4 arrays of `len` coefficients, one set of launches
void k_syn_div_compose(dst_ctx* c, const fe* a, fe* out, size_t len, fe b, const fe* t, size_t tn, size_t inc, fe k1, fe k2, fe k3);   // out = k3 * a / (x - b) + (k1 + k2 x^inc) * t
void k_combine_fused(dst_ctx* c, const fe* work, const fe* q4, size_t q_stride, fe* cpoly);   // kernels_ntt.hip: 8-point step + division + boundary quotients -> constraint polynomial
void k_syn_div_expanded(dst_ctx* c, const fe* a, fe* out, size_t len, size_t degree, fe exception);
void k_horner(dst_ctx* c, const fe* polys, size_t ncols, size_t len, fe x, fe* out_dev);
void k_lincomb(dst_ctx* c, const fe* cols, size_t ncols, size_t len, const fe* coeffs_dev, fe* out);
void k_lincomb2(dst_ctx* c, const fe* cols, size_t ncols, size_t len, const fe* coeffs_dev, size_t coef_stride, fe* out0, fe* out1);
void k_lincomb4(dst_ctx* c, const fe* cols, size_t ncols, size_t len, const fe* coeffs_dev, fe* out0, fe* out1, fe* out2, fe* out3);   // four combinations, one pass over the columns
void k_axpy(dst_ctx* c, fe* y, const fe* x, fe a, size_t len);                           // y += a * x
void k_add(dst_ctx* c, fe* y, const fe* x, size_t len);                                  // y += x
void k_sub_dot_at0(dst_ctx* c, fe* y, const fe* values_dev, const fe* coeffs_dev, size_t count);   // y[0] -= sum values[k]*coeffs[k]
void k_sub_at0(dst_ctx* c, fe* y, const fe* v_dev);                                                   // y[0] -= v[0]
// FRI
void k_fri_fold(dst_ctx* c, int layer, fe special_x);
void k_fri_fold_dev(dst_ctx* c, int layer, const fe* alpha_dev);                        // the same with x read from device memory
void k_fri_draw(dst_ctx* c, int layer, fe* alpha_out, digest* root_out);                // x = prng(root of `layer`) on the device; files the root
void k_intt_cosets_local(dst_ctx* c, fe* vals, fe* out, size_t cosets);
void k_cross8(dst_ctx* c, const fe* work, fe* out8n);
int k_fri_tail(dst_ctx* c, int first, uint8_t* roots_out);       // natural-order layers first .. last in one launch (kernels_poly.hip)
// PoW
int k_pow(dst_ctx* c, const uint8_t seed[32], uint32_t grinding, uint64_t* nonce);
// gathers for openings
void k_gather(dst_ctx* c, const void* src, size_t item_bytes, const uint64_t* idx_dev, size_t count, void* dst);
void k_gather_rows(dst_ctx* c, const uint64_t* positions_dev, size_t count, fe* out);
void k_gather_pieces(dst_ctx* c, const uint64_t* addr_dev, size_t count, void* dst);          // dst[t] = the 16 bytes at device address addr[t]
int k_bench_mulmod(dst_ctx* c, uint64_t lanes, uint32_t iters, double* ms);
int k_bench_mad(dst_ctx* c, uint64_t lanes, uint32_t iters, double* ms);
int k_bench_clock(dst_ctx* c, uint64_t lanes, uint32_t iters, double* mhz);          // shader clock (MHz) sustained under four fe_mul chains per lane on `lanes` lanes
int k_bench_code(dst_ctx* c, uint32_t code_kib, double* ms);      // kernels_probe.hip
// coset-sharded (multi-GPU) helpers
void k_merkle_local_levels(dst_ctx* c, digest* nodes, size_t count, size_t stop_count);
void k_merkle_levels_to(dst_ctx* c, const digest* leaves, digest* nodes, size_t num_leaves, size_t stop_count);
void k_upper_tree(dst_ctx* c, const digest* gathered, digest* upper, size_t nb, uint32_t G);
void k_merkle_upper(dst_ctx* c, digest* nodes, size_t count);       // nodes[1 .. count) from the filled level nodes[count .. 2*count)
void k_digests_from_records(dst_ctx* c, const void* recs, size_t stride, digest* dst, size_t count);   // dst[i] = first 32 bytes of record i (count <= 8)
void k_constraint_level1(dst_ctx* c);
void k_fri_leaves_cm(dst_ctx* c, const fe* e, digest* leaves, size_t nd);
void k_fri_fold_cm(dst_ctx* c, const fe* e, fe* out, size_t nd, int layer, fe special_x, const fe* alpha_dev = nullptr);
void k_fri_draw_at(dst_ctx* c, const digest* nodes, fe* alpha_out, digest* root_out);    // x = prng(nodes[1]) on the device
void k_copy(dst_ctx* c, void* dst, const void* src, size_t bytes);
int k_field_op(dst_ctx* c, int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t count);
