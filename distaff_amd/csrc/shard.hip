// Coset-sharded (multi-GPU) prover phases of libdistaff_hip.so -- one context per GPU, rank g of G owns the cosets
// [g*B/G, (g+1)*B/G) of every LDE.  The phases mirror stark::prove (/root/reference/src/stark/prover.rs:17-168)
// exactly like the single-GPU entry points of api.hip; what differs is that Merkle trees are finished from all-gathered
// boundary nodes and that the constraint evaluations are all-gathered before the cross-coset inverse transform.  The
// collectives themselves are issued by the host (torch.distributed / RCCL in distaff_amd/sharded.py): this file only
// exports and imports the shards.
#include <chrono>
#include <functional>
#include <set>
#include <thread>
#include "ctx.h"
#include "comm.h"
#include "host_util.h"
#include "host_proof.h"

using namespace dsth;

extern "C" void dst_internal_transition_coefficients(const dst_ctx* c, const fe* draws344, std::vector<fe>& tc);   // api.hip
extern "C" int dst_internal_upload_draws(dst_ctx* c, const fe* draws344, const std::vector<fe>& tc, fe* d_coef, fe* d_tc);   // api.hip: queued from the page-locked staging area

enum { SH_TRACE_TREE = 0, SH_CONSTRAINT_TREE = 1, SH_FRI_TREE = 2, SH_CEVAL = 3, SH_FRI_LAST = 4, SH_FRI_SEND_CAP = 5 };
enum { RD_TRACE_LEAF = 0, RD_TRACE_NODE = 1, RD_TRACE_UPPER = 2, RD_CEVAL = 3, RD_C_NODE = 4, RD_C_UPPER = 5, RD_FRI_E = 6, RD_FRI_LEAF = 7,
       RD_FRI_NODE = 8, RD_FRI_UPPER = 9, RD_LDE_ROW = 10, RD_TEVAL = 11,
       RD_MID_OFFSET = 32 };            // RD_*_UPPER + RD_MID_OFFSET: the rank's subtree heap of a k-range tree (behind the 2G entries of the top heap)

static size_t fri_nd(const dst_ctx* c, int d) { return c->fri_size[d] / c->B; }     // elements per coset in layer d

// FRI across ranks.  The large layers stay sharded: coset-major evaluations, rank-local leaves and tree levels, one all-gather
// of boundary nodes per layer.  From the first layer that is small (at most 2^17 elements, 2 MiB) or has fewer than one
// 4-element row per coset, the ranks all-gather the layer's EVALUATIONS once and every rank finishes the commit phase on its
// own in natural order (same kernels as the single-GPU path): no further exchanges, and no limit on the blowup factor.  The
// remainder layer always qualifies, so the tail is never empty.
static int fri_replicated_from(const dst_ctx* c) {
    const char* e = c->sw("DISTAFF_FRI_REPLICATE_LOG");                      // tests lower the limit to get sharded layers at small sizes
    int log_limit = e ? atoi(e) : 17;
    log_limit = log_limit < 0 ? 0 : (log_limit > 40 ? 40 : log_limit);
    for (int d = 0; d < c->num_fri_layers; d++)
        if (c->fri_size[d] <= ((size_t)1 << log_limit) || fri_nd(c, d) < 4) return d;
    return c->num_fri_layers - 1;
}
static const fe* fri_layer_natural(const dst_ctx* c, int d) { return (d == 0 && c->fri_rep_from == 0) ? c->fri_nat0 : c->fri_e[d]; }
static bool fri_layer_replicated(const dst_ctx* c, int d) { return c->sharded_layout && d >= c->fri_rep_from; }

static int ensure_shard_buffers(dst_ctx* c) {
    if (c->gather_buf && c->d_status) return DST_OK;
    if (c->gather_buf) { HIP_TRY(c, hipMalloc((void**)&c->d_status, 1024)); return DST_OK; }
    const size_t n = c->n, G = c->prm.world;
    c->fri_rep_from = fri_replicated_from(c);
    size_t need = 32 * n * G;
    if (384 * n > need) need = 384 * n;
    if (c->fri_size[c->fri_rep_from] * 16 > need) need = c->fri_size[c->fri_rep_from] * 16;      // the gathered evaluations of the first replicated layer
    if (c->fri_rep_from == 0) HIP_TRY(c, hipMalloc((void**)&c->fri_nat0, c->fri_size[0] * sizeof(fe)));
    HIP_TRY(c, hipMalloc((void**)&c->gather_buf, need));
    c->gather_bytes = need;
    HIP_TRY(c, hipMalloc((void**)&c->trace_upper, 2 * n * G * sizeof(digest)));
    HIP_TRY(c, hipMalloc((void**)&c->c_upper, 2 * n * G * sizeof(digest)));
    for (int d = 0; d < c->num_fri_layers; d++) {
        size_t nb = c->fri_size[d] / c->B / 4;             // boundary nodes per rank = rows per coset
        HIP_TRY(c, hipMalloc((void**)&c->fri_upper[d], (2 * nb * G > 2 ? 2 * nb * G : 2) * sizeof(digest)));
    }
    HIP_TRY(c, hipMalloc((void**)&c->d_status, 1024));     // status records of dst_prove_sharded (one per rank)
    return DST_OK;
}
// Every buffer a collective of the sharded protocol touches exists from context creation on (api.hip, world > 1): a rank that fails
// locally at proving time (no trace uploaded, a bad argument) can still take part in every exchange and report its status through
// them, instead of leaving its peers in a collective it never enters.
extern "C" int dst_internal_shard_buffers(dst_ctx* c) { return ensure_shard_buffers(c); }
static double wall_ms_shard() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static int copy_in(dst_ctx* c, void* dst, const void* src, size_t bytes, int src_is_device) {
    if (src_is_device && dst == src) return DST_OK;              // dst_prove_sharded gathers straight into the landing buffer
    if (src_is_device) k_copy(c, dst, src, bytes);
    else HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    return DST_OK;
}
static int copy_out(dst_ctx* c, void* dst, const void* src, size_t bytes, int dst_is_device) {
    if (dst_is_device) k_copy(c, dst, src, bytes);
    else HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DST_OK;
}

extern "C" {

// steps 1-2, local part: LDE of the owned cosets, their leaf digests and the local tree levels (down to one node per k)
int dst_shard_commit_trace(dst_ctx* c) {
    if (!c) return DST_ERR_ARG;
    if (!c->have_trace) { c->err = "dst_shard_commit_trace: no trace uploaded"; return DST_ERR_STATE; }
    if (c->trace_owned_only) { c->err = "dst_shard_commit_trace: only this rank's registers were uploaded (dst_trace_upload_owned): use dst_prove_sharded"; return DST_ERR_STATE; }
    HIP_TRY(c, hipSetDevice(c->device));
    int r = ensure_shard_buffers(c);
    if (r) return r;
    c->sharded_layout = true;
    for (bool& b : c->tree_krange) b = false;                  // dst_prove_sharded switches trees to k-ranges as it exchanges them
    if (c->upload_pending) {
        // dst_trace_upload_async: the copies run on upload_stream; the transforms below read every register, so they wait for every group
        for (size_t g = 0; g + 1 < c->upload_bounds.size(); g++) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->upload_done[g], 0));
        c->upload_pending = false;
    }
    k_intt_columns(c, c->trace, c->trace_stride, c->polys, c->W);
    k_lde_columns(c, c->polys, c->lde, c->W);
    k_trace_leaves(c);
    k_merkle_levels_to(c, c->trace_leaves, c->trace_nodes, c->Bc * c->n, c->n);
    fe last[3];
    for (int i = 0; i < 3; i++) HIP_TRY(c, hipMemcpyAsync(&last[i], c->trace + (size_t)i * c->trace_stride + (c->n - 1), 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    c->op_count = (uint64_t)fe_to_u128(last[0]);
    c->program_hash[0] = last[1]; c->program_hash[1] = last[2];
    c->committed = true; c->constraints_done = c->composed = false;
    return DST_OK;
}

// step 3, local part: AIR evaluation on the owned evaluation cosets.  *bad_step = first failing trace step seen by this rank (-1: none)
static int shard_eval_constraints(dst_ctx* c, const dst_public* pub, const uint8_t* coeffs, int64_t* bad_step, bool defer_check);
int dst_shard_eval_constraints(dst_ctx* c, const dst_public* pub, const uint8_t* coeffs, int64_t* bad_step) { return shard_eval_constraints(c, pub, coeffs, bad_step, false); }
// defer_check: the host does not wait for the verdict; the failing step (device word c->d_u64, ~0 = none) is picked up later
static int shard_eval_constraints(dst_ctx* c, const dst_public* pub, const uint8_t* coeffs, int64_t* bad_step, bool defer_check) {
    if (!c || !pub || !coeffs) return DST_ERR_ARG;
    if (!c->committed) { c->err = "dst_shard_eval_constraints: trace not committed"; return DST_ERR_STATE; }
    HIP_TRY(c, hipSetDevice(c->device));
    c->pub = *pub;
    std::vector<fe> draws(344), tc;
    memcpy(draws.data(), coeffs, 344 * 16);
    c->shard_draws = draws;                                  // dst_shard_combine builds the boundary polynomials from them
    dst_internal_transition_coefficients(c, draws.data(), tc);
    fe* d_coef = c->scratch + c->scratch_elems - 1024;
    fe* d_tc = d_coef + 344;
    if (int ru = dst_internal_upload_draws(c, draws.data(), tc, d_coef, d_tc)) return ru;
    int r = k_eval_constraints(c, d_coef, d_tc, bad_step, defer_check);
    if (r == DST_ERR_AIR) c->err = "transition constraints were not satisfied";
    return r;
}

// steps 4-5 after the constraint evaluations of all ranks were imported (SH_CEVAL): combination (replicated), LDE of the owned
// cosets and the local levels of the constraint tree
// parts: 1 = the two boundary combinations and their divisions (need nothing from other ranks), 2 = transition part, sum, extension over
// the owned cosets and the local tree levels.  dst_prove_sharded runs part 1 while the exchange of the evaluations is in flight.
static int shard_combine_parts(dst_ctx* c, int parts) {
    const size_t n = c->n, D = 8 * n;
    fe* ip = c->cwork; fe* fp = c->cwork + D; fe* tp = c->cwork + 2 * D; fe* work = c->cwork + 3 * D;
    const bool steps = dst_internal_combine_by_steps(c);         // the reference's sequence of whole-array steps (tests), else the fused pass of the single-GPU path
    fe* q4 = c->cwork; const size_t qs = n + 16;
    if (parts & 1) {
        if (!steps) {
            int rb = dst_internal_boundary_quotients(c, c->shard_draws.data(), q4, qs);
            if (rb) return rb;
        } else {
            if (dst_internal_boundary_by_evaluation(c)) {
                k_intt8_cosets(c, c->ceval, ip, work);
                k_intt8_cosets(c, c->ceval + D, fp, work);
            } else {
                int rb = dst_internal_boundary_polys(c, c->shard_draws.data(), ip, fp);
                if (rb) return rb;
            }
            k_syn_div(c, ip, D, fe_one());
            k_syn_div(c, fp, D, c->x_last);
        }
    }
    if (parts & 2) {
        if (!steps) {
            const fe* inv = c->ceval + 2 * D;                   // already inverse-transformed per coset by its owner (dst_prove_sharded) ...
            if (!c->ceval_inverted) { k_intt_cosets_local(c, c->ceval + 2 * D, work, 8); inv = work; }      // ... or not (host-orchestrated path)
            c->ceval_inverted = false;
            k_combine_fused(c, inv, q4, qs, c->cpoly);
        } else {
            if (c->ceval_inverted) { k_cross8(c, c->ceval + 2 * D, tp); c->ceval_inverted = false; }
            else k_intt8_cosets(c, c->ceval + 2 * D, tp, work);
            k_syn_div_expanded(c, tp, c->cpoly, D, n, c->x_last);
            k_add(c, c->cpoly, ip, D);
            k_add(c, c->cpoly, fp, D);
        }
        if (c->wait_comm && c->sh_ev[4]) (void)hipEventRecord(c->sh_ev[4], c->stream);      // dst_prove_sharded: end of the combination (phase 3 | 4)
        k_lde_fold8(c, c->cpoly, c->cevals);
        if (c->Bc >= 4) {                                   // with two cosets per rank the leaves themselves are the boundary (see dst_shard_export)
            k_constraint_level1(c);
            k_merkle_local_levels(c, c->cnodes, c->Bc * n / 4, n);
        }
        c->constraints_done = true; c->composed = false;
    }
    return DST_OK;
}
int dst_shard_combine(dst_ctx* c) {
    if (!c) return DST_ERR_ARG;
    if (!c->committed || c->shard_draws.size() != 344) { c->err = "dst_shard_combine: constraints not evaluated (dst_shard_eval_constraints first)"; return DST_ERR_STATE; }
    HIP_TRY(c, hipSetDevice(c->device));
    int r = shard_combine_parts(c, 3);
    if (r) return r;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    return DST_OK;
}

// step 7: leaves and local tree levels of the current FRI layer (all layers are coset-major in sharded mode)
int dst_shard_fri_layer(dst_ctx* c, int* more) {
    if (!c || !more) return DST_ERR_ARG;
    if (!c->composed) { c->err = "dst_shard_fri_layer: composition not built"; return DST_ERR_STATE; }
    int d = c->fri_committed;
    if (d >= c->num_fri_layers || d != c->fri_folded) { c->err = "dst_shard_fri_layer: fold the previous layer first"; return DST_ERR_STATE; }
    if (d >= c->fri_rep_from) { c->err = "dst_shard_fri_layer: this layer is part of the replicated tail (dst_shard_fri_begin / dst_shard_fri_end)"; return DST_ERR_STATE; }
    HIP_TRY(c, hipSetDevice(c->device));
    size_t nd = fri_nd(c, d), nb = nd / 4;
    k_fri_leaves_cm(c, c->fri_e[d], c->fri_leaves[d], nd);
    k_merkle_levels_to(c, c->fri_leaves[d], c->fri_nodes[d], nb * c->Bc, nb);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    c->fri_committed = d + 1;
    *more = (d + 1 < c->num_fri_layers) ? 1 : 0;
    return DST_OK;
}
int dst_shard_fri_fold(dst_ctx* c, const uint8_t special_x[16]) {
    if (!c || !special_x) return DST_ERR_ARG;
    int d = c->fri_folded;
    if (d + 1 != c->fri_committed || d + 1 >= c->num_fri_layers) { c->err = "dst_shard_fri_fold: nothing to fold"; return DST_ERR_STATE; }
    HIP_TRY(c, hipSetDevice(c->device));
    k_fri_fold_cm(c, c->fri_e[d], c->fri_e[d + 1], fri_nd(c, d), d, fe_from_bytes(special_x));
    c->fri_folded = d + 1;
    return DST_OK;
}

// size in bytes of what this rank exports for (what, arg)
int dst_shard_export_size(dst_ctx* c, uint32_t what, uint32_t arg, size_t* bytes) {
    if (!c || !bytes) return DST_ERR_ARG;
    switch (what) {
        case SH_TRACE_TREE: case SH_CONSTRAINT_TREE: *bytes = c->n * 32; return DST_OK;
        case SH_FRI_TREE: if ((int)arg >= c->num_fri_layers) return DST_ERR_ARG; *bytes = fri_nd(c, arg) / 4 * 32; return DST_OK;
        case SH_CEVAL: *bytes = (dst_internal_boundary_by_evaluation(c) ? 3 : 1) * (c->Bc / (c->B / 8)) * c->n * 16; return DST_OK;   // [i, f,] t
        case SH_FRI_LAST: *bytes = c->fri_size[c->num_fri_layers - 1] * 16; return DST_OK;       // the whole remainder, natural order (replicated)
        case SH_FRI_SEND_CAP: {                                  // the largest item dst_shard_fri_begin hands out
            const int t = c->gather_buf ? c->fri_rep_from : fri_replicated_from(c);    // fixed once the shard buffers exist
            size_t m = c->Bc * fri_nd(c, t) * 16;
            for (int d = 0; d < t; d++) if (fri_nd(c, d) / 4 * 32 > m) m = fri_nd(c, d) / 4 * 32;
            *bytes = m; return DST_OK;
        }
    }
    return DST_ERR_ARG;
}
int dst_shard_export(dst_ctx* c, uint32_t what, uint32_t arg, void* dst, int dst_is_device) {
    if (!c || !dst) return DST_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    size_t bytes = 0;
    if (dst_shard_export_size(c, what, arg, &bytes)) { c->err = "dst_shard_export: bad item"; return DST_ERR_ARG; }
    const void* src = nullptr;
    switch (what) {
        case SH_TRACE_TREE: src = c->trace_nodes + c->n; break;
        case SH_CONSTRAINT_TREE: src = c->cnodes + c->n; break;
        case SH_FRI_TREE: src = c->fri_nodes[arg] + fri_nd(c, arg) / 4; break;
        case SH_CEVAL: src = dst_internal_boundary_by_evaluation(c) ? c->ceval : c->ceval + 2 * (c->Bc / (c->B / 8)) * c->n; break;   // local layout [3][Q][n]
        case SH_FRI_LAST:
            if (c->fri_committed != c->num_fri_layers) { c->err = "dst_shard_export: FRI commit phase not finished"; return DST_ERR_STATE; }
            src = fri_layer_natural(c, c->num_fri_layers - 1); break;
        default: c->err = "dst_shard_export: bad item"; return DST_ERR_ARG;
    }
    if (what == SH_CONSTRAINT_TREE && c->Bc == 2) {
        // two cosets per rank: a rank holds exactly one constraint-tree leaf per k, the raw pair (C(x_{B k + 2g}), C(x_{B k + 2g + 1}))
        // (prover.rs:180-187), and no node below the replicated part: the "boundary nodes" are the leaves, interleaved from the
        // coset-major evaluations [2][n]
        k_coset_to_natural_len(c, c->cevals, 2, c->n, (fe*)c->cnodes);      // [2][n] elements -> n pairs; the local heap is unused in this case
        return copy_out(c, dst, c->cnodes, bytes, dst_is_device);
    }
    return copy_out(c, dst, src, bytes, dst_is_device);
}
// `src` = the all-gathered items of all ranks, rank-major.  For trees the replicated upper part is built and its root returned.
int dst_shard_import(dst_ctx* c, uint32_t what, uint32_t arg, const void* src, int src_is_device, uint8_t root_out[32]) {
    if (!c || !src) return DST_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    int r = ensure_shard_buffers(c);
    if (r) return r;
    size_t bytes = 0;
    if (dst_shard_export_size(c, what, arg, &bytes)) { c->err = "dst_shard_import: bad item"; return DST_ERR_ARG; }
    const size_t G = c->prm.world, total = bytes * G;
    if (total > c->gather_bytes) { c->err = "dst_shard_import: gather buffer too small"; return DST_ERR_ARG; }
    if ((r = copy_in(c, c->gather_buf, src, total, src_is_device))) return r;
    if (what == SH_CEVAL) {
        c->ceval_inverted = false;                                  // dst_prove_sharded sets it after importing arrays it has inverse-transformed
        const size_t Q = c->Bc / (c->B / 8), blk = Q * c->n;        // gathered [G][V][Q][n] -> ceval [3][8][n], V = 3 (i, f, t) or 1 (t)
        const size_t V = dst_internal_boundary_by_evaluation(c) ? 3 : 1;
        CTX_SYNC(c, "an imported tree");
        for (size_t g = 0; g < G; g++)
            for (size_t v = 0; v < V; v++)
                HIP_TRY(c, hipMemcpyAsync(c->ceval + ((V == 3 ? v : 2) * 8 + g * Q) * c->n, (const fe*)c->gather_buf + (g * V + v) * blk, blk * 16, hipMemcpyDeviceToDevice, c->stream));
        CTX_SYNC(c, "an imported tree");
        return DST_OK;
    }
    digest* upper = nullptr; size_t nb = 0;
    switch (what) {
        case SH_TRACE_TREE: upper = c->trace_upper; nb = c->n; break;
        case SH_CONSTRAINT_TREE: upper = c->c_upper; nb = c->n; break;
        case SH_FRI_TREE: upper = c->fri_upper[arg]; nb = fri_nd(c, arg) / 4; break;
        default: c->err = "dst_shard_import: item cannot be imported"; return DST_ERR_ARG;
    }
    k_upper_tree(c, (const digest*)c->gather_buf, upper, nb, (uint32_t)G);
    uint8_t root[32];
    HIP_TRY(c, hipMemcpyAsync(root, upper + 1, 32, hipMemcpyDeviceToHost, c->stream));
    CTX_SYNC(c, "the imported constraint evaluations");
    HIP_TRY(c, hipGetLastError());
    if (what == SH_TRACE_TREE) memcpy(c->trace_root, root, 32);
    else if (what == SH_CONSTRAINT_TREE) memcpy(c->constraint_root, root, 32);
    else { if (c->fri_roots.size() <= arg) c->fri_roots.resize(arg + 1); c->fri_roots[arg].assign(root, root + 32); }
    if (root_out) memcpy(root_out, root, 32);
    return DST_OK;
}

// fetches `count` items by LOCAL index from one of the rank's buffers (openings); RD_LDE_ROW returns W elements per index
int dst_shard_read(dst_ctx* c, uint32_t buffer, uint32_t arg, const uint64_t* idx, uint32_t count, uint8_t* out) {
    if (!c || !idx || !out) return DST_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (count == 0) return DST_OK;
    const void* src = nullptr; size_t item = 32;
    switch (buffer) {
        case RD_TRACE_LEAF: src = c->trace_leaves; break;
        case RD_TRACE_NODE: src = c->trace_nodes; break;
        case RD_TRACE_UPPER: src = c->trace_upper; break;
        case RD_CEVAL: src = c->cevals; item = 16; break;
        case RD_C_NODE: src = c->cnodes; break;
        case RD_C_UPPER: src = c->c_upper; break;
        case RD_FRI_E: if ((int)arg >= c->num_fri_layers) return DST_ERR_ARG; src = fri_layer_replicated(c, (int)arg) ? fri_layer_natural(c, (int)arg) : c->fri_e[arg]; item = 16; break;
        case RD_FRI_LEAF: if ((int)arg >= c->num_fri_layers) return DST_ERR_ARG; src = c->fri_leaves[arg]; break;
        case RD_FRI_NODE: if ((int)arg >= c->num_fri_layers) return DST_ERR_ARG; src = c->fri_nodes[arg]; break;
        case RD_FRI_UPPER: if ((int)arg >= c->num_fri_layers) return DST_ERR_ARG; src = fri_layer_replicated(c, (int)arg) ? c->fri_nodes[arg] : c->fri_upper[arg]; break;
        case RD_LDE_ROW: break;
        case RD_TEVAL: src = c->ceval + 2 * (c->Bc / (c->B / 8)) * c->n; item = 16; break;        // transition combination, this rank's evaluation cosets [Q][n]
        default: c->err = "dst_shard_read: unknown buffer"; return DST_ERR_ARG;
    }
    size_t idx_bytes = ((size_t)count * 8 + 15) / 16 * 16;
    size_t out_bytes = buffer == RD_LDE_ROW ? (size_t)count * c->W * 16 : (size_t)count * item;
    if (idx_bytes + out_bytes > c->stage_bytes) { c->err = "dst_shard_read: staging buffer too small"; return DST_ERR_ARG; }
    uint64_t* d_idx = (uint64_t*)c->d_stage; uint8_t* d_out = c->d_stage + idx_bytes;
    if (buffer == RD_LDE_ROW) {
        // idx holds natural positions B*k + j owned by this rank
        HIP_TRY(c, hipMemcpyAsync(d_idx, idx, (size_t)count * 8, hipMemcpyHostToDevice, c->stream));
        k_gather_rows(c, d_idx, count, (fe*)d_out);
    } else {
        if (!src) { c->err = "dst_shard_read: buffer not allocated (world == 1?)"; return DST_ERR_STATE; }
        HIP_TRY(c, hipMemcpyAsync(d_idx, idx, (size_t)count * 8, hipMemcpyHostToDevice, c->stream));
        k_gather(c, src, item, d_idx, count, d_out);
    }
    HIP_TRY(c, hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    return DST_OK;
}

// One FRI layer in two calls around the all-gather (fri/prover.rs:11-53): begin = leaves + local tree levels of the next layer and
// its boundary nodes into the caller's send buffer; end = import of the gathered boundary nodes (layer root), and, if another
// layer follows, the Fiat-Shamir draw from that root (field::prng) and the fold.
int dst_shard_fri_begin(dst_ctx* c, void* send, int send_is_device, size_t cap, size_t* bytes, int* more) {
    if (!c || !send || !bytes || !more) return DST_ERR_ARG;
    int r = ensure_shard_buffers(c);
    if (r) return r;
    if (c->fri_tail_pending) { c->err = "dst_shard_fri_begin: finish the pending exchange with dst_shard_fri_end first"; return DST_ERR_STATE; }
    if (c->fri_committed == c->fri_rep_from) {
        // first replicated layer: hand out this rank's cosets of its evaluations; dst_shard_fri_end finishes the commit phase
        if (!c->composed) { c->err = "dst_shard_fri_begin: composition not built"; return DST_ERR_STATE; }
        const int d = c->fri_rep_from;
        if (d != c->fri_folded) { c->err = "dst_shard_fri_begin: fold the previous layer first"; return DST_ERR_STATE; }
        HIP_TRY(c, hipSetDevice(c->device));
        *bytes = c->Bc * fri_nd(c, d) * 16;
        if (*bytes > cap) { c->err = "dst_shard_fri_begin: send buffer too small"; return DST_ERR_ARG; }
        *more = 0;
        c->fri_tail_pending = true;
        return copy_out(c, send, c->fri_e[d], *bytes, send_is_device);
    }
    r = dst_shard_fri_layer(c, more);
    if (r) return r;
    *more = 1;                                                  // the replicated tail follows every sharded layer
    const uint32_t d = (uint32_t)c->fri_committed - 1;
    if ((r = dst_shard_export_size(c, SH_FRI_TREE, d, bytes))) return r;
    if (*bytes > cap) { c->err = "dst_shard_fri_begin: send buffer too small"; return DST_ERR_ARG; }
    return dst_shard_export(c, SH_FRI_TREE, d, send, send_is_device);
}
// commit phase of the replicated tail (fri/prover.rs:11-53 from layer d0 on): every rank on its own, natural order
static int fri_replicated_tail(dst_ctx* c, const void* gathered, int src_is_device, uint8_t root_out[32]) {
    const int d0 = c->fri_rep_from, L = c->num_fri_layers;
    const size_t size0 = c->fri_size[d0];
    int r = copy_in(c, c->gather_buf, gathered, size0 * 16, src_is_device);            // rank-major pieces = coset-major [B][nd]
    if (r) return r;
    fe* nat0 = d0 == 0 ? c->fri_nat0 : c->fri_e[d0];
    k_coset_to_natural_len(c, (const fe*)c->gather_buf, c->B, fri_nd(c, d0), nat0);
    if (c->fri_roots.size() < (size_t)L) c->fri_roots.resize(L);
    const char* tail_env = c->sw("DISTAFF_FRI_TAIL");
    // layers above the single-launch tail: no host round trip per layer -- x = prng(root) is drawn on the device (fri_draw_kernel) and the
    // fold reads it there; their roots come back with the tail's (as in dst_prove's commit phase, api.hip)
    digest* d_roots = reinterpret_cast<digest*>(c->d_fri_chain);
    fe* d_alpha = reinterpret_cast<fe*>(c->d_fri_chain + DST_MAX_FRI_LAYERS * 32);
    uint8_t* h_roots = c->h_stage + HS_FRI_ROOTS;
    int d = d0;
    for (; d < L; d++) {
        if (d >= 1 && c->fri_size[d] <= ((size_t)1 << 13) && !(tail_env && tail_env[0] == '0')) break;
        const size_t R = c->fri_size[d] / 4;
        const fe* e = fri_layer_natural(c, d);
        k_fri_leaves_at(c, e, c->fri_leaves[d], R);
        k_merkle_levels(c, c->fri_leaves[d], c->fri_nodes[d], R);
        k_fri_draw(c, d, d_alpha + d, d_roots + d);
        if (d + 1 < L) k_fri_fold_at(c, e, c->fri_e[d + 1], R, d, fe_zero(), d_alpha + d);
    }
    if (d > d0) HIP_TRY(c, hipMemcpyAsync(h_roots, d_roots + d0, (size_t)(d - d0) * 32, hipMemcpyDeviceToHost, c->stream));
    if (d < L) {
        // the small layers in one launch (k_fri_tail, as on a single GPU): the layer's evaluations are in fri_e[d] in natural order
        std::vector<uint8_t> rs((size_t)(L - d) * 32);
        int rt = k_fri_tail(c, d, rs.data());                   // synchronises the stream
        if (rt) return rt;
        for (int i = d; i < L; i++) c->fri_roots[i].assign(rs.begin() + 32 * (i - d), rs.begin() + 32 * (i - d + 1));
    } else {
        CTX_SYNC(c, "the gathered FRI layer");
        HIP_TRY(c, hipGetLastError());
    }
    for (int i = d0; i < d; i++) c->fri_roots[i].assign(h_roots + 32 * (i - d0), h_roots + 32 * (i - d0 + 1));
    if (root_out) memcpy(root_out, c->fri_roots[d0].data(), 32);
    c->fri_committed = L; c->fri_folded = L - 1;
    c->fri_tail_pending = false;
    return DST_OK;
}
int dst_shard_fri_end(dst_ctx* c, const void* gathered, int src_is_device, uint8_t root_out[32]) {
    if (!c || !gathered || !root_out) return DST_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->fri_tail_pending) return fri_replicated_tail(c, gathered, src_is_device, root_out);
    if (c->fri_committed == 0) { c->err = "dst_shard_fri_end: no layer in flight"; return DST_ERR_STATE; }
    const uint32_t d = (uint32_t)c->fri_committed - 1;
    int r = dst_shard_import(c, SH_FRI_TREE, d, gathered, src_is_device, root_out);
    if (r) return r;
    fe x;
    prng_vector(root_out, 1, &x);                              // fri/prover.rs:40 field::prng(root)
    uint8_t xb[16]; memcpy(xb, &x, 16);
    return dst_shard_fri_fold(c, xb);                          // a sharded layer is never the last one
}
// roots of all FRI layers once the commit phase is over (the layers of the replicated tail are committed inside dst_shard_fri_end)
int dst_shard_fri_roots(dst_ctx* c, uint8_t* roots, size_t cap, uint32_t* num_layers, uint32_t* replicated_from) {
    if (!c || !num_layers) return DST_ERR_ARG;
    *num_layers = (uint32_t)c->num_fri_layers;
    if (replicated_from) *replicated_from = (uint32_t)(c->gather_buf ? c->fri_rep_from : fri_replicated_from(c));
    if (!roots) return DST_OK;
    if (c->fri_committed != c->num_fri_layers || (int)c->fri_roots.size() < c->num_fri_layers) { c->err = "dst_shard_fri_roots: FRI commit phase not finished"; return DST_ERR_STATE; }
    if (cap < (size_t)32 * c->num_fri_layers) { c->err = "dst_shard_fri_roots: buffer too small"; return DST_ERR_ARG; }
    for (int d = 0; d < c->num_fri_layers; d++) memcpy(roots + 32 * d, c->fri_roots[d].data(), 32);
    return DST_OK;
}

// ---- step 9 across ranks -------------------------------------------------------------------------------------------------------------
// Every rank derives the same ordered request list from the query positions (which leaf / node / element / row the proof
// needs, who owns it, where it sits in the owner's buffers) and the same proof template with one slot per request.
// dst_shard_open gathers the items this rank owns into one blob (one batched device gather, one copy back);
// dst_shard_assemble fills the template from the blobs of all ranks.  The orchestration all-gathers the blobs in between.
namespace {
struct OpenReq { int owner; uint32_t buffer, arg; uint64_t index; uint32_t bytes; };       // owner -1: replicated, served by rank 0
struct OpenPlan { std::vector<OpenReq> reqs; std::vector<size_t> slot; Writer w; };

struct Geometry {                      // distaff_amd/sharded.py TreeGeometry: leaves Bt*k + j', columns j' split over G ranks
    uint64_t L, Bt, G, Bct, K;
    bool krange;                       // levels between the rank-local ones and the top log2(G): owned by k-ranges (dst_prove_sharded) instead of replicated
    Geometry(uint64_t leaves, uint64_t bt, uint64_t g, bool kr = false) : L(leaves), Bt(bt), G(g), Bct(bt / g), K(leaves / bt), krange(kr && g > 1) {}
    void leaf(uint64_t i, int& g, uint64_t& local) const { uint64_t k = i / Bt, j = i % Bt; g = (int)(j / Bct); local = k * Bct + (j - (uint64_t)g * Bct); }
    // zone: 0 rank-local heap, 1 replicated heap (global indices), 2 the owner's subtree heap (k-range mode)
    void node(uint64_t heap, int& g, uint64_t& idx, int* zone = nullptr) const {
        uint64_t level = 1; while (level * 2 <= heap) level *= 2;              // nodes on this level
        uint64_t t = heap - level, span = L / level;
        if (zone) *zone = 0;
        if (span >= Bct) {
            if (krange && level > G) {                                         // subtree of the rank that owns k in [g*K/G, (g+1)*K/G): level/G nodes of this level each
                const uint64_t per = level / G;
                g = (int)(t / per); idx = per + t % per;
                if (zone) *zone = 2;
                return;
            }
            g = -1; idx = heap;
            if (zone) *zone = 1;
            return;
        }
        uint64_t local_leaf; leaf(t * span, g, local_leaf);
        idx = L / (G * span) + local_leaf / span;                              // nodes of this level held by one rank = L / (G * span), also when L < Bt
    }
};

void plan_item(OpenPlan& p, int owner, uint32_t buffer, uint32_t arg, uint64_t index, uint32_t bytes) {
    p.reqs.push_back({owner, buffer, arg, index, bytes});
    p.slot.push_back(p.w.b.size());
    p.w.b.resize(p.w.b.size() + bytes);
}
// element at natural position pos of a coset-major [B][nd] array
void plan_element(const dst_ctx* c, OpenPlan& p, uint32_t buffer, uint32_t arg, uint64_t pos, uint64_t nd) {
    if (buffer == RD_FRI_E && arg > 0 && !c->sharded_layout) { plan_item(p, 0, buffer, arg, pos, 16); return; }     // single-GPU phases keep layers >= 1 in natural order
    if (buffer == RD_FRI_E && fri_layer_replicated(c, (int)arg)) { plan_item(p, -1, buffer, arg, pos, 16); return; }   // replicated tail of the sharded phases: natural order, served by rank 0
    uint64_t k = pos / c->B, j = pos % c->B, g = j / c->Bc;
    plan_item(p, (int)g, buffer, arg, (j - g * c->Bc) * nd + k, 16);
}
void plan_tree_nodes(const dst_ctx* c, OpenPlan& p, const BatchPlan& bp, const Geometry& geo, uint32_t leaf_buf, uint32_t node_buf, uint32_t upper_buf,
                     uint32_t arg, bool raw_pair_leaves) {
    p.w.u64(bp.nodes.size());
    for (auto& l : bp.nodes) {
        p.w.u64(l.size());
        for (auto& r : l) {
            int g; uint64_t idx;
            if (r.is_leaf) {
                if (raw_pair_leaves) { plan_element(c, p, RD_CEVAL, 0, 2 * r.index, c->n); plan_element(c, p, RD_CEVAL, 0, 2 * r.index + 1, c->n); }
                else { geo.leaf(r.index, g, idx); plan_item(p, g, leaf_buf, arg, idx, 32); }
            } else {
                int zone = 0;
                geo.node(r.index, g, idx, &zone);
                plan_item(p, g, zone == 0 ? node_buf : zone == 1 ? upper_buf : upper_buf + RD_MID_OFFSET, arg, idx, 32);
            }
        }
    }
}

// the serialised StarkProof (proof.rs:11-22) with empty slots; same field order as dst_build_proof (api.hip)
int build_open_plan(dst_ctx* c, const uint64_t* positions_in, uint32_t num_positions, uint64_t pow_nonce, OpenPlan& p) {
    const uint64_t G = c->prm.world, B = c->B, N = c->N, n = c->n, W = c->W;
    std::vector<uint64_t> positions(positions_in, positions_in + num_positions);
    for (uint64_t q : positions) if (q >= N) { c->err = "query position out of range"; return DST_ERR_ARG; }
    {   // MerkleTree::prove_batch asserts this (merkle.rs:69): a repeated position would silently drop an opening from the batch proof
        std::set<uint64_t> seen(positions.begin(), positions.end());
        if (seen.size() != positions.size()) { c->err = "repeating indexes detected"; return DST_ERR_ARG; }
    }
    const int L = c->num_fri_layers;
    if ((int)c->fri_roots.size() < L) { c->err = "dst_shard_open: FRI commit phase not finished"; return DST_ERR_STATE; }
    Writer& w = p.w;
    w.raw(c->trace_root, 32);
    w.u8((uint8_t)c->log_N); w.u8((uint8_t)c->prm.ctx_depth); w.u8((uint8_t)c->prm.loop_depth); w.u8((uint8_t)c->stack_depth); w.u32((uint32_t)c->op_count);
    BatchPlan tp = plan_batch(positions, N);
    plan_tree_nodes(c, p, tp, Geometry(N, B, G, c->sharded_layout && c->tree_krange[0]), RD_TRACE_LEAF, RD_TRACE_NODE, RD_TRACE_UPPER, 0, false);
    w.u64(positions.size());
    for (uint64_t q : positions) { w.u64(W); plan_item(p, (int)((q % B) / c->Bc), RD_LDE_ROW, 0, q, (uint32_t)(W * 16)); }
    w.raw(c->constraint_root, 32);
    {
        std::vector<uint64_t> cpos = constraint_positions(positions);
        BatchPlan cp = plan_batch(cpos, N / 2);
        w.u64(cp.values.size());
        for (uint64_t u : cp.values) { plan_element(c, p, RD_CEVAL, 0, 2 * u, n); plan_element(c, p, RD_CEVAL, 0, 2 * u + 1, n); }
        plan_tree_nodes(c, p, cp, Geometry(N / 2, B / 2, G, c->sharded_layout && c->tree_krange[1]), 0, RD_C_NODE, RD_C_UPPER, 0, true);
        w.u8(cp.depth);
    }
    w.u64(W); w.raw(c->deep_z1.data(), W * 16);
    w.u64(W); w.raw(c->deep_z2.data(), W * 16);
    std::vector<uint64_t> pos = positions;
    w.u64((uint64_t)(L - 1));
    for (int d = 0; d + 1 < L; d++) {
        const uint64_t size = c->fri_size[d], R = size / 4, nd = size / B;
        pos = augmented_positions(pos, size);
        BatchPlan fp = plan_batch(pos, R);
        w.raw(c->fri_roots[d].data(), 32);
        w.u64(pos.size());
        for (uint64_t r : pos) for (uint64_t s4 = 0; s4 < 4; s4++) plan_element(c, p, RD_FRI_E, (uint32_t)d, r + s4 * R, nd);
        plan_tree_nodes(c, p, fp, Geometry(R, B, fri_layer_replicated(c, d) ? 1 : G, c->sharded_layout && c->tree_krange[2 + d]), RD_FRI_LEAF, RD_FRI_NODE, RD_FRI_UPPER, (uint32_t)d, false);
        w.u8(fp.depth);
    }
    w.raw(c->fri_roots[L - 1].data(), 32);
    const uint64_t rem = c->fri_size[L - 1];
    w.u64(rem);
    for (uint64_t i = 0; i < rem; i++) plan_element(c, p, RD_FRI_E, (uint32_t)(L - 1), i, rem / B);
    w.u64(pow_nonce);
    w.u8((uint8_t)c->log_b); w.u8((uint8_t)c->prm.num_queries); w.u8((uint8_t)c->prm.grinding_factor); w.u8(0);
    return DST_OK;
}

const void* read_source(dst_ctx* c, uint32_t buffer, uint32_t arg) {
    switch (buffer) {
        case RD_TRACE_LEAF: return c->trace_leaves;
        case RD_TRACE_NODE: return c->trace_nodes;
        case RD_TRACE_UPPER: return c->sharded_layout ? c->trace_upper : c->trace_nodes;      // single-GPU phases: one full heap per tree
        case RD_CEVAL: return c->cevals;
        case RD_C_NODE: return c->cnodes;
        case RD_C_UPPER: return c->sharded_layout ? c->c_upper : c->cnodes;
        case RD_FRI_E: return (int)arg < c->num_fri_layers ? (fri_layer_replicated(c, (int)arg) ? (const void*)fri_layer_natural(c, (int)arg) : (const void*)c->fri_e[arg]) : nullptr;
        case RD_FRI_LEAF: return (int)arg < c->num_fri_layers ? c->fri_leaves[arg] : nullptr;
        case RD_FRI_NODE: return (int)arg < c->num_fri_layers ? c->fri_nodes[arg] : nullptr;
        case RD_FRI_UPPER: return (int)arg < c->num_fri_layers ? (c->sharded_layout && !fri_layer_replicated(c, (int)arg) ? c->fri_upper[arg] : c->fri_nodes[arg]) : nullptr;
        // subtree heaps of k-range trees sit behind the 2G entries of the replicated top heap
        case RD_TRACE_UPPER + RD_MID_OFFSET: return c->trace_upper + 2 * c->prm.world;
        case RD_C_UPPER + RD_MID_OFFSET: return c->c_upper + 2 * c->prm.world;
        case RD_FRI_UPPER + RD_MID_OFFSET: return (int)arg < c->num_fri_layers ? c->fri_upper[arg] + 2 * c->prm.world : nullptr;
    }
    return nullptr;
}
}  // namespace

// gathers the requests selected by `me` / `everything` (one staging upload, one gather launch, one copy back) into `blob`, concatenated in
// request order
static int gather_requests(dst_ctx* c, const OpenPlan& p, int me, bool everything, std::vector<uint8_t>& blob) {
    std::vector<uint64_t> addr;                     // device address of every 16-byte piece
    addr.reserve(16384);
    // every requested item as 16-byte pieces in blob order (an LDE row = W pieces, one per register: element (register, coset, k) of the
    // coset-major extension): ONE batched gather writes the blob as it is, one copy brings it back
    size_t total = 0;
    for (auto& r : p.reqs) {
        if (!(everything || r.owner == me || (r.owner < 0 && me == 0))) continue;
        if (r.buffer == RD_LDE_ROW) {
            const uint64_t j = r.index % c->B, k = r.index / c->B;
            if (j < c->j0 || j >= c->j0 + c->Bc) { c->err = "openings: LDE row of a coset this rank does not own"; return DST_ERR_ARG; }
            const fe* base = c->lde + (j - c->j0) * c->n + k;
            for (size_t reg = 0; reg < c->W; reg++) addr.push_back((uint64_t)(uintptr_t)(base + reg * c->Bc * c->n));
        } else {
            const uint8_t* src = (const uint8_t*)read_source(c, r.buffer, r.arg);
            if (!src) { c->err = "openings: buffer not allocated"; return DST_ERR_STATE; }
            for (uint32_t o = 0; o < r.bytes; o += 16) addr.push_back((uint64_t)(uintptr_t)(src + r.index * r.bytes + o));
        }
        total += r.bytes;
    }
    blob.resize(total);
    if (addr.size() * 16 != total) { c->err = "openings: an item is not a multiple of 16 bytes"; return DST_ERR_STATE; }
    const size_t idx_bytes = (addr.size() * 8 + 15) / 16 * 16;
    if (idx_bytes + total > c->stage_bytes) { c->err = "openings: staging buffer too small"; return DST_ERR_ARG; }
    if (!addr.empty()) HIP_TRY(c, hipMemcpyAsync(c->d_stage, addr.data(), addr.size() * 8, hipMemcpyHostToDevice, c->stream));
    uint8_t* d_out = c->d_stage + idx_bytes;
    k_gather_pieces(c, (const uint64_t*)c->d_stage, addr.size(), d_out);
    if (total) HIP_TRY(c, hipMemcpyAsync(blob.data(), d_out, total, hipMemcpyDeviceToHost, c->stream));
    CTX_SYNC(c, "the openings");
    HIP_TRY(c, hipGetLastError());
    return DST_OK;
}

// single-GPU dst_build_proof (api.hip): one plan, every item local
int dst_internal_build_proof(dst_ctx* c, const uint64_t* positions, uint32_t num_positions, uint64_t pow_nonce, std::vector<uint8_t>& proof) {
    OpenPlan p;
    int rc = build_open_plan(c, positions, num_positions, pow_nonce, p);
    if (rc) return rc;
    std::vector<uint8_t> blob;
    if ((rc = gather_requests(c, p, 0, true, blob))) return rc;
    // the slots follow each other in request order; consecutive items with no template bytes between them are copied as one run
    size_t cur = 0;
    for (size_t i = 0; i < p.reqs.size();) {
        size_t run = p.reqs[i].bytes, e = i + 1;
        while (e < p.reqs.size() && p.slot[e] == p.slot[i] + run) { run += p.reqs[e].bytes; e++; }
        memcpy(p.w.b.data() + p.slot[i], blob.data() + cur, run);
        cur += run; i = e;
    }
    proof.swap(p.w.b);
    return DST_OK;
}

// this rank's items, concatenated in request order
int dst_shard_open(dst_ctx* c, const uint64_t* positions, uint32_t num_positions, uint8_t* blob, size_t cap, size_t* blob_len, uint64_t* all_lens) {
    if (!c || !positions || !blob_len) return DST_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    std::shared_ptr<OpenPlan> sp;
    const std::vector<uint64_t> key(positions, positions + num_positions);
    // the size query just before -- for the same positions AND the same commitments (the template embeds the roots and the values at z)
    if (c->open_plan && c->open_plan_positions == key && !memcmp(c->open_plan_root, c->trace_root, 32)) sp = std::static_pointer_cast<OpenPlan>(c->open_plan);
    int rc = DST_OK;
    if (!sp) {
        sp = std::make_shared<OpenPlan>();
        if ((rc = build_open_plan(c, positions, num_positions, 0, *sp))) return rc;
        c->open_plan = sp; c->open_plan_positions = key; memcpy(c->open_plan_root, c->trace_root, 32);
    }
    OpenPlan& p = *sp;
    const int me = (int)c->prm.rank;
    size_t total = 0;
    for (auto& r : p.reqs) if (r.owner == me || (r.owner < 0 && me == 0)) total += r.bytes;
    *blob_len = total;
    if (all_lens) {                                        // every rank's share follows from the same plan
        for (uint32_t g = 0; g < c->prm.world; g++) all_lens[g] = 0;
        for (auto& r : p.reqs) all_lens[r.owner < 0 ? 0 : r.owner] += r.bytes;
    }
    if (!blob) return DST_OK;
    if (cap < total) { c->err = "dst_shard_open: blob buffer too small"; return DST_ERR_ARG; }
    std::vector<uint8_t> mine;
    if ((rc = gather_requests(c, p, me, false, mine))) return rc;
    memcpy(blob, mine.data(), mine.size());
    return DST_OK;
}

// blobs: the ranks' blobs back to back, blob_lens[g] bytes each
int dst_shard_assemble(dst_ctx* c, const uint64_t* positions, uint32_t num_positions, uint64_t pow_nonce, const uint8_t* blobs, const uint64_t* blob_lens,
                       uint8_t* out, size_t cap, size_t* out_len) {
    if (!c || !positions || !blobs || !blob_lens || !out_len) return DST_ERR_ARG;
    // the plan of the preceding dst_shard_open for the same positions is reused (a copy: the slots get filled); the nonce is the only
    // field that was not known then and sits 12 bytes before the end of the template
    OpenPlan p;
    int rc = DST_OK;
    const std::vector<uint64_t> key(positions, positions + num_positions);
    const bool cached = c->open_plan && c->open_plan_positions == key && !memcmp(c->open_plan_root, c->trace_root, 32);
    if (cached) {
        p = *std::static_pointer_cast<OpenPlan>(c->open_plan);
        for (int i = 0; i < 8; i++) p.w.b[p.w.b.size() - 12 + i] = (uint8_t)(pow_nonce >> (8 * i));
        if (out) { c->open_plan.reset(); c->open_plan_positions.clear(); }      // a size query (out == NULL) keeps the plan for the call that follows
    } else if ((rc = build_open_plan(c, positions, num_positions, pow_nonce, p))) return rc;
    const size_t G = c->prm.world;
    std::vector<size_t> cursor(G, 0), base(G, 0);
    for (size_t g = 1; g < G; g++) base[g] = base[g - 1] + blob_lens[g - 1];
    for (size_t i = 0; i < p.reqs.size(); i++) {
        const size_t g = p.reqs[i].owner < 0 ? 0 : (size_t)p.reqs[i].owner;
        if (cursor[g] + p.reqs[i].bytes > blob_lens[g]) { c->err = "dst_shard_assemble: a rank's blob is shorter than its share of the openings"; return DST_ERR_ARG; }
        memcpy(p.w.b.data() + p.slot[i], blobs + base[g] + cursor[g], p.reqs[i].bytes);
        cursor[g] += p.reqs[i].bytes;
    }
    for (size_t g = 0; g < G; g++) if (cursor[g] != blob_lens[g]) { c->err = "dst_shard_assemble: a rank's blob is longer than its share of the openings"; return DST_ERR_ARG; }
    *out_len = p.w.b.size();
    if (!out) return DST_OK;
    if (cap < p.w.b.size()) { c->err = "proof buffer too small"; return DST_ERR_ARG; }
    memcpy(out, p.w.b.data(), p.w.b.size());
    return DST_OK;
}

// host-side view of the last dst_prove_sharded on this rank: [0] milliseconds inside the transport's calls, [1] milliseconds waiting for
// the tree roots (the only points where the host waits for the device), [2] number of tree exchanges
int dst_shard_stage_ms(const dst_ctx* c, double out[3]) {
    if (!c || !out) return DST_ERR_ARG;
    out[0] = c->shard_ms[0]; out[1] = c->shard_ms[1]; out[2] = (double)c->shard_trees;
    return DST_OK;
}

int dst_shard_exchange_ms(const dst_ctx* c, double out[8]) {
    if (!c || !out) return DST_ERR_ARG;
    for (int i = 0; i < 8; i++) out[i] = c->exchange_ms[i];
    return DST_OK;
}
int dst_shard_info(dst_ctx* c, uint64_t* op_count, uint32_t* num_fri_layers, uint32_t* stack_depth) {
    if (!c) return DST_ERR_ARG;
    if (op_count) *op_count = c->op_count;
    if (num_fri_layers) *num_fri_layers = (uint32_t)c->num_fri_layers;
    if (stack_depth) *stack_depth = (uint32_t)c->stack_depth;
    return DST_OK;
}


// ---- the whole sharded proof behind the C-ABI ------------------------------------------------------------------------------------------
// dst_prove_sharded = stark::prove (prover.rs:17-168) across the ranks of a communicator, with the collectives issued here (RCCL or the
// in-process transport, comm.hip).  What is split and what is exchanged:
//   * interpolation by trace COLUMNS: rank g interpolates the registers r = g (mod G) and the coefficient vectors are all-gathered in
//     place, in rounds of G registers (16 n bytes sent per rank and round); on a stream-ordered transport the all-gather of round k + 1
//     runs on its own stream while round k is extended.  A host that uploads with dst_trace_upload_owned sends each GPU 1/G of the trace;
//   * everything evaluated on the LDE domain by COSETS: rank g owns the cosets [g B/G, (g+1) B/G) of every register;
//   * every Merkle tree: after the rank-local levels (one boundary node per k and rank) an ALL-TO-ALL moves the boundary nodes so that
//     rank g holds those of k in [g*K/G, (g+1)*K/G) of every rank, builds the subtree above them (K - 1 hashes), the G subtree roots
//     are all-gathered (G x 32 bytes) and only the top log2(G) levels are repeated on every rank.  Per rank and tree: 32*K*(G-1)/G bytes
//     sent, K + G - 2 hashes above the local levels (the host-orchestrated path all-gathers 32*K*(G-1) bytes and repeats K*G - 1 hashes);
//   * the transition-constraint evaluations (8n/G elements per rank): all-gather before the cross-coset inverse transform;
//   * the first small FRI layer: all-gather of its evaluations, the commit phase then finishes replicated;
//   * openings: every rank gathers what it owns; the blobs are all-gathered and every rank fills the same proof template.
// Failure handling.  A rank that fails locally keeps ISSUING the collectives of the protocol (its peers wait in them) but skips its own
// work; its status travels with data that is exchanged anyway -- ONE 96-byte record per rank and tree (subtree root + status; for the
// constraint tree also the first failing step of the rank's constraint check, copied from the device word the kernels wrote: no host
// exchange and no wait of its own after the evaluation), read back with the root, and inside the opening-length exchange.  Every rank
// returns the first failing rank's code.  (Round 2 agreed on a status word through a host-staged all-gather after every phase: ~26 per
// proof; round 3 sent roots and status in two collectives per tree and exchanged the failing step through the host.)
namespace {
struct StatusRec { int32_t rc; int32_t pad; uint64_t bad_step; fe payload[3]; };      // 64 bytes per rank; bad_step: first trace step whose constraints fail on this rank's cosets (~0: none; constraint tree only); payload: rank 0's last trace state (op counter, program hash)
static_assert(sizeof(StatusRec) == 64, "StatusRec is exchanged as 64 bytes");
struct TreeRec { digest root; StatusRec st; };                         // what a rank contributes to the ONE small all-gather of a tree exchange: its subtree root and its status
static_assert(sizeof(TreeRec) == 96, "TreeRec is exchanged as 96 bytes");

struct Sharded {
    dst_ctx* c; dst_comm* comm;
    int rc = DST_OK;                       // this rank's own status: sticky, skips its local steps
    int agreed = DST_OK;                   // first failing rank's code once an exchange has shown one
    // a rank-local step; collectives are issued regardless (see above)
    void local(const std::function<int()>& f) { if (rc == DST_OK) { rc = f(); } }           // (this file's functions sit in an extern "C" block: no member templates)
    // host time inside the transport's calls (an enqueue on a stream-ordered transport, the whole exchange on a blocking one)
    int timed(const std::function<int()>& f) { const double t = wall_ms_shard(); const int r = f(); c->shard_ms[0] += wall_ms_shard() - t; return r; }
    // ... and, for a collective on device buffers, two events around it on the stream it is queued on: enqueue -> completion as the device
    // saw it (on RCCL that includes the wait for the slowest peer), read once after the proof's last wait (dst_shard_exchange_ms).
    // kind: index into dst_ctx::exchange_ms (0 coefficients, 1 tree all-to-all, 2 tree all-gather, 3 constraint evaluations, 4 FRI tail)
    int coll_on(int kind, hipStream_t stream, const std::function<int()>& f) {
        dst_ctx::CollEv* ev = nullptr;
        if (c->coll_used < c->coll_ev.size()) ev = &c->coll_ev[c->coll_used];
        else if (c->coll_ev.size() < 256) {
            dst_ctx::CollEv n{nullptr, nullptr, 0};
            if (hipEventCreate(&n.e0) == hipSuccess && hipEventCreate(&n.e1) == hipSuccess) { c->coll_ev.push_back(n); ev = &c->coll_ev.back(); }
            else { if (n.e0) hipEventDestroy(n.e0); (void)hipGetLastError(); }
        }
        c->exchange_ms[6] += 1;
        if (ev && hipEventRecord(ev->e0, stream) != hipSuccess) { (void)hipGetLastError(); ev = nullptr; }
        const int r = timed(f);
        if (ev && hipEventRecord(ev->e1, stream) == hipSuccess) { ev->kind = kind; c->coll_used++; }
        return r;
    }
    // all-gather of host values: complete on return, host wall time
    int coll_host(const std::function<int()>& f) { const double t = wall_ms_shard(); const int r = timed(f); c->exchange_ms[5] += wall_ms_shard() - t; c->exchange_ms[6] += 1; return r; }
    void read_collective_events() {
        for (size_t i = 0; i < c->coll_used; i++) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, c->coll_ev[i].e0, c->coll_ev[i].e1) == hipSuccess) { c->exchange_ms[c->coll_ev[i].kind] += ms; c->exchange_ms[7] += 1; }
            else (void)hipGetLastError();
        }
    }
    void fail(int code, const std::string& msg) { if (rc == DST_OK) { rc = code; c->err = msg; } }
    // collective errors are not rank-local: the transport failed for everyone (or will hang for everyone)
    bool coll(int r, const char* what) { if (r != DST_OK) { if (rc == DST_OK) { rc = r; c->err = std::string(what) + ": " + comm->err; } agreed = agreed ? agreed : r; return false; } return true; }
    void saw(const StatusRec* all, const char* phase) {
        for (uint32_t g = 0; g < comm->world && agreed == DST_OK; g++)
            if (all[g].rc != DST_OK) { agreed = all[g].rc; if (rc == DST_OK) c->err = std::string(phase) + ": rank " + std::to_string(g) + " reported error " + std::to_string(all[g].rc); }
    }
};

// this rank's boundary nodes of a tree (device pointer, K of them): the level at which nodes stop being rank-local
int shard_boundary(dst_ctx* c, uint32_t what, uint32_t arg, const digest** src, size_t* K) {
    switch (what) {
        case SH_TRACE_TREE: *src = c->trace_nodes + c->n; *K = c->n; return DST_OK;
        case SH_CONSTRAINT_TREE:
            *K = c->n;
            if (c->Bc == 2) {                                    // two cosets per rank: the raw leaf pairs are the boundary (see dst_shard_export)
                k_coset_to_natural_len(c, c->cevals, 2, c->n, (fe*)c->cnodes);
                *src = c->cnodes;
            } else *src = c->cnodes + c->n;
            return DST_OK;
        case SH_FRI_TREE: *K = fri_nd(c, (int)arg) / 4; *src = c->fri_nodes[arg] + *K; return DST_OK;
    }
    c->err = "shard_boundary: bad item"; return DST_ERR_ARG;
}

// Finishes a tree from the ranks' boundary nodes and returns its root (replicated).  The ranks' status records ride with the subtree
// roots: on return S.agreed holds the first failing rank's code.  `payload` (rank 0 -> everyone), when given, is three field elements
// read from rank 0's device memory at `payload_src[i]`.
// defer_slot != nullptr: nothing is waited for -- the records and the root are copied into that page-locked slot (G * 96 + 32 bytes) behind the
// exchange, and tree_exchange_finish looks at them after the caller's next synchronisation (the sharded FRI layers: their x is drawn on
// the device from the root, so the host needs neither root nor status before the end of the commit phase).
void tree_exchange_finish(Sharded& S, uint32_t what, uint32_t arg, const uint8_t* slot, fe* payload_out, int64_t* first_bad);
void tree_exchange(Sharded& S, uint32_t what, uint32_t arg, uint8_t root[32], const fe* const* payload_src = nullptr, fe* payload_out = nullptr,
                   const uint64_t* bad_src = nullptr, int64_t* first_bad = nullptr, uint8_t* defer_slot = nullptr) {
    dst_ctx* c = S.c; dst_comm* comm = S.comm;
    const size_t G = comm->world;
    const digest* src = nullptr; size_t K = 0;
    if (shard_boundary(c, what, arg, &src, &K)) { S.fail(DST_ERR_ARG, c->err); S.agreed = DST_ERR_ARG; return; }      // the same on every rank
    digest* upper = what == SH_TRACE_TREE ? c->trace_upper : what == SH_CONSTRAINT_TREE ? c->c_upper : c->fri_upper[arg];
    const int slot = what == SH_TRACE_TREE ? 0 : what == SH_CONSTRAINT_TREE ? 1 : 2 + (int)arg;
    c->shard_trees++;
    const bool krange = G > 1 && K >= G && K % G == 0 && !c->sw_flag("DISTAFF_SHARD_TREE_GATHER");
    c->tree_krange[slot] = krange;
    if (c->sw_flag("DISTAFF_SHARD_DEBUG") && comm->rank == 0) fprintf(stderr, "[distaff] tree %u/%u: %zu boundary nodes per rank, %s\n", what, arg, K, krange ? "k-range exchange (all-to-all + root all-gather)" : "all-gather of boundary nodes");
    if ((krange ? K * 32 : K * 32 * G) > c->gather_bytes) { S.fail(DST_ERR_ARG, "tree_exchange: gather buffer too small"); S.agreed = DST_ERR_ARG; return; }   // the same on every rank
    // this rank's record of the exchange: status (and rank 0's payload) staged now, the subtree root joins it below -- ONE small all-gather
    // per tree carries both (the roots used to travel in a collective of their own)
    // (queued from a slot of the page-locked staging area: with defer_slot nothing below waits for the stream before `mine` would go out of scope)
    StatusRec& mine = reinterpret_cast<StatusRec*>(c->h_stage + HS_STATUS)[(c->shard_trees - 1) % HS_STATUS_SLOTS];
    mine = StatusRec{}; mine.rc = S.rc; mine.bad_step = ~0ull;
    TreeRec* recs = reinterpret_cast<TreeRec*>(c->d_status);
    bool staged = hipMemcpyAsync(&recs[comm->rank].st, &mine, sizeof(mine), hipMemcpyHostToDevice, c->stream) == hipSuccess;
    // the verdict of this rank's constraint evaluation, straight from the device word the kernels wrote (no host wait of its own)
    if (staged && bad_src && S.rc == DST_OK) staged = hipMemcpyAsync(&recs[comm->rank].st.bad_step, bad_src, 8, hipMemcpyDeviceToDevice, c->stream) == hipSuccess;
    if (staged && payload_src && comm->rank == 0 && S.rc == DST_OK)
        for (int i = 0; i < 3 && staged; i++) staged = hipMemcpyAsync(&recs[0].st.payload[i], payload_src[i], sizeof(fe), hipMemcpyDeviceToDevice, c->stream) == hipSuccess;
    if (!staged) S.fail(DST_ERR_HIP, "tree_exchange: staging of the status record failed");
    if (krange) {
        const size_t chunk = K / G;                              // boundary nodes per (sender, owner) pair
        if (!S.coll(S.coll_on(1, c->stream, [&] { return comm->all_to_all(src, c->gather_buf, chunk * 32, c->stream); }), "tree_exchange")) return;
        digest* mid = upper + 2 * G;                             // this rank's subtree heap: mid[1] = its root, mid[K + kl*G + r] = boundary node of rank r at k = g*K/G + kl
        S.local([&]() -> int {
            k_upper_tree(c, (const digest*)c->gather_buf, mid, chunk, (uint32_t)G);
            HIP_TRY(c, hipMemcpyAsync(&recs[comm->rank].root, mid + 1, sizeof(digest), hipMemcpyDeviceToDevice, c->stream));
            return DST_OK;
        });
        if (!S.coll(S.coll_on(2, c->stream, [&] { return comm->all_gather(recs + comm->rank, recs, sizeof(TreeRec), c->stream); }), "tree_exchange")) return;
        S.local([&] { k_digests_from_records(c, recs, sizeof(TreeRec), upper + G, G); k_merkle_upper(c, upper, G); return DST_OK; });      // the top log2(G) levels, on every rank
    } else {
        if (!S.coll(S.coll_on(2, c->stream, [&] { return comm->all_gather(src, c->gather_buf, K * 32, c->stream); }), "tree_exchange")) return;
        S.local([&] { k_upper_tree(c, (const digest*)c->gather_buf, upper, K, (uint32_t)G); return DST_OK; });
        if (!S.coll(S.coll_on(2, c->stream, [&] { return comm->all_gather(recs + comm->rank, recs, sizeof(TreeRec), c->stream); }), "tree_exchange")) return;
    }
    if (defer_slot) {
        bool okd = hipMemcpyAsync(defer_slot, recs, G * sizeof(TreeRec), hipMemcpyDeviceToHost, c->stream) == hipSuccess;
        okd = okd && hipMemcpyAsync(defer_slot + G * sizeof(TreeRec), upper + 1, 32, hipMemcpyDeviceToHost, c->stream) == hipSuccess;
        if (!okd) S.fail(DST_ERR_HIP, "tree_exchange: read-back could not be queued");
        return;
    }
    const double t_wait = wall_ms_shard();
    // into PAGE-LOCKED memory: a copy into pageable memory would make the host wait inside hipMemcpyAsync for everything queued before it --
    // the exchange included -- without any bound (found by the stalled-collective test over the stream-ordered in-process transport)
    uint8_t* const host = c->h_stage + HS_TREE_READBACK;
    bool ok = hipMemcpyAsync(host, recs, G * sizeof(TreeRec), hipMemcpyDeviceToHost, c->stream) == hipSuccess;
    ok = ok && hipMemcpyAsync(host + G * sizeof(TreeRec), upper + 1, 32, hipMemcpyDeviceToHost, c->stream) == hipSuccess;
    // the host waits here for everything queued before the root, kernels and exchanges: a bounded poll (a peer that never joined the
    // exchange would otherwise hold this rank for ever); on expiry the communicator is aborted and the rank returns DST_ERR_COMM
    const int rw = ok ? ctx_sync(c, what == SH_TRACE_TREE ? "the root of the trace tree" : what == SH_CONSTRAINT_TREE ? "the root of the constraint tree" : "the root of a FRI tree") : DST_ERR_HIP;
    ok = ok && rw == DST_OK && hipGetLastError() == hipSuccess;
    c->shard_ms[1] += wall_ms_shard() - t_wait;
    if (!ok) { const int code = rw ? rw : DST_ERR_HIP; S.fail(code, rw ? c->err : std::string("tree_exchange: root read-back failed")); S.agreed = S.agreed ? S.agreed : code; return; }
    tree_exchange_finish(S, what, arg, host, payload_out, first_bad);
    if (root) memcpy(root, host + G * sizeof(TreeRec), 32);
}
// the ranks' records and the root of a tree exchange, once they are on the host: first failing rank's code into S.agreed, else the root is filed
void tree_exchange_finish(Sharded& S, uint32_t what, uint32_t arg, const uint8_t* slot, fe* payload_out, int64_t* first_bad) {
    dst_ctx* c = S.c;
    const size_t G = S.comm->world;
    std::vector<TreeRec> all(G);
    memcpy(all.data(), slot, G * sizeof(TreeRec));
    const uint8_t* root = slot + G * sizeof(TreeRec);
    std::vector<StatusRec> sts(G);
    for (size_t g = 0; g < G; g++) sts[g] = all[g].st;
    S.saw(sts.data(), what == SH_TRACE_TREE ? "trace tree" : what == SH_CONSTRAINT_TREE ? "constraint tree" : "FRI tree");
    if (payload_out) for (int i = 0; i < 3; i++) payload_out[i] = all[0].st.payload[i];
    if (first_bad) { *first_bad = -1; for (size_t g = 0; g < G; g++) if (all[g].st.rc == DST_OK && all[g].st.bad_step != ~0ull && (*first_bad < 0 || (int64_t)all[g].st.bad_step < *first_bad)) *first_bad = (int64_t)all[g].st.bad_step; }
    if (S.agreed) return;
    if (what == SH_TRACE_TREE) memcpy(c->trace_root, root, 32);
    else if (what == SH_CONSTRAINT_TREE) memcpy(c->constraint_root, root, 32);
    else { if (c->fri_roots.size() <= arg) c->fri_roots.resize(arg + 1); c->fri_roots[arg].assign(root, root + 32); }
}

// steps 1-2 up to the rank-local tree levels.  Interpolation is split by COLUMNS: rank g interpolates the registers r = g (mod G); the
// coefficient vectors of a round of G registers are all-gathered in place (polys[round * G + rank] is this rank's piece), and each round
// is extended over this rank's cosets as soon as it has arrived.
void commit_trace_columns(Sharded& S) {
    dst_ctx* c = S.c; dst_comm* comm = S.comm;
    const size_t G = comm->world, W = c->W, n = c->n, rounds = (W + G - 1) / G;
    S.local([&]() -> int {
        if (!c->have_trace) { c->err = "dst_prove_sharded: no trace uploaded"; return DST_ERR_STATE; }
        c->sharded_layout = true;
        for (bool& b : c->tree_krange) b = false;
        if (c->upload_pending) {
            for (size_t g = 0; g + 1 < c->upload_bounds.size(); g++) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->upload_done[g], 0));
            c->upload_pending = false;
        }
        return DST_OK;
    });
    // a stream-ordered transport runs the all-gathers on their own stream, ordered against the transforms by events
    // (DISTAFF_SHARD_FORCE_OVERLAP=1: the same stream / event choreography over a blocking transport -- how the tests reach this path
    // without several RCCL ranks)
    const bool overlap = (comm->stream_ordered() || c->sw_flag("DISTAFF_SHARD_FORCE_OVERLAP")) && G > 1 && !c->sw_flag("DISTAFF_SHARD_NO_OVERLAP");
    if (overlap) S.local([&]() -> int {
        if (!c->comm_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
        while (c->comm_events.size() < 2 * rounds) { hipEvent_t e; HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming)); c->comm_events.push_back(e); }
        return DST_OK;
    });
    const bool use_side = overlap && S.rc == DST_OK;
    if (G == 1) {                                              // one rank: nothing to exchange, all registers in one pair of launches
        S.local([&]() -> int {
            k_intt_columns(c, c->trace, c->trace_stride, c->polys, W);
            k_lde_columns(c, c->polys, c->lde, W);
            if (c->sh_ev[1]) (void)hipEventRecord(c->sh_ev[1], c->stream);        // end of the extension (phase 0 | 1)
            k_trace_leaves(c);
            k_merkle_levels_to(c, c->trace_leaves, c->trace_nodes, c->Bc * n, n);
            return DST_OK;
        });
        return;
    }
    // interpolate the owned register of every round (a rank past the last register of the final round contributes an unused piece)
    for (size_t k = 0; k < rounds; k++) {
        const size_t col = k * G + comm->rank;
        S.local([&]() -> int {
            if (col < W) k_intt_columns(c, c->trace + col * c->trace_stride, c->trace_stride, c->polys + col * n, 1);
            if (use_side) { HIP_TRY(c, hipEventRecord(c->comm_events[2 * k], c->stream)); HIP_TRY(c, hipStreamWaitEvent(c->comm_stream, c->comm_events[2 * k], 0)); }
            return DST_OK;
        });
        if (use_side) {
            if (!S.coll(S.coll_on(0, c->comm_stream, [&] { return comm->all_gather(c->polys + col * n, c->polys + k * G * n, n * sizeof(fe), c->comm_stream); }), "coefficient all-gather")) return;
            S.local([&]() -> int { HIP_TRY(c, hipEventRecord(c->comm_events[2 * k + 1], c->comm_stream)); return DST_OK; });
        }
    }
    for (size_t k = 0; k < rounds; k++) {
        const size_t first = k * G, cnt = first + G <= W ? G : W - first;
        if (use_side) S.local([&]() -> int { HIP_TRY(c, hipStreamWaitEvent(c->stream, c->comm_events[2 * k + 1], 0)); return DST_OK; });
        else if (!S.coll(S.coll_on(0, c->stream, [&] { return comm->all_gather(c->polys + (first + comm->rank) * n, c->polys + first * n, n * sizeof(fe), c->stream); }), "coefficient all-gather")) return;
        S.local([&]() -> int { k_lde_columns(c, c->polys + first * n, c->lde + first * c->Bc * n, cnt); return DST_OK; });
    }
    if (c->sh_ev[1]) (void)hipEventRecord(c->sh_ev[1], c->stream);        // end of the extension (phase 0 | 1)
    S.local([&]() -> int {
        k_trace_leaves(c);
        k_merkle_levels_to(c, c->trace_leaves, c->trace_nodes, c->Bc * n, n);
        return DST_OK;
    });
}
}  // namespace

int dst_prove_sharded(dst_ctx* c, dst_comm* comm, const dst_public* pub, uint8_t* proof_out, size_t cap, size_t* proof_len) {
    if (!c || !comm) return DST_ERR_ARG;                      // nothing to agree through: the caller's bug, peers are its to stop
    Sharded S{c, comm};
    // A communicator of another shape than the context's cannot take part at all: the exchange sizes follow comm->world, the buffers
    // prm.world (in-place all-gathers would run past them).  Returned before the first collective; the caller built both handles, and its
    // peers leave their first wait after the communicator's limit (dst_comm_set_timeout) instead of hanging.
    if (comm->world != c->prm.world || comm->rank != c->prm.rank) { c->err = "dst_prove_sharded: the communicator's rank / world differ from the context's"; return DST_ERR_ARG; }
    if (comm->world > 8) return DST_ERR_ARG;                  // contexts cannot be created for more
    // Rank-local pre-flight failures travel with the first status record like every later one: the rank keeps issuing its collectives.
    if (hipSetDevice(c->device) != hipSuccess) { (void)hipGetLastError(); S.fail(DST_ERR_HIP, "dst_prove_sharded: hipSetDevice failed"); }
    // The exchange buffers exist from context creation when world > 1 (dst_ctx_create fails otherwise); a one-rank context gets them here.
    // A rank WITHOUT them has nothing to hand to a collective: it gives its communicator up (in-process peers wake at once, RCCL peers
    // leave their first wait after the limit) -- unreachable through the public API, kept so that it cannot become a hang.
    if (S.rc == DST_OK) { const int rb = ensure_shard_buffers(c); if (rb) S.fail(rb, c->err); }
    if (!c->gather_buf || !c->d_status) { comm->abort("dst_prove_sharded: rank " + std::to_string(comm->rank) + " has no exchange buffers"); return S.rc ? S.rc : DST_ERR_HIP; }
    if (!pub || !proof_len) S.fail(DST_ERR_ARG, "dst_prove_sharded: null argument");
    // from here on every host wait on this context's stream is the communicator's bounded poll (ctx_sync)
    struct WaitScope { dst_ctx* c; ~WaitScope() { c->wait_comm = nullptr; } } wait_scope{c};
    c->wait_comm = comm;
    const size_t G = comm->world;
    double t0 = wall_ms_shard();
    c->shard_ms[0] = c->shard_ms[1] = 0; c->shard_trees = 0;
    for (double& x : c->exchange_ms) x = 0;
    c->coll_used = 0;
    auto mark = [&](int i) { const double t = wall_ms_shard(); c->phase_ms[i] = t - t0; t0 = t; };
    // Phase boundaries are events on the stream (the reference's nine timings, prover.rs:28,36,66,74,87,103,112,134,167): the host only
    // ENQUEUES between two waits, so its own clock says nothing about where the device's time went.  sh_ev: 0 start, 1 end of the
    // extension, 2 / 3 around the constraint evaluation, 4 end of the combination.
    bool timed = true;
    for (int i = 0; i < 5 && timed; i++) if (!c->sh_ev[i] && hipEventCreate(&c->sh_ev[i]) != hipSuccess) { (void)hipGetLastError(); c->sh_ev[i] = nullptr; timed = false; }
    auto ev_ms = [&](int a, int b) -> double { float ms = 0; if (timed && hipEventElapsedTime(&ms, c->sh_ev[a], c->sh_ev[b]) == hipSuccess) return ms; (void)hipGetLastError(); return -1.0; };
    // steps 1-2.  Nothing waits for the extension on the host (the tree exchange is queued behind it)
    timed = timed && hipEventRecord(c->sh_ev[0], c->stream) == hipSuccess;
    commit_trace_columns(S);                                    // records sh_ev[1] behind the last extension launch
    if (S.agreed) return S.agreed;                              // a collective itself failed
    const double t_commit = t0;
    uint8_t trace_root[32], constraint_root[32];
    {
        // last state of the un-extended trace (op counter, program hash: evaluator.rs:37,73-74): rank 0 owns coset 0 of the extension,
        // which IS the trace; the three values ride with the status records
        const size_t stride = c->Bc * c->n;
        const fe* from[3] = {c->lde + (c->n - 1), c->lde + stride + (c->n - 1), c->lde + 2 * stride + (c->n - 1)};
        fe last[3];
        tree_exchange(S, SH_TRACE_TREE, 0, trace_root, from, last);
        if (S.agreed) return S.agreed;
        c->op_count = (uint64_t)fe_to_u128(last[0]);
        c->program_hash[0] = last[1]; c->program_hash[1] = last[2];
        c->committed = true; c->constraints_done = c->composed = false;
    }
    {
        // phase 0 = interpolation + extension (the stream's own clock up to sh_ev[1]), phase 1 = leaves, tree levels, the exchange and the
        // host's share: the rest of the wall time up to the root
        const double now = wall_ms_shard(), both = now - t_commit, dev_ms = S.rc == DST_OK ? ev_ms(0, 1) : -1.0;
        if (dev_ms >= 0 && dev_ms < both) { c->phase_ms[0] = dev_ms; c->phase_ms[1] = both - dev_ms; }
        else { c->phase_ms[0] = 0; c->phase_ms[1] = both; }
        t0 = now;
    }
    // step 3
    std::vector<fe> coef(344);
    prng_vector(trace_root, 344, coef.data());
    // the evaluation is queued; its verdict (first failing step of this rank's cosets, evaluator.rs:152-158) rides with the status records of the
    // constraint tree's exchange below instead of a host exchange and a wait of its own -- a trace that fails is reported one phase later
    const double t_eval = t0;
    if (timed) (void)hipEventRecord(c->sh_ev[2], c->stream);
    S.local([&] { return shard_eval_constraints(c, pub, (const uint8_t*)coef.data(), nullptr, true); });
    if (timed) (void)hipEventRecord(c->sh_ev[3], c->stream);
    // steps 4-5: the transition evaluations of all ranks, then combination (replicated) and the constraint tree
    {
        size_t bytes = 0;
        if (dst_shard_export_size(c, SH_CEVAL, 0, &bytes) || bytes * G > c->gather_bytes) { c->err = "dst_prove_sharded: gather buffer too small for the constraint evaluations"; return DST_ERR_ARG; }   // the same on every rank
        const size_t Q = c->Bc / (c->B / 8);
        const void* send = dst_internal_boundary_by_evaluation(c) ? (const void*)c->ceval : (const void*)(c->ceval + 2 * Q * c->n);
        const bool invert_first = !dst_internal_boundary_by_evaluation(c);
        if (invert_first) S.local([&]() -> int {
            // the size-n inverse transforms of this rank's evaluation cosets, before the exchange (in place through the scratch area)
            fe* mine = c->ceval + 2 * Q * c->n; fe* work = c->cwork + 3 * 8 * c->n;
            k_intt_cosets_local(c, mine, work, Q);
            HIP_TRY(c, hipMemcpyAsync(mine, work, Q * c->n * sizeof(fe), hipMemcpyDeviceToDevice, c->stream));
            return DST_OK;
        });
        // On a stream-ordered transport the exchange runs on the collective stream while this rank writes the boundary combinations
        // (they need nothing from other ranks); otherwise in sequence.  In boundary-by-evaluation mode part 1 reads the gathered arrays.
        bool part1_done = false;
        const bool side = (comm->stream_ordered() || c->sw_flag("DISTAFF_SHARD_FORCE_OVERLAP")) && G > 1 && invert_first && S.rc == DST_OK && c->comm_stream && c->comm_events.size() >= 2 && !c->sw_flag("DISTAFF_SHARD_NO_OVERLAP");
        if (side) {
            S.local([&]() -> int { HIP_TRY(c, hipEventRecord(c->comm_events[0], c->stream)); HIP_TRY(c, hipStreamWaitEvent(c->comm_stream, c->comm_events[0], 0)); return DST_OK; });
            if (!S.coll(S.coll_on(3, c->comm_stream, [&] { return comm->all_gather(send, c->gather_buf, bytes, c->comm_stream); }), "constraint evaluations")) return S.agreed;
            S.local([&]() -> int { HIP_TRY(c, hipEventRecord(c->comm_events[1], c->comm_stream)); return DST_OK; });
            S.local([&] { part1_done = true; return shard_combine_parts(c, 1); });
            S.local([&]() -> int { HIP_TRY(c, hipStreamWaitEvent(c->stream, c->comm_events[1], 0)); return DST_OK; });
        } else if (!S.coll(S.coll_on(3, c->stream, [&] { return comm->all_gather(send, c->gather_buf, bytes, c->stream); }), "constraint evaluations")) return S.agreed;        // the import below rewrites `ceval` only after the exchange
        S.local([&] { const int r = dst_shard_import(c, SH_CEVAL, 0, c->gather_buf, 1, nullptr); c->ceval_inverted = invert_first && r == DST_OK; return r; });
        S.local([&] { return shard_combine_parts(c, part1_done ? 2 : 3); });       // records sh_ev[4] between the combination and the extension of its result
    }
    {
        int64_t first_bad = -1;
        tree_exchange(S, SH_CONSTRAINT_TREE, 0, constraint_root, nullptr, nullptr, c->d_u64, &first_bad);
        if (S.agreed) return S.agreed;
        if (first_bad >= 0) { c->err = "transition constraints were not satisfied at step " + std::to_string(first_bad); return DST_ERR_AIR; }
    }
    {
        // phases 2 (constraint evaluation), 3 (combination incl. the exchange of the evaluations), 4 (extension of the constraint polynomial,
        // its tree, the tree exchange): the host queued all three without a wait, so they are split at the stream's events; what the host spent
        // before the first and after the last event goes to the outer phases, as in dst_eval_constraints
        const double now = wall_ms_shard(), total = now - t_eval;
        const double e1 = S.rc == DST_OK ? ev_ms(2, 3) : -1.0, e2 = S.rc == DST_OK ? ev_ms(3, 4) : -1.0;
        if (e1 >= 0 && e2 >= 0 && e1 + e2 <= total) { c->phase_ms[2] = e1; c->phase_ms[3] = e2; c->phase_ms[4] = total - e1 - e2; }
        else { c->phase_ms[2] = 0; c->phase_ms[3] = 0; c->phase_ms[4] = total; }
        t0 = now;
    }
    // step 6
    std::vector<fe> draws(516);
    prng_vector(constraint_root, 516, draws.data());
    std::vector<uint8_t> z1(c->W * 16), z2(c->W * 16);
    S.local([&] { return dst_compose(c, (const uint8_t*)draws.data(), z1.data(), z2.data()); });
    mark(5);
    // step 7: sharded layers (leaves + local levels | tree exchange with status | draw + fold), then the replicated tail from one
    // all-gather of evaluations.  fri_rep_from is fixed by the parameters: every rank walks the same layers.
    const int rep_from = c->gather_buf ? c->fri_rep_from : fri_replicated_from(c);
    // Sharded layers.  chained (default): no host wait per layer -- x = prng(root) is drawn on the device from the replicated root
    // (fri_draw_kernel) and the fold reads it there; the layers' records and roots are looked at once, after the tail.
    // DISTAFF_FRI_CHAIN=0: root read-back, host draw and status check per layer (tests).
    const char* ce = c->sw("DISTAFF_FRI_CHAIN");
    const bool chained = !(ce && ce[0] == '0');
    fe* d_alpha = reinterpret_cast<fe*>(c->d_fri_chain + DST_MAX_FRI_LAYERS * 32);
    digest* d_roots = reinterpret_cast<digest*>(c->d_fri_chain);
    auto slot_of = [&](int d) { return c->h_stage + HS_FRI_SLOTS + (size_t)d * HS_FRI_SLOT_BYTES; };       // page-locked: G * 96 + 32 bytes per layer
    for (int d = 0; d < rep_from; d++) {
        S.local([&]() -> int {
            if (!c->composed || d != c->fri_committed || d != c->fri_folded) { c->err = "dst_prove_sharded: FRI layer out of order"; return DST_ERR_STATE; }
            const size_t nd = fri_nd(c, d), nb = nd / 4;
            k_fri_leaves_cm(c, c->fri_e[d], c->fri_leaves[d], nd);
            k_merkle_levels_to(c, c->fri_leaves[d], c->fri_nodes[d], nb * c->Bc, nb);
            c->fri_committed = d + 1;
            return DST_OK;
        });
        uint8_t root[32];
        tree_exchange(S, SH_FRI_TREE, (uint32_t)d, root, nullptr, nullptr, nullptr, nullptr, chained ? slot_of(d) : nullptr);
        if (S.agreed) return S.agreed;
        if (chained) {
            S.local([&]() -> int {
                k_fri_draw_at(c, c->fri_upper[d], d_alpha + d, d_roots + d);
                k_fri_fold_cm(c, c->fri_e[d], c->fri_e[d + 1], fri_nd(c, d), d, fe_zero(), d_alpha + d);
                c->fri_folded = d + 1;
                return DST_OK;
            });
        } else {
            fe x;
            prng_vector(root, 1, &x);                                 // fri/prover.rs:40 field::prng(root)
            S.local([&] { return dst_shard_fri_fold(c, (const uint8_t*)&x); });
        }
    }
    {
        const int d = rep_from;
        const size_t bytes = c->Bc * fri_nd(c, d) * 16;
        if (bytes * G > c->gather_bytes) { c->err = "dst_prove_sharded: gather buffer too small for the FRI layer"; return DST_ERR_ARG; }        // the same on every rank
        // this rank's cosets of the layer (contiguous, coset-major) -> all cosets in the gather buffer -> natural order, then the rest
        // of the commit phase on every rank (the natural-order layer overwrites fri_e[d] only after the exchange has completed)
        uint8_t root[32];
        if (!S.coll(S.coll_on(4, c->stream, [&] { return comm->all_gather(c->fri_e[d], c->gather_buf, bytes, c->stream); }), "FRI tail")) return S.agreed;
        S.local([&] { c->fri_tail_pending = true; return fri_replicated_tail(c, c->gather_buf, 1, root); });
    }
    if (chained && rep_from > 0) {
        // the deferred records of the sharded layers: everything queued has completed once the stream is idle (a rank that failed locally
        // did not run the tail's own wait)
        const double t_wait = wall_ms_shard();
        { const int rw = ctx_sync(c, "the FRI layers"); if (rw) { S.fail(rw, c->err); if (rw == DST_ERR_COMM) return rw; } }
        c->shard_ms[1] += wall_ms_shard() - t_wait;
        for (int d = 0; d < rep_from && !S.agreed; d++) tree_exchange_finish(S, SH_FRI_TREE, (uint32_t)d, slot_of(d), nullptr, nullptr);
        if (S.agreed) return S.agreed;
    }
    mark(6);
    // step 8 (replicated: every rank grinds the same seed and finds the same first nonce)
    uint64_t nonce = 0;
    std::vector<uint64_t> positions;
    S.local([&]() -> int {
        std::vector<uint8_t> roots;
        for (int d = 0; d < c->num_fri_layers; d++) roots.insert(roots.end(), c->fri_roots[d].begin(), c->fri_roots[d].end());
        uint8_t seed0[32], seed1[32];
        if (!blake3_short(roots.data(), roots.size(), seed0)) { c->err = "too many FRI roots"; return DST_ERR_ARG; }
        int r = dst_pow_grind(c, seed0, c->prm.grinding_factor, seed1, &nonce);
        if (r) return r;
        if (query_positions(seed1, c->N, (uint32_t)c->B, c->prm.num_queries, positions)) { c->err = "could not generate enough query positions"; return DST_ERR_ARG; }
        return DST_OK;
    });
    mark(7);
    // step 9: the lengths of the ranks' opening blobs travel with their status
    std::vector<uint64_t> lens(G, 0);
    size_t mine = 0;
    S.local([&] { return dst_shard_open(c, positions.data(), (uint32_t)positions.size(), nullptr, 0, &mine, lens.data()); });
    {
        struct LenRec { uint64_t len; int64_t rc; } rec{(uint64_t)mine, S.rc};
        std::vector<LenRec> all(G);
        if (!S.coll(S.coll_host([&] { return comm->all_gather_host(&rec, all.data(), sizeof(LenRec), c->stream); }), "openings")) return S.agreed;
        for (size_t g = 0; g < G; g++) {
            if (all[g].rc != DST_OK && S.agreed == DST_OK) { S.agreed = (int)all[g].rc; if (S.rc == DST_OK) c->err = "before the openings: rank " + std::to_string(g) + " reported error " + std::to_string(all[g].rc); }
            lens[g] = all[g].len;            // every rank derives the same plan, so this equals what dst_shard_open computed locally
        }
        if (S.agreed) return S.agreed;
    }
    size_t width = 1;
    for (uint64_t l : lens) if (l > width) width = (size_t)l;
    std::vector<uint8_t> blob(width + 8, 0), all(( width + 8) * G);
    S.local([&] { return dst_shard_open(c, positions.data(), (uint32_t)positions.size(), blob.data(), width, &mine, nullptr); });
    { const int64_t rc64 = S.rc; memcpy(blob.data() + width, &rc64, 8); }                    // the status rides behind the blob
    if (!S.coll(S.coll_host([&] { return comm->all_gather_host(blob.data(), all.data(), width + 8, c->stream); }), "openings")) return S.agreed;
    for (size_t g = 0; g < G && S.agreed == DST_OK; g++) {
        int64_t rc64; memcpy(&rc64, all.data() + g * (width + 8) + width, 8);
        if (rc64 != DST_OK) { S.agreed = (int)rc64; if (S.rc == DST_OK) c->err = "openings: rank " + std::to_string(g) + " reported error " + std::to_string(rc64); }
    }
    if (S.agreed) return S.agreed;
    std::vector<uint8_t> packed;
    for (size_t g = 0; g < G; g++) packed.insert(packed.end(), all.begin() + g * (width + 8), all.begin() + g * (width + 8) + lens[g]);
    // assembly is deterministic host work on identical inputs: it succeeds or fails on every rank alike
    const int rc = dst_shard_assemble(c, positions.data(), (uint32_t)positions.size(), nonce, packed.data(), lens.data(), proof_out, cap, proof_len);
    mark(8);
    S.read_collective_events();                                 // everything queued has completed: the last exchange was waited for
    return rc;
}

// One call for a single-process host that drives all GPUs of the node itself: `world` contexts (rank r of world, one per device, each
// with the trace uploaded), one thread per context, in-process transport.  Every context's proof is identical; the first is returned.
int dst_prove_sharded_local(dst_ctx** ctxs, uint32_t world, const dst_public* pub, uint8_t* proof_out, size_t cap, size_t* proof_len) {
    if (!ctxs || !pub || !proof_len || world == 0 || world > 8) return DST_ERR_ARG;
    // validated BEFORE the rank threads exist: a thread that returned early would leave the others in the first barrier
    for (uint32_t r = 0; r < world; r++) {
        if (!ctxs[r]) return DST_ERR_ARG;
        if (ctxs[r]->prm.world != world || ctxs[r]->prm.rank != r) { ctxs[r]->err = "dst_prove_sharded_local: context " + std::to_string(r) + " was not created as rank " + std::to_string(r) + " of " + std::to_string(world); return DST_ERR_ARG; }
    }
    std::vector<dst_comm*> comms(world, nullptr);
    int rc = dst_comm_init_local(world, comms.data());
    if (rc) return rc;
    std::vector<int> codes(world, DST_OK);
    std::vector<std::vector<uint8_t>> proofs(world, std::vector<uint8_t>(cap));
    std::vector<size_t> lens(world, 0);
    std::vector<std::thread> threads;
    for (uint32_t r = 0; r < world; r++)
        threads.emplace_back([&, r] { codes[r] = dst_prove_sharded(ctxs[r], comms[r], pub, proofs[r].data(), cap, &lens[r]); });
    for (auto& t : threads) t.join();
    for (dst_comm* cm : comms) dst_comm_destroy(cm);
    for (uint32_t r = 0; r < world; r++) if (codes[r]) return codes[r];
    *proof_len = lens[0];
    if (proof_out) memcpy(proof_out, proofs[0].data(), lens[0]);
    return DST_OK;
}

}  // extern "C"
