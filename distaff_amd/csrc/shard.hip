// Coset-sharded (multi-GPU) prover phases of libdistaff_hip.so -- one context per GPU, rank g of G owns the cosets
// [g*B/G, (g+1)*B/G) of every LDE.  The phases mirror stark::prove (/root/reference/src/stark/prover.rs:17-168)
// exactly like the single-GPU entry points of api.hip; what differs is that Merkle trees are finished from all-gathered
// boundary nodes and that the constraint evaluations are all-gathered before the cross-coset inverse transform.  The
// collectives themselves are issued by the host (torch.distributed / RCCL in distaff_amd/sharded.py): this file only
// exports and imports the shards.
#include "ctx.h"
#include "host_util.h"

using namespace dsth;

extern "C" void dst_internal_transition_coefficients(const dst_ctx* c, const fe* draws344, std::vector<fe>& tc);   // api.hip

enum { SH_TRACE_TREE = 0, SH_CONSTRAINT_TREE = 1, SH_FRI_TREE = 2, SH_CEVAL = 3, SH_FRI_LAST = 4 };
enum { RD_TRACE_LEAF = 0, RD_TRACE_NODE = 1, RD_TRACE_UPPER = 2, RD_CEVAL = 3, RD_C_NODE = 4, RD_C_UPPER = 5, RD_FRI_E = 6, RD_FRI_LEAF = 7,
       RD_FRI_NODE = 8, RD_FRI_UPPER = 9, RD_LDE_ROW = 10 };

static int ensure_shard_buffers(dst_ctx* c) {
    if (c->gather_buf) return DST_OK;
    const size_t n = c->n, G = c->prm.world;
    size_t need = 32 * n * G;
    if (384 * n > need) need = 384 * n;
    HIP_TRY(c, hipMalloc((void**)&c->gather_buf, need));
    c->gather_bytes = need;
    HIP_TRY(c, hipMalloc((void**)&c->trace_upper, 2 * n * G * sizeof(digest)));
    HIP_TRY(c, hipMalloc((void**)&c->c_upper, 2 * n * G * sizeof(digest)));
    for (int d = 0; d < c->num_fri_layers; d++) {
        size_t nb = c->fri_size[d] / c->B / 4;             // boundary nodes per rank = rows per coset
        HIP_TRY(c, hipMalloc((void**)&c->fri_upper[d], (2 * nb * G > 2 ? 2 * nb * G : 2) * sizeof(digest)));
    }
    return DST_OK;
}
static size_t fri_nd(const dst_ctx* c, int d) { return c->fri_size[d] / c->B; }     // elements per coset in layer d

static int copy_in(dst_ctx* c, void* dst, const void* src, size_t bytes, int src_is_device) {
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
    HIP_TRY(c, hipSetDevice(c->device));
    int r = ensure_shard_buffers(c);
    if (r) return r;
    k_intt_columns(c, c->trace, c->polys, c->W);
    k_lde_columns(c, c->polys, c->lde, c->W);
    k_trace_leaves(c);
    k_merkle_levels_to(c, c->trace_leaves, c->trace_nodes, c->Bc * c->n, c->n);
    fe last[3];
    for (int i = 0; i < 3; i++) HIP_TRY(c, hipMemcpyAsync(&last[i], c->trace + (size_t)i * c->n + (c->n - 1), 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    c->op_count = (uint64_t)fe_to_u128(last[0]);
    c->program_hash[0] = last[1]; c->program_hash[1] = last[2];
    c->committed = true; c->constraints_done = c->composed = false;
    return DST_OK;
}

// step 3, local part: AIR evaluation on the owned evaluation cosets.  *bad_step = first failing trace step seen by this rank (-1: none)
int dst_shard_eval_constraints(dst_ctx* c, const dst_public* pub, const uint8_t* coeffs, int64_t* bad_step) {
    if (!c || !pub || !coeffs) return DST_ERR_ARG;
    if (!c->committed) { c->err = "dst_shard_eval_constraints: trace not committed"; return DST_ERR_STATE; }
    HIP_TRY(c, hipSetDevice(c->device));
    c->pub = *pub;
    std::vector<fe> draws(344), tc;
    memcpy(draws.data(), coeffs, 344 * 16);
    dst_internal_transition_coefficients(c, draws.data(), tc);
    fe* d_coef = c->scratch + c->scratch_elems - 1024;
    fe* d_tc = d_coef + 344;
    HIP_TRY(c, hipMemcpyAsync(d_coef, draws.data(), 344 * 16, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(d_tc, tc.data(), tc.size() * 16, hipMemcpyHostToDevice, c->stream));
    int r = k_eval_constraints(c, d_coef, d_tc, bad_step);
    if (r == DST_ERR_AIR) c->err = "transition constraints were not satisfied";
    return r;
}

// steps 4-5 after the constraint evaluations of all ranks were imported (SH_CEVAL): combination (replicated), LDE of the owned
// cosets and the local levels of the constraint tree
int dst_shard_combine(dst_ctx* c) {
    if (!c) return DST_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = c->n, D = 8 * n;
    fe* ip = c->cwork; fe* fp = c->cwork + D; fe* tp = c->cwork + 2 * D; fe* work = c->cwork + 3 * D;
    k_intt8_cosets(c, c->ceval, ip, work);
    k_syn_div(c, ip, D, fe_one());
    k_intt8_cosets(c, c->ceval + D, fp, work);
    k_syn_div(c, fp, D, c->x_last);
    k_intt8_cosets(c, c->ceval + 2 * D, tp, work);
    k_syn_div_expanded(c, tp, c->cpoly, D, n, c->x_last);
    k_add(c, c->cpoly, ip, D);
    k_add(c, c->cpoly, fp, D);
    k_lde_fold8(c, c->cpoly, c->cevals);
    k_constraint_level1(c);
    k_merkle_local_levels(c, c->cnodes, c->Bc * n / 4, n);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    c->constraints_done = true; c->composed = false;
    return DST_OK;
}

// step 7: leaves and local tree levels of the current FRI layer (all layers are coset-major in sharded mode)
int dst_shard_fri_layer(dst_ctx* c, int* more) {
    if (!c || !more) return DST_ERR_ARG;
    if (!c->composed) { c->err = "dst_shard_fri_layer: composition not built"; return DST_ERR_STATE; }
    int d = c->fri_committed;
    if (d >= c->num_fri_layers || d != c->fri_folded) { c->err = "dst_shard_fri_layer: fold the previous layer first"; return DST_ERR_STATE; }
    HIP_TRY(c, hipSetDevice(c->device));
    size_t nd = fri_nd(c, d), nb = nd / 4;
    if (nb == 0) { c->err = "dst_shard_fri_layer: layer too small for this blowup (sharded mode needs blowup <= 32)"; return DST_ERR_ARG; }
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
        case SH_CEVAL: *bytes = 3 * (c->Bc / (c->B / 8)) * c->n * 16; return DST_OK;
        case SH_FRI_LAST: *bytes = c->Bc * fri_nd(c, c->num_fri_layers - 1) * 16; return DST_OK;
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
        case SH_CEVAL: src = c->ceval; break;
        case SH_FRI_LAST: src = c->fri_e[c->num_fri_layers - 1]; break;
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
        const size_t Q = c->Bc / (c->B / 8), blk = Q * c->n;        // gathered [G][3][Q][n] -> ceval [3][8][n]
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        for (size_t g = 0; g < G; g++)
            for (size_t v = 0; v < 3; v++)
                HIP_TRY(c, hipMemcpyAsync(c->ceval + (v * 8 + g * Q) * c->n, (const fe*)c->gather_buf + (g * 3 + v) * blk, blk * 16, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
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
    HIP_TRY(c, hipStreamSynchronize(c->stream));
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
        case RD_FRI_E: if ((int)arg >= c->num_fri_layers) return DST_ERR_ARG; src = c->fri_e[arg]; item = 16; break;
        case RD_FRI_LEAF: if ((int)arg >= c->num_fri_layers) return DST_ERR_ARG; src = c->fri_leaves[arg]; break;
        case RD_FRI_NODE: if ((int)arg >= c->num_fri_layers) return DST_ERR_ARG; src = c->fri_nodes[arg]; break;
        case RD_FRI_UPPER: if ((int)arg >= c->num_fri_layers) return DST_ERR_ARG; src = c->fri_upper[arg]; break;
        case RD_LDE_ROW: break;
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

int dst_shard_info(dst_ctx* c, uint64_t* op_count, uint32_t* num_fri_layers, uint32_t* stack_depth) {
    if (!c) return DST_ERR_ARG;
    if (op_count) *op_count = c->op_count;
    if (num_fri_layers) *num_fri_layers = (uint32_t)c->num_fri_layers;
    if (stack_depth) *stack_depth = (uint32_t)c->stack_depth;
    return DST_OK;
}

}  // extern "C"
