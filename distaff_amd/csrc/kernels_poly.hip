// Polynomial helpers, FRI folding, proof-of-work and gather kernels for gfx950.
//
// Replaces, with parallel formulations that give the same field elements:
//   polynom::syn_div_in_place (/root/reference/src/math/polynom.rs:190-197)      -> scaled suffix scan
//   polynom::syn_div_expanded_in_place (polynom.rs:202-236)                       -> strided suffix sums + one axpy
//   polynom::eval (polynom.rs:9-17, Horner)                                       -> power-table product + tree reduction
//   parallel::mul_acc / add_in_place (src/math/parallel.rs:132,37)                -> axpy / add / linear-combination kernels
//   fri::reduce's quartic::interpolate_batch + evaluate_batch (src/stark/fri/prover.rs:23-31, src/math/quartic.rs:20,37)
//                                                                                 -> closed-form 4-point fold
//   utils::find_pow_nonce (src/stark/utils/proof_of_work.rs:4-32)                 -> batched nonce search with atomicMin
#include <algorithm>
#include "ctx.h"
#include "blake3_dev.h"

#define PT 256

// ---- power tables of an arbitrary base: lo[t] = b^t (t < 2^lb), hi[h] = b^(h << lb) ------------------------------------------
__global__ void pow_table_kernel(fe* lo, fe* hi, fe b, uint32_t lb, uint32_t hb) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < (1u << lb)) lo[t] = fe_pow_u64(b, t);
    if (t < (1u << hb)) hi[t] = fe_is_zero(b) ? (t == 0 ? fe_one() : fe_zero()) : fe_pow_u64(b, (uint64_t)t << lb);
    if (t == 0) { lo[0] = fe_one(); hi[0] = fe_one(); }
}
struct PowTab { const fe* lo; const fe* hi; uint32_t lb; };
__device__ __forceinline__ fe ptab(const PowTab& p, uint64_t e) {
    uint32_t l = (uint32_t)e & ((1u << p.lb) - 1u), h = (uint32_t)(e >> p.lb);
    fe v = p.lo[l];
    return h ? fe_mul(v, p.hi[h]) : v;
}
// builds the tables for `b` into scratch memory at `where` (needs 2^lb + 2^hb elements)
static PowTab build_pow_table(dst_ctx* c, fe* where, fe b, size_t max_exp) {
    uint32_t bits = 1; while (((size_t)1 << bits) < max_exp + 1) bits++;
    uint32_t lb = (bits + 1) / 2, hb = bits - lb;
    fe* lo = where; fe* hi = where + ((size_t)1 << lb);
    uint32_t cnt = 1u << lb;
    { KScope ks_(c, "pow_table_kernel", 0.0); hipLaunchKernelGGL(pow_table_kernel, dim3((cnt + PT - 1) / PT), dim3(PT), 0, c->stream, lo, hi, b, lb, hb); }
    PowTab t; t.lo = lo; t.hi = hi; t.lb = lb;
    return t;
}
static size_t pow_table_elems(size_t max_exp) {
    uint32_t bits = 1; while (((size_t)1 << bits) < max_exp + 1) bits++;
    uint32_t lb = (bits + 1) / 2, hb = bits - lb;
    return ((size_t)1 << lb) + ((size_t)1 << hb);
}

// ---- additive exclusive suffix scan: data[i] <- sum_{t > i} data[t] --------------------------------------------------------------
#define SCAN_CHUNK 1024     // elements per workgroup (4 per lane)
__global__ void __launch_bounds__(PT) scan_block_kernel(fe* data, size_t len, fe* block_sums) {
    __shared__ fe sh[PT];
    const size_t base = (size_t)blockIdx.x * SCAN_CHUNK;
    fe v[4], run = fe_zero();
    // lane owns 4 consecutive elements; local exclusive suffix within the lane
#pragma unroll
    for (int e = 3; e >= 0; e--) {
        size_t i = base + (size_t)threadIdx.x * 4 + e;
        fe x = i < len ? data[i] : fe_zero();
        v[e] = run;
        run = fe_add(run, x);
    }
    sh[threadIdx.x] = run;
    __syncthreads();
    // exclusive suffix scan of lane totals (Hillis-Steele on 256 entries)
    fe incl = run;
    for (int off = 1; off < PT; off <<= 1) {
        fe other = (threadIdx.x + off < PT) ? sh[threadIdx.x + off] : fe_zero();
        __syncthreads();
        incl = fe_add(incl, other);
        sh[threadIdx.x] = incl;
        __syncthreads();
    }
    fe excl = (threadIdx.x + 1 < PT) ? sh[threadIdx.x + 1] : fe_zero();
#pragma unroll
    for (int e = 0; e < 4; e++) {
        size_t i = base + (size_t)threadIdx.x * 4 + e;
        if (i < len) data[i] = fe_add(v[e], excl);
    }
    if (threadIdx.x == 0 && block_sums) block_sums[blockIdx.x] = sh[0];
}
__global__ void scan_add_offsets_kernel(fe* data, size_t len, const fe* block_offsets) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    data[i] = fe_add(data[i], block_offsets[i / SCAN_CHUNK]);
}
// scratch must hold ceil(len / 1024) + ceil(len / 1024^2) + ... elements
static void suffix_scan(dst_ctx* c, fe* data, size_t len, fe* scratch) {
    size_t blocks = (len + SCAN_CHUNK - 1) / SCAN_CHUNK;
    if (blocks == 1) {
        { KScope ks_(c, "scan_block_kernel", 32.0 * len); hipLaunchKernelGGL(scan_block_kernel, dim3(1), dim3(PT), 0, c->stream, data, len, (fe*)nullptr); }
        return;
    }
    { KScope ks_(c, "scan_block_kernel", 32.0 * len); hipLaunchKernelGGL(scan_block_kernel, dim3((unsigned)blocks), dim3(PT), 0, c->stream, data, len, scratch); }
    suffix_scan(c, scratch, blocks, scratch + blocks);
    { KScope ks_(c, "scan_add_offsets_kernel", 32.0 * len); hipLaunchKernelGGL(scan_add_offsets_kernel, dim3((unsigned)((len + PT - 1) / PT)), dim3(PT), 0, c->stream, data, len, (const fe*)scratch); }
}

// ---- synthetic division by (x - b): a[i] <- sum_{t > i} a[t] * b^(t - i - 1) ---------------------------------------------------
__global__ void scale_by_powers_kernel(fe* a, size_t len, PowTab p, uint64_t offset) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    uint64_t e = i + offset;
    if (e) a[i] = fe_mul(a[i], ptab(p, e));
}
__global__ void shift_down_kernel(const fe* src, fe* dst, size_t len) {      // b == 0: quotient is a shift
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    dst[i] = i + 1 < len ? src[i + 1] : fe_zero();
}
// Blocked form: chunks of SD_CHUNK elements, one workgroup each.  With E_c = sum_{t in chunk c} a_t b^(t - c*C) the contribution of everything to
// the right of chunk c is R_c = sum_{u > c} E_u (b^C)^(u - c - 1) -- the same division one level up, on 1/2048 of the data -- and inside a
// chunk a lane owns eight consecutive elements (Horner), the lanes are combined by a suffix scan whose step k multiplies by b^(8 * 2^k).  Three
// passes over the data (read, read + write) and no power tables, against eight passes + two tables for scale / scan / scale; every power of b
// is a kernel argument in table-pair form (fe_mul_tw).
#define SD_CHUNK 2048
struct SynDivArgs { fe_tw b; fe_tw step[9]; };       // b, and b^(8 * 2^k) for k = 0 .. 8
// lane value e = sum_j v[j] b^j, then Y_lane = sum_{u >= lane} e_u b^(8 (u - lane)) over the 256 lanes plus a 257th entry `right`
__device__ __forceinline__ fe sd_lane_scan(fe* sh, const SynDivArgs& p, fe e, fe right) {
    sh[threadIdx.x] = e;
    if (threadIdx.x == 0) sh[PT] = right;
    __syncthreads();
    fe y = e;
    for (int k = 0; k < 9; k++) {
        const uint32_t other = threadIdx.x + (1u << k);
        const bool has = other <= PT;
        const fe o = has ? sh[other] : fe_zero();
        __syncthreads();
        if (has) y = fe_add(y, fe_mul_tw(o, p.step[k]));
        sh[threadIdx.x] = y;
        __syncthreads();
    }
    return y;
}
// A lane owns eight CONSECUTIVE coefficients; global memory is touched in lane-consecutive order and the chunk is turned round in LDS
// (one 16-byte pad per eight elements: the lanes of a wavefront then read 144 bytes apart).
__device__ __forceinline__ uint32_t sd_slot(uint32_t i) { return i + (i >> 3); }
#define SD_TILE (SD_CHUNK + SD_CHUNK / 8)
__device__ __forceinline__ void sd_load_chunk(fe* tile, const fe* __restrict__ a, size_t len, fe (&v)[8]) {
    const size_t base = (size_t)blockIdx.x * SD_CHUNK;
#pragma unroll
    for (int r = 0; r < 8; r++) { const uint32_t i = r * PT + threadIdx.x; tile[sd_slot(i)] = base + i < len ? a[base + i] : fe_zero(); }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = tile[sd_slot(threadIdx.x * 8 + j)];
}
__global__ void __launch_bounds__(PT) syn_div_chunk_sums_kernel(const fe* __restrict__ a, size_t len, SynDivArgs p, fe* __restrict__ sums) {
    __shared__ fe tile[SD_TILE];
    __shared__ fe sh[PT + 1];
    fe v[8], e = fe_zero();
    sd_load_chunk(tile, a, len, v);
#pragma unroll
    for (int j = 7; j >= 0; j--) e = fe_add(fe_mul_tw(e, p.b), v[j]);
    const fe y = sd_lane_scan(sh, p, e, fe_zero());
    if (threadIdx.x == 0) sums[blockIdx.x] = y;
}
__global__ void __launch_bounds__(PT) syn_div_chunk_kernel(fe* a, size_t len, SynDivArgs p, const fe* __restrict__ right) {
    __shared__ fe tile[SD_TILE];
    __shared__ fe sh[PT + 1];
    fe v[8], q[8], e = fe_zero();
    sd_load_chunk(tile, a, len, v);
#pragma unroll
    for (int j = 7; j >= 0; j--) { q[j] = e; e = fe_add(fe_mul_tw(e, p.b), v[j]); }   // q[j] = sum_{j' > j in lane} v[j'] b^(j'-j-1);  e = sum_j v[j] b^j
    (void)sd_lane_scan(sh, p, e, right ? right[blockIdx.x] : fe_zero());
    fe x = sh[threadIdx.x + 1];                      // everything to the right of this lane, relative to the lane's end
#pragma unroll
    for (int j = 7; j >= 0; j--) {                   // q_i += b^(7 - j) * x      (a lane reads and writes only its own eight tile slots)
        tile[sd_slot(threadIdx.x * 8 + j)] = fe_add(q[j], x);
        x = fe_mul_tw(x, p.b);
    }
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * SD_CHUNK;
#pragma unroll
    for (int r = 0; r < 8; r++) { const uint32_t i = r * PT + threadIdx.x; if (base + i < len) a[base + i] = tile[sd_slot(i)]; }
}
// The same division written to ANOTHER array with the DEEP composition's epilogue (constraint_poly.rs:39-52 merge_into after
// trace_table.rs:248-260): out[i] = k3 * q_i + [i < tn] k1 * t[i] + [inc <= i < inc + tn] k2 * t[i - inc], q = a / (x - b).
// Replaces: copy of the constraint polynomial, subtraction of C(z) (the constant term enters no quotient coefficient), in-place division,
// zero fill of the composition polynomial and three multiply-adds over 8n coefficients each.
struct SynDivEpilogue { fe_tw k1, k2, k3; const fe* t; size_t tn, inc; };
__global__ void __launch_bounds__(PT) syn_div_chunk_out_kernel(const fe* __restrict__ a, fe* __restrict__ out, size_t len, SynDivArgs p, const fe* __restrict__ right, SynDivEpilogue ep) {
    __shared__ fe tile[SD_TILE];
    __shared__ fe sh[PT + 1];
    fe v[8], q[8], e = fe_zero();
    sd_load_chunk(tile, a, len, v);
#pragma unroll
    for (int j = 7; j >= 0; j--) { q[j] = e; e = fe_add(fe_mul_tw(e, p.b), v[j]); }
    (void)sd_lane_scan(sh, p, e, right ? right[blockIdx.x] : fe_zero());
    fe x = sh[threadIdx.x + 1];
#pragma unroll
    for (int j = 7; j >= 0; j--) {
        tile[sd_slot(threadIdx.x * 8 + j)] = fe_add(q[j], x);
        x = fe_mul_tw(x, p.b);
    }
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * SD_CHUNK;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const uint32_t i = r * PT + threadIdx.x;
        const size_t gi = base + i;
        if (gi < len) {
            fe w = fe_mul_tw(tile[sd_slot(i)], ep.k3);
            if (gi < ep.tn) w = fe_add(w, fe_mul_tw(ep.t[gi], ep.k1));
            if (gi >= ep.inc && gi - ep.inc < ep.tn) w = fe_add(w, fe_mul_tw(ep.t[gi - ep.inc], ep.k2));
            out[gi] = w;
        }
    }
}
static SynDivArgs syn_div_args(fe b) {
    SynDivArgs p;
    p.b = fe_tw_make(b);
    fe s = b;
    for (int i = 0; i < 3; i++) s = fe_mul(s, s);                          // b^8
    for (int k = 0; k < 9; k++) { p.step[k] = fe_tw_make(s); s = fe_mul(s, s); }
    return p;
}
static void syn_div_blocked(dst_ctx* c, fe* a, size_t len, fe b, fe* scratch) {
    const size_t chunks = (len + SD_CHUNK - 1) / SD_CHUNK;
    const SynDivArgs p = syn_div_args(b);
    if (chunks > 1) {
        { KScope ks_(c, "syn_div_chunk_sums_kernel", 16.0 * len); hipLaunchKernelGGL(syn_div_chunk_sums_kernel, dim3((unsigned)chunks), dim3(PT), 0, c->stream, (const fe*)a, len, p, scratch); }
        fe bc = b;
        for (int i = 0; i < 11; i++) bc = fe_mul(bc, bc);                   // b^2048
        syn_div_blocked(c, scratch, chunks, bc, scratch + chunks);
    }
    { KScope ks_(c, "syn_div_chunk_kernel", 32.0 * len); hipLaunchKernelGGL(syn_div_chunk_kernel, dim3((unsigned)chunks), dim3(PT), 0, c->stream, a, len, p, chunks > 1 ? (const fe*)scratch : (const fe*)nullptr); }
}
// out = k3 * (a / (x - b)) + k1 * t + k2 * x^inc * t  (t of tn coefficients); `a` is left untouched
void k_syn_div_compose(dst_ctx* c, const fe* a, fe* out, size_t len, fe b, const fe* t, size_t tn, size_t inc, fe k1, fe k2, fe k3) {
    fe* scr = c->scratch;
    const size_t chunks = (len + SD_CHUNK - 1) / SD_CHUNK;
    const SynDivArgs p = syn_div_args(b);
    if (chunks > 1) {
        { KScope ks_(c, "syn_div_chunk_sums_kernel", 16.0 * len); hipLaunchKernelGGL(syn_div_chunk_sums_kernel, dim3((unsigned)chunks), dim3(PT), 0, c->stream, a, len, p, scr); }
        fe bc = b;
        for (int i = 0; i < 11; i++) bc = fe_mul(bc, bc);                   // b^2048
        syn_div_blocked(c, scr, chunks, bc, scr + chunks);
    }
    SynDivEpilogue ep{fe_tw_make(k1), fe_tw_make(k2), fe_tw_make(k3), t, tn, inc};
    { KScope ks_(c, "syn_div_chunk_out_kernel", 32.0 * len); hipLaunchKernelGGL(syn_div_chunk_out_kernel, dim3((unsigned)chunks), dim3(PT), 0, c->stream, a, out, len, p, chunks > 1 ? (const fe*)scr : (const fe*)nullptr, ep); }
}
// ---- several divisions / evaluations in one set of launches -------------------------------------------------------------------------
// Up to four arrays of the same length, each with its own divisor (x - b_k), through the blocked division above: blockIdx.y selects the array.
// (A division of 2^20 coefficients is three launches of a few hundred workgroups; the boundary quotients need four of them and the DEEP
// composition two -- batched they are three launches of four times the width.)  b = 1 (plain suffix sums) goes the same way.
struct SynDivBatch { fe* a[4]; fe* sums[4]; SynDivArgs p[4]; };
__global__ void __launch_bounds__(PT) syn_div_sums_batch_kernel(SynDivBatch B, size_t len) {
    __shared__ fe tile[SD_TILE];
    __shared__ fe sh[PT + 1];
    const SynDivArgs& p = B.p[blockIdx.y];
    fe v[8], e = fe_zero();
    sd_load_chunk(tile, B.a[blockIdx.y], len, v);
#pragma unroll
    for (int j = 7; j >= 0; j--) e = fe_add(fe_mul_tw(e, p.b), v[j]);
    const fe y = sd_lane_scan(sh, p, e, fe_zero());
    if (threadIdx.x == 0) B.sums[blockIdx.y][blockIdx.x] = y;
}
__global__ void __launch_bounds__(PT) syn_div_chunk_batch_kernel(SynDivBatch B, size_t len, int use_right) {
    __shared__ fe tile[SD_TILE];
    __shared__ fe sh[PT + 1];
    const SynDivArgs& p = B.p[blockIdx.y];
    fe* a = B.a[blockIdx.y];
    fe v[8], q[8], e = fe_zero();
    sd_load_chunk(tile, a, len, v);
#pragma unroll
    for (int j = 7; j >= 0; j--) { q[j] = e; e = fe_add(fe_mul_tw(e, p.b), v[j]); }
    (void)sd_lane_scan(sh, p, e, use_right ? B.sums[blockIdx.y][blockIdx.x] : fe_zero());
    fe x = sh[threadIdx.x + 1];
#pragma unroll
    for (int j = 7; j >= 0; j--) {
        tile[sd_slot(threadIdx.x * 8 + j)] = fe_add(q[j], x);
        x = fe_mul_tw(x, p.b);
    }
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * SD_CHUNK;
#pragma unroll
    for (int r = 0; r < 8; r++) { const uint32_t i = r * PT + threadIdx.x; if (base + i < len) a[base + i] = tile[sd_slot(i)]; }
}
// a[k] <- a[k] / (x - b[k]) in place, k < count <= 4, all of `len` coefficients; scratch: count * (chunks + chunks' + ..) elements
static void syn_div_batch_rec(dst_ctx* c, fe* const* a, const fe* b, int count, size_t len, fe* scratch) {
    const size_t chunks = (len + SD_CHUNK - 1) / SD_CHUNK;
    SynDivBatch B{};
    for (int k = 0; k < count; k++) { B.a[k] = a[k]; B.sums[k] = scratch + (size_t)k * chunks; B.p[k] = syn_div_args(b[k]); }
    if (chunks > 1) {
        { KScope ks_(c, "syn_div_sums_batch_kernel", 16.0 * len * count); hipLaunchKernelGGL(syn_div_sums_batch_kernel, dim3((unsigned)chunks, (unsigned)count), dim3(PT), 0, c->stream, B, len); }
        fe bc[4]; fe* sub[4];
        for (int k = 0; k < count; k++) { bc[k] = b[k]; for (int i = 0; i < 11; i++) bc[k] = fe_mul(bc[k], bc[k]); sub[k] = B.sums[k]; }      // b^2048
        syn_div_batch_rec(c, sub, bc, count, chunks, scratch + (size_t)count * chunks);
    }
    { KScope ks_(c, "syn_div_chunk_batch_kernel", 32.0 * len * count); hipLaunchKernelGGL(syn_div_chunk_batch_kernel, dim3((unsigned)chunks, (unsigned)count), dim3(PT), 0, c->stream, B, len, chunks > 1 ? 1 : 0); }
}
void k_syn_div_batch(dst_ctx* c, fe* const* a, const fe* b, int count, size_t len) {
    for (int k = 0; k < count; k++) if (fe_is_zero(b[k]) || c->sw("DISTAFF_SYN_DIV_TABLES")) { for (int q = 0; q < count; q++) k_syn_div(c, a[q], len, b[q]); return; }   // the special forms keep their own path
    syn_div_batch_rec(c, a, b, count, len, c->scratch);
}

void k_syn_div(dst_ctx* c, fe* a, size_t len, fe b) {
    unsigned g = (unsigned)((len + PT - 1) / PT);
    fe* scr = c->scratch;
    if (fe_is_zero(b)) {
        { KScope ks_(c, "shift_down_kernel", 32.0 * len); hipLaunchKernelGGL(shift_down_kernel, dim3(g), dim3(PT), 0, c->stream, (const fe*)a, scr, len); }
        hipMemcpyAsync(a, scr, len * sizeof(fe), hipMemcpyDeviceToDevice, c->stream);
        return;
    }
    if (fe_eq(b, fe_one())) { suffix_scan(c, a, len, scr); return; }     // division by (x - 1): q_i = sum_{t > i} a_t, the exclusive scan itself (first-step boundary polynomial)
    if (!c->sw("DISTAFF_SYN_DIV_TABLES")) { syn_div_blocked(c, a, len, b, scr); return; }
    // the first formulation, kept as an independent statement (tests run both): scale by b^t, additive suffix scan, scale by b^-(i+1)
    size_t te = pow_table_elems(len + 1);
    PowTab fw = build_pow_table(c, scr, b, len + 1);
    PowTab bw = build_pow_table(c, scr + te, fe_inv(b), len + 1);
    { KScope ks_(c, "scale_by_powers_kernel", 32.0 * len); hipLaunchKernelGGL(scale_by_powers_kernel, dim3(g), dim3(PT), 0, c->stream, a, len, fw, (uint64_t)0); }
    suffix_scan(c, a, len, scr + 2 * te);
    { KScope ks_(c, "scale_by_powers_kernel", 32.0 * len); hipLaunchKernelGGL(scale_by_powers_kernel, dim3(g), dim3(PT), 0, c->stream, a, len, bw, (uint64_t)1); }
}

// ---- division by (x^degree - 1) / (x - e) (polynom.rs:202-236) -------------------------------------------------------------------
// r[i] = sum_{s >= 0} a[i + s*degree]; out[i] = r[degree + i - 1] - e * r[degree + i] for i <= len - degree, 0 above
__global__ void syn_div_expanded_kernel(const fe* __restrict__ a, fe* __restrict__ out, size_t len, size_t degree, fe e) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    if (i > len - degree) { out[i] = fe_zero(); return; }
    // r[t] for t = degree + i - 1 and t = degree + i (r[len] = 0)
    fe r0 = fe_zero(), r1 = fe_zero();
    for (size_t t = degree + i - 1; t < len; t += degree) r0 = fe_add(r0, a[t]);
    for (size_t t = degree + i; t < len; t += degree) r1 = fe_add(r1, a[t]);
    out[i] = fe_sub(r0, fe_mul(e, r1));
}
void k_syn_div_expanded(dst_ctx* c, const fe* a, fe* out, size_t len, size_t degree, fe exception) {
    { KScope ks_(c, "syn_div_expanded_kernel", 32.0 * len); hipLaunchKernelGGL(syn_div_expanded_kernel, dim3((unsigned)((len + PT - 1) / PT)), dim3(PT), 0, c->stream, a, out, len, degree, exception); }
}

// ---- evaluation of `ncols` polynomials of `len` coefficients at x -----------------------------------------------------------------
__global__ void __launch_bounds__(PT) horner_partial_kernel(const fe* __restrict__ polys, size_t len, PowTab p, fe* __restrict__ partial) {
    __shared__ fe sh[PT];
    const fe* poly = polys + (size_t)blockIdx.y * len;
    fe acc = fe_zero();
    for (size_t i = (size_t)blockIdx.x * PT + threadIdx.x; i < len; i += (size_t)gridDim.x * PT) {
        fe v = poly[i];
        acc = fe_add(acc, i ? fe_mul(v, ptab(p, i)) : v);
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int off = PT / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] = fe_add(sh[threadIdx.x], sh[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = sh[0];
}
__global__ void __launch_bounds__(PT) reduce_rows_kernel(const fe* __restrict__ partial, size_t per_row, fe* __restrict__ out) {
    __shared__ fe sh[PT];
    fe acc = fe_zero();
    for (size_t i = threadIdx.x; i < per_row; i += PT) acc = fe_add(acc, partial[(size_t)blockIdx.x * per_row + i]);
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int off = PT / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] = fe_add(sh[threadIdx.x], sh[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = sh[0];
}
// results land in device memory `out_dev[ncols]`
void k_horner(dst_ctx* c, const fe* polys, size_t ncols, size_t len, fe x, fe* out_dev) {
    fe* scr = c->scratch;
    size_t te = pow_table_elems(len);
    PowTab p = build_pow_table(c, scr, x, len);
    size_t blocks = (len + PT * 8 - 1) / (PT * 8);
    if (blocks > 512) blocks = 512;
    fe* partial = scr + te;
    { KScope ks_(c, "horner_partial_kernel", 16.0 * len * ncols); hipLaunchKernelGGL(horner_partial_kernel, dim3((unsigned)blocks, (unsigned)ncols), dim3(PT), 0, c->stream, polys, len, p, partial); }
    { KScope ks_(c, "reduce_rows_kernel", 0.0); hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)ncols), dim3(PT), 0, c->stream, (const fe*)partial, blocks, out_dev); }
}

// ---- out[i] = sum_k coeffs[k] * cols[k][i] ------------------------------------------------------------------------------------------
// NOUT linear combinations of the same columns in one pass over them: out[q][i] = sum_k cols[k][i] * coeffs[q][k], each a sum of
// products with one reduction (fe_acc); coeffs is [NOUT][ncols], outputs are given as pointers (they may be unrelated arrays)
struct LincombOut { fe* p[4]; };
template <int NOUT>
__global__ void lincomb_kernel(const fe* __restrict__ cols, size_t ncols, size_t len, const fe* __restrict__ coeffs, size_t coef_stride, LincombOut out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    fe_acc acc[NOUT];
#pragma unroll
    for (int q = 0; q < NOUT; q++) fe_acc_zero(acc[q]);
    for (size_t k = 0; k < ncols; k++) {
        const fe v = cols[k * len + i];
#pragma unroll
        for (int q = 0; q < NOUT; q++) fe_acc_mac(acc[q], v, coeffs[q * coef_stride + k]);
    }
#pragma unroll
    for (int q = 0; q < NOUT; q++) out.p[q][i] = fe_acc_reduce(acc[q]);
}
void k_lincomb(dst_ctx* c, const fe* cols, size_t ncols, size_t len, const fe* coeffs_dev, fe* out) {
    LincombOut o{{out, nullptr, nullptr, nullptr}};
    { KScope ks_(c, "lincomb_kernel", 16.0 * len * (ncols + 1)); hipLaunchKernelGGL(lincomb_kernel<1>, dim3((unsigned)((len + PT - 1) / PT)), dim3(PT), 0, c->stream, cols, ncols, len, coeffs_dev, ncols, o); }
}
void k_lincomb4(dst_ctx* c, const fe* cols, size_t ncols, size_t len, const fe* coeffs_dev, fe* out0, fe* out1, fe* out2, fe* out3) {
    LincombOut o{{out0, out1, out2, out3}};
    { KScope ks_(c, "lincomb_kernel", 16.0 * len * (ncols + 4)); hipLaunchKernelGGL(lincomb_kernel<4>, dim3((unsigned)((len + PT - 1) / PT)), dim3(PT), 0, c->stream, cols, ncols, len, coeffs_dev, ncols, o); }
}
// two combinations whose coefficient vectors are coef_stride elements apart
void k_lincomb2(dst_ctx* c, const fe* cols, size_t ncols, size_t len, const fe* coeffs_dev, size_t coef_stride, fe* out0, fe* out1) {
    LincombOut o{{out0, out1, nullptr, nullptr}};
    { KScope ks_(c, "lincomb_kernel", 16.0 * len * (ncols + 2)); hipLaunchKernelGGL(lincomb_kernel<2>, dim3((unsigned)((len + PT - 1) / PT)), dim3(PT), 0, c->stream, cols, ncols, len, coeffs_dev, coef_stride, o); }
}
__global__ void axpy_kernel(fe* y, const fe* x, fe a, size_t len) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < len) y[i] = fe_add(y[i], fe_mul(x[i], a));
}
void k_axpy(dst_ctx* c, fe* y, const fe* x, fe a, size_t len) {
    { KScope ks_(c, "axpy_kernel", 48.0 * len); hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)((len + PT - 1) / PT)), dim3(PT), 0, c->stream, y, x, a, len); }
}
__global__ void add_kernel(fe* y, const fe* x, size_t len) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < len) y[i] = fe_add(y[i], x[i]);
}
void k_add(dst_ctx* c, fe* y, const fe* x, size_t len) {
    { KScope ks_(c, "add_kernel", 48.0 * len); hipLaunchKernelGGL(add_kernel, dim3((unsigned)((len + PT - 1) / PT)), dim3(PT), 0, c->stream, y, x, len); }
}
// y[0] -= sum_k coeffs[k] * values[k]   (the constant terms T_k(z) * cc_k of trace_table.rs:226-233)
__global__ void sub_dot_at0_kernel(fe* y, const fe* values, const fe* coeffs, size_t count) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        fe acc = y[0];
        for (size_t k = 0; k < count; k++) acc = fe_sub(acc, fe_mul(values[k], coeffs[k]));
        y[0] = acc;
    }
}
void k_sub_dot_at0(dst_ctx* c, fe* y, const fe* values_dev, const fe* coeffs_dev, size_t count) {
    { KScope ks_(c, "sub_dot_at0_kernel", 0.0); hipLaunchKernelGGL(sub_dot_at0_kernel, dim3(1), dim3(64), 0, c->stream, y, values_dev, coeffs_dev, count); }
}
__global__ void sub_at0_kernel(fe* y, const fe* v) { if (threadIdx.x == 0 && blockIdx.x == 0) y[0] = fe_sub(y[0], v[0]); }
void k_sub_at0(dst_ctx* c, fe* y, const fe* v_dev) { { KScope ks_(c, "sub_at0_kernel", 0.0); hipLaunchKernelGGL(sub_at0_kernel, dim3(1), dim3(64), 0, c->stream, y, v_dev); } }

// ---- FRI fold -----------------------------------------------------------------------------------------------------------------------
// row r of a layer of size M (R = M/4 rows) sits on xs = x * {1, i, -1, -i} with x = w_N^(r * stride) and i = w_N^(N/4);
// the degree-3 interpolant evaluated at alpha is (1/4) * sum_j (alpha / x)^j * sum_k y_k i^(-jk).
struct FoldArgs { const fe* itw_lo; const fe* itw_hi; uint32_t lo_bits; uint32_t log_N; fe alpha, iota, quarter;
                  const fe* alpha_dev; };      // not null: alpha is read from device memory (drawn there from the layer's root, fri_draw_kernel)
__device__ __forceinline__ fe fold_row(const FoldArgs& a, const fe& y0, const fe& y1, const fe& y2, const fe& y3, uint64_t exp_x) {
    uint64_t e = exp_x & (((uint64_t)1 << a.log_N) - 1);
    uint32_t l = (uint32_t)e & ((1u << a.lo_bits) - 1u), h = (uint32_t)(e >> a.lo_bits);
    fe xinv = a.itw_lo[l];
    if (h) xinv = fe_mul(xinv, a.itw_hi[h]);
    fe t = fe_mul(a.alpha, xinv);
    fe s02 = fe_add(y0, y2), s13 = fe_add(y1, y3), d02 = fe_sub(y0, y2);
    fe id13 = fe_mul(a.iota, fe_sub(y1, y3));
    fe u0 = fe_add(s02, s13), u2 = fe_sub(s02, s13), u1 = fe_sub(d02, id13), u3 = fe_add(d02, id13);
    fe t2 = fe_sqr(t), t3 = fe_mul(t2, t);
    fe acc = fe_add(fe_add(u0, fe_mul(t, u1)), fe_add(fe_mul(t2, u2), fe_mul(t3, u3)));
    return fe_mul(acc, a.quarter);
}
// layer 0 -> 1: coset-major input comp[Bc][n]; r = B*k + j, k < n/4; output natural order via an LDS tile transpose
__global__ void __launch_bounds__(PT) fri_fold0_kernel(const fe* __restrict__ comp, fe* __restrict__ out, size_t n, uint32_t Bc, uint32_t log_b,
                                                      uint32_t log_jt, uint32_t j0, FoldArgs a) {
    __shared__ fe tile[PT];
    if (a.alpha_dev) a.alpha = *a.alpha_dev;
    const uint32_t JT = 1u << log_jt, KT = blockDim.x >> log_jt;          // the launcher shrinks the block for tiny traces
    const uint32_t kk = threadIdx.x % KT, jj = threadIdx.x / KT;
    const size_t k = (size_t)blockIdx.x * KT + kk;
    const uint32_t j = blockIdx.y * JT + jj;
    const fe* base = comp + (size_t)j * n + k;
    const size_t q = n / 4;
    uint64_t r = ((uint64_t)k << log_b) + j0 + j;
    tile[kk * JT + jj] = fold_row(a, base[0], base[q], base[2 * q], base[3 * q], r);
    __syncthreads();
    const uint32_t kk2 = threadIdx.x >> log_jt, jj2 = threadIdx.x & (JT - 1);
    out[((size_t)blockIdx.x * KT + kk2) * Bc + blockIdx.y * JT + jj2] = tile[kk2 * JT + jj2];
}
__global__ void __launch_bounds__(PT) fri_fold_kernel(const fe* __restrict__ e, fe* __restrict__ out, size_t R, uint32_t log_stride, FoldArgs a) {
    size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    if (a.alpha_dev) a.alpha = *a.alpha_dev;
    out[r] = fold_row(a, e[r], e[r + R], e[r + 2 * R], e[r + 3 * R], (uint64_t)r << log_stride);
}
static void fri_fold_launch(dst_ctx* c, int layer, fe special_x, const fe* alpha_dev);
void k_fri_fold(dst_ctx* c, int layer, fe special_x) { fri_fold_launch(c, layer, special_x, nullptr); }
void k_fri_fold_dev(dst_ctx* c, int layer, const fe* alpha_dev) { fri_fold_launch(c, layer, fe_zero(), alpha_dev); }
static void fri_fold_launch(dst_ctx* c, int layer, fe special_x, const fe* alpha_dev) {
    FoldArgs a{};
    a.itw_lo = c->itw_lo; a.itw_hi = c->itw_hi; a.lo_bits = c->tw_lo_bits; a.log_N = c->log_N;
    a.alpha = special_x; a.iota = c->iota; a.quarter = c->four_inv; a.alpha_dev = alpha_dev;
    size_t R = c->fri_size[layer] / 4;
    if (layer == 0) {
        uint32_t jt = c->Bc < 32 ? (uint32_t)c->Bc : 32u, log_jt = 0;
        while ((1u << log_jt) < jt) log_jt++;
        const size_t want = (c->n / 4) << log_jt;
        const uint32_t threads = (uint32_t)(want < PT ? want : PT), KT = threads >> log_jt;      // never an empty grid (tiny traces, many ranks)
        dim3 g((unsigned)((c->n / 4) / KT), (unsigned)(c->Bc >> log_jt));
        { KScope ks_(c, "fri_fold0_kernel", 80.0 * (c->n / 4) * c->Bc); hipLaunchKernelGGL(fri_fold0_kernel, g, dim3(threads), 0, c->stream, (const fe*)c->comp, c->fri_e[1], c->n, (uint32_t)c->Bc, c->log_b, log_jt, (uint32_t)c->j0, a); }
    } else {
        { KScope ks_(c, "fri_fold_kernel", 80.0 * R); hipLaunchKernelGGL(fri_fold_kernel, dim3((unsigned)((R + PT - 1) / PT)), dim3(PT), 0, c->stream, (const fe*)c->fri_e[layer], c->fri_e[layer + 1], R,
                           (uint32_t)(2 * layer), a); }
    }
}

// ---- the small FRI layers in ONE launch -------------------------------------------------------------------------------------------------
// fri::reduce (fri/prover.rs:11-53) from a layer of at most 2^13 evaluations on: per layer hash the rows, build the tree, draw
// x = field::prng(root) (field.rs:264-275: StdRng = ChaCha20 seeded with the root, Uniform over the field), fold with it.  On the host each
// of these layers is four launches and a root read-back; here one workgroup walks through all of them and the roots are read back once.
// The draw is the device statement of host_util.h's StdRng / uniform_field (same word order and rejection rule; the tests compare proofs).
__device__ void fri_chacha_block(const uint32_t key[8], uint64_t counter, uint32_t out[16]) {
    uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                       (uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u};
    uint32_t x[16];
    for (int i = 0; i < 16; i++) x[i] = in[i];
    auto rotl = [](uint32_t v, int n) { return (v << n) | (v >> (32 - n)); };
    auto qr = [&](int a, int b, int c, int d) {
        x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16); x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12);
        x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8);  x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7);
    };
    for (int r = 0; r < 10; r++) {
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
    }
    for (int i = 0; i < 16; i++) out[i] = x[i] + in[i];
}
// Uniform::from(0..p).sample(StdRng::from_seed(root)): v = next u128 (four words, low first); (hi, lo) = v * p; accept hi when lo <= zone,
// zone = MAX - (MAX - p + 1) % p = p - 1 (2^128 - p < p)
__device__ fe fri_prng(const uint32_t root[8]) {
    uint32_t buf[16];
    for (uint64_t counter = 0;; counter++) {
        fri_chacha_block(root, counter, buf);
        for (int w = 0; w < 16; w += 4) {
            const uint32_t v[4] = {buf[w], buf[w + 1], buf[w + 2], buf[w + 3]};
            const uint32_t pl[4] = {FE_P0, FE_P1, FE_P2, FE_P3};
            uint32_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = 0; i < 4; i++) {
                uint64_t carry = 0;
                for (int j = 0; j < 4; j++) { const uint64_t cur = (uint64_t)v[i] * pl[j] + t[i + j] + carry; t[i + j] = (uint32_t)cur; carry = cur >> 32; }
                t[i + 4] = (uint32_t)carry;
            }
            // lo = t[0..3] <= p - 1  <=>  lo < p
            bool below = false, decided = false;
            for (int i = 3; i >= 0 && !decided; i--) { if (t[i] != pl[i]) { below = t[i] < pl[i]; decided = true; } }
            if (decided && below) return fe_make(t[4], t[5], t[6], t[7]);
        }
    }
}
#define FRI_TAIL_MAX_LAYERS 12
#define FRI_TAIL_THREADS 1024
struct FriTailArgs {
    fe* e[FRI_TAIL_MAX_LAYERS]; digest* leaves[FRI_TAIL_MAX_LAYERS]; digest* nodes[FRI_TAIL_MAX_LAYERS];
    uint32_t rows[FRI_TAIL_MAX_LAYERS];            // R = size / 4 of layer first + i
    uint32_t first, count;                         // layers first .. first + count - 1 (the last one is the remainder: not folded)
    uint32_t* roots;                               // [count][8]
    FoldArgs fold;                                 // alpha is drawn per layer
};
__global__ void __launch_bounds__(FRI_TAIL_THREADS) fri_tail_kernel(FriTailArgs a) {
    __shared__ fe alpha_sh;
    const uint32_t tid = threadIdx.x;
    for (uint32_t li = 0; li < a.count; li++) {
        const uint32_t R = a.rows[li];
        const fe* e = a.e[li];
        digest* leaves = a.leaves[li]; digest* nodes = a.nodes[li];
        for (uint32_t r = tid; r < R; r += FRI_TAIL_THREADS) {            // fri/utils.rs:16-22 hash_values
            uint32_t m[16], h[8];
#pragma unroll
            for (uint32_t q = 0; q < 4; q++) { const fe v = e[r + q * R]; m[4 * q] = v.v[0]; m[4 * q + 1] = v.v[1]; m[4 * q + 2] = v.v[2]; m[4 * q + 3] = v.v[3]; }
            b3_hash64(m, h);
#pragma unroll
            for (int i = 0; i < 8; i++) leaves[r].w[i] = h[i];
        }
        __threadfence_block(); __syncthreads();
        for (uint32_t cnt = R >> 1; cnt >= 1; cnt >>= 1) {                 // merkle.rs:269-294: nodes[cnt + i] = H(children 2i, 2i + 1 of the level below)
            const digest* below = (cnt == (R >> 1)) ? leaves : nodes + 2 * cnt;
            for (uint32_t i = tid; i < cnt; i += FRI_TAIL_THREADS) {
                uint32_t m[16], h[8];
#pragma unroll
                for (int w = 0; w < 8; w++) { m[w] = below[2 * i].w[w]; m[8 + w] = below[2 * i + 1].w[w]; }
                b3_hash64(m, h);
#pragma unroll
                for (int w = 0; w < 8; w++) nodes[cnt + i].w[w] = h[w];
            }
            __threadfence_block(); __syncthreads();
        }
        if (tid < 8) { nodes[0].w[tid] = 0; a.roots[li * 8 + tid] = nodes[1].w[tid]; }          // merkle.rs:275
        if (li + 1 == a.count) break;                                      // the remainder layer is committed, not folded
        if (tid == 0) { uint32_t root[8]; for (int i = 0; i < 8; i++) root[i] = nodes[1].w[i]; alpha_sh = fri_prng(root); }      // fri/prover.rs:40
        __syncthreads();
        FoldArgs f = a.fold; f.alpha = alpha_sh;
        fe* out = a.e[li + 1];
        const uint32_t log_stride = 2 * (a.first + li);
        for (uint32_t r = tid; r < R; r += FRI_TAIL_THREADS) out[r] = fold_row(f, e[r], e[r + R], e[r + 2 * R], e[r + 3 * R], (uint64_t)r << log_stride);
        __threadfence_block(); __syncthreads();
    }
}
// x = field::prng(root) of a layer committed by the per-layer kernels, drawn where the root is: the host need not see the root before the
// fold is queued (fri/prover.rs:40).  One lane; also files the root for the read-back at the end of the commit phase.
__global__ void fri_draw_kernel(const digest* __restrict__ nodes, fe* __restrict__ alpha_out, digest* __restrict__ root_out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t root[8];
    for (int i = 0; i < 8; i++) root[i] = nodes[1].w[i];
    *alpha_out = fri_prng(root);
    for (int i = 0; i < 8; i++) root_out->w[i] = root[i];
}
void k_fri_draw_at(dst_ctx* c, const digest* nodes, fe* alpha_out, digest* root_out) {        // nodes[1] = the root
    KScope ks_(c, "fri_draw_kernel", 0.0);
    hipLaunchKernelGGL(fri_draw_kernel, dim3(1), dim3(64), 0, c->stream, nodes, alpha_out, root_out);
}
void k_fri_draw(dst_ctx* c, int layer, fe* alpha_out, digest* root_out) { k_fri_draw_at(c, c->fri_nodes[layer], alpha_out, root_out); }
// commits (and folds) the natural-order layers first .. num_fri_layers - 1 in one launch; roots_out: (num_fri_layers - first) x 32 bytes
int k_fri_tail(dst_ctx* c, int first, uint8_t* roots_out) {
    const int L = c->num_fri_layers, count = L - first;
    if (first < 1 || count < 1 || count > FRI_TAIL_MAX_LAYERS) { c->err = "k_fri_tail: bad layer range"; return DST_ERR_ARG; }
    static_assert(FRI_TAIL_THREADS == 1024, "fri_tail_kernel is ONE workgroup: its barriers are the only ordering between the layers");
    static_assert(8 + 4 * FRI_TAIL_MAX_LAYERS <= 64, "the roots sit behind the first 8 of the 64 words of d_u64");
    for (int i = 0; i < count; i++)
        if (c->fri_size[first + i] / 4 < 2) { c->err = "k_fri_tail: a layer of fewer than two rows has no tree level to build"; return DST_ERR_ARG; }      // root = leaf never occurs: layers hold >= 64 evaluations
    FriTailArgs a{};
    for (int i = 0; i < count; i++) { a.e[i] = c->fri_e[first + i]; a.leaves[i] = c->fri_leaves[first + i]; a.nodes[i] = c->fri_nodes[first + i]; a.rows[i] = (uint32_t)(c->fri_size[first + i] / 4); }
    a.first = (uint32_t)first; a.count = (uint32_t)count;
    a.roots = (uint32_t*)(c->d_u64 + 8);                                   // 12 x 32 bytes behind the small device scalars (d_u64 holds 64 words)
    a.fold.itw_lo = c->itw_lo; a.fold.itw_hi = c->itw_hi; a.fold.lo_bits = c->tw_lo_bits; a.fold.log_N = c->log_N;
    a.fold.alpha = fe_zero(); a.fold.iota = c->iota; a.fold.quarter = c->four_inv;
    { KScope ks_(c, "fri_tail_kernel", 0.0); hipLaunchKernelGGL(fri_tail_kernel, dim3(1), dim3(FRI_TAIL_THREADS), 0, c->stream, a); }
    // read back into page-locked memory (the sharded prover queues this behind its exchanges: a pageable destination would make the host
    // wait inside hipMemcpyAsync, in front of the bounded wait below)
    uint8_t* pinned = c->h_stage + HS_TAIL_ROOTS;
    HIP_TRY(c, hipMemcpyAsync(pinned, a.roots, (size_t)count * 32, hipMemcpyDeviceToHost, c->stream));
    CTX_SYNC(c, "the last FRI layers");
    HIP_TRY(c, hipGetLastError());
    memcpy(roots_out, pinned, (size_t)count * 32);
    return DST_OK;
}

// ---- proof of work ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(PT) pow_kernel(const uint32_t* __restrict__ seed8, uint64_t base, uint32_t grinding, unsigned long long* best) {
    uint64_t nonce = base + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // a smaller nonce has been found by a workgroup that started earlier: nothing this one could contribute (the launch covers 2^22
    // nonces, four times the expected search length at grinding 20; the result is the minimum either way)
    if (__atomic_load_n(best, __ATOMIC_RELAXED) < base + (uint64_t)blockIdx.x * blockDim.x) return;
    uint32_t m[16], h[8];
#pragma unroll
    for (int i = 0; i < 8; i++) m[i] = seed8[i];
    m[8] = (uint32_t)nonce; m[9] = (uint32_t)(nonce >> 32);
#pragma unroll
    for (int i = 10; i < 16; i++) m[i] = 0;
    b3_hash64(m, h);
    uint64_t w = ((uint64_t)h[1] << 32) | h[0];
    uint32_t tz = w == 0 ? 64u : (uint32_t)__ffsll((long long)w) - 1u;
    if (tz >= grinding) atomicMin(best, (unsigned long long)nonce);
}
int k_pow(dst_ctx* c, const uint8_t seed[32], uint32_t grinding, uint64_t* nonce) {
    uint32_t* d_seed = (uint32_t*)(c->d_u64 + 2);
    unsigned long long* d_best = (unsigned long long*)(c->d_u64 + 1);
    HIP_TRY(c, hipMemcpyAsync(d_seed, seed, 32, hipMemcpyHostToDevice, c->stream));
    unsigned long long init = ~0ull;
    HIP_TRY(c, hipMemcpyAsync(d_best, &init, 8, hipMemcpyHostToDevice, c->stream));
    const uint64_t batch = (uint64_t)1 << 22;
    for (uint64_t base = 1;; base += batch) {
        { KScope ks_(c, "pow_kernel", 0.0); hipLaunchKernelGGL(pow_kernel, dim3((unsigned)(batch / PT)), dim3(PT), 0, c->stream, (const uint32_t*)d_seed, base, grinding, d_best); }
        unsigned long long* h_best = reinterpret_cast<unsigned long long*>(c->h_stage + HS_POW);      // page-locked: queued, then waited for with a bound
        HIP_TRY(c, hipMemcpyAsync(h_best, d_best, 8, hipMemcpyDeviceToHost, c->stream));
        CTX_SYNC(c, "the proof-of-work search");
        const unsigned long long best = *h_best;
        if (best != ~0ull) { *nonce = best; return DST_OK; }
        if (base > ((uint64_t)1 << 40)) { c->err = "proof-of-work search exhausted"; return DST_ERR_ARG; }
    }
}

// ---- gathers --------------------------------------------------------------------------------------------------------------------------------
__global__ void gather_kernel(const uint4* __restrict__ src, uint32_t vecs_per_item, const uint64_t* __restrict__ idx, size_t count, uint4* __restrict__ dst) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count * vecs_per_item) return;
    size_t item = t / vecs_per_item, v = t % vecs_per_item;
    dst[t] = src[idx[item] * vecs_per_item + v];
}
void k_gather(dst_ctx* c, const void* src, size_t item_bytes, const uint64_t* idx_dev, size_t count, void* dst) {
    uint32_t vp = (uint32_t)(item_bytes / 16);
    size_t total = count * vp;
    if (!total) return;
    { KScope ks_(c, "gather_kernel", 0.0); hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((total + PT - 1) / PT)), dim3(PT), 0, c->stream, (const uint4*)src, vp, idx_dev, count, (uint4*)dst); }
}
// every 16-byte piece of every opened item in ONE launch: addr[t] is the device address of piece t (the openings touch a dozen trees and
// layers; one gather launch per source buffer was 41 launches of a few microseconds each)
__global__ void gather_pieces_kernel(const uint64_t* __restrict__ addr, size_t count, uint4* __restrict__ dst) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    dst[t] = *reinterpret_cast<const uint4*>(addr[t]);
}
void k_gather_pieces(dst_ctx* c, const uint64_t* addr_dev, size_t count, void* dst) {
    if (!count) return;
    { KScope ks_(c, "gather_pieces_kernel", 0.0); hipLaunchKernelGGL(gather_pieces_kernel, dim3((unsigned)((count + PT - 1) / PT)), dim3(PT), 0, c->stream, addr_dev, count, (uint4*)dst); }
}
// gather of one trace row per position from the coset-major LDE: out[p][c] = lde[c][j][k] with position = B*k + j
__global__ void gather_rows_kernel(const fe* __restrict__ lde, size_t n, uint32_t Bc, uint32_t log_b, uint32_t j0, uint32_t W,
                                   const uint64_t* __restrict__ positions, size_t count, fe* __restrict__ out) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count * W) return;
    size_t p = t / W, col = t % W;
    uint64_t pos = positions[p];
    size_t k = pos >> log_b, j = (pos & ((1u << log_b) - 1)) - j0;
    out[t] = lde[(col * Bc + j) * n + k];
}
void k_gather_rows(dst_ctx* c, const uint64_t* positions_dev, size_t count, fe* out) {
    size_t total = count * c->W;
    { KScope ks_(c, "gather_rows_kernel", 0.0); hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((total + PT - 1) / PT)), dim3(PT), 0, c->stream, (const fe*)c->lde, c->n, (uint32_t)c->Bc, c->log_b,
                       (uint32_t)c->j0, (uint32_t)c->W, positions_dev, count, out); }
}

#if DST_TEST_HOOKS            // calibration kernels: test / bench build only
// ---- mulmod micro-benchmark (bench.py's ALU ceiling) --------------------------------------------------------------------------------------
template <int VARIANT>
__global__ void __launch_bounds__(PT) mulmod_bench_kernel(fe* out, uint32_t iters) {
    auto MUL = [](const fe& x, const fe& y) { return VARIANT == 0 ? fe_mul(x, y) : fe_mul_portable(x, y); };
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    fe a = fe_make(g * 2654435761u + 12345u, g ^ 0x9E3779B9u, g + 77u, 0x12345678u);
    fe b = fe_make(g + 1u, 0xABCDEF01u, g * 3u + 5u, 0x0FEDCBA9u);
    fe c2 = fe_make(0x11111111u + g, 0x22222222u, 0x33333333u, 0x04444444u);
    fe d = fe_make(0x55555555u, 0x66666666u + g, 0x77777777u, 0x08888888u);
    for (uint32_t i = 0; i < iters; i++) {      // four independent dependency chains per lane
        a = MUL(a, b); b = MUL(b, c2); c2 = MUL(c2, d); d = MUL(d, a);
    }
    out[g] = fe_add(fe_add(a, b), fe_add(c2, d));
}
int k_bench_mulmod(dst_ctx* c, uint64_t lanes, uint32_t iters, double* ms) {
    const bool portable = (iters & 0x80000000u) != 0;     // test hook: top bit selects the portable C formulation
    iters &= 0x7FFFFFFFu;
    if (lanes > c->scratch_elems || lanes < PT) { c->err = "dst_bench_mulmod: lanes must be in [256, 2^21]"; return DST_ERR_ARG; }
    lanes = lanes / PT * PT;
    hipEvent_t e0, e1;
    HIP_TRY(c, hipEventCreate(&e0)); HIP_TRY(c, hipEventCreate(&e1));
    { KScope ks_(c, "mulmod_bench_kernel", 0.0); hipLaunchKernelGGL(mulmod_bench_kernel<0>, dim3((unsigned)(lanes / PT)), dim3(PT), 0, c->stream, c->scratch, 4u); }   // warm-up
    HIP_TRY(c, hipEventRecord(e0, c->stream));
    if (portable) { KScope ks_(c, "mulmod_bench_kernel", 0.0); hipLaunchKernelGGL(mulmod_bench_kernel<1>, dim3((unsigned)(lanes / PT)), dim3(PT), 0, c->stream, c->scratch, iters); }
    else { KScope ks_(c, "mulmod_bench_kernel", 0.0); hipLaunchKernelGGL(mulmod_bench_kernel<0>, dim3((unsigned)(lanes / PT)), dim3(PT), 0, c->stream, c->scratch, iters); }
    HIP_TRY(c, hipEventRecord(e1, c->stream));
    HIP_TRY(c, hipEventSynchronize(e1));
    float f = 0; HIP_TRY(c, hipEventElapsedTime(&f, e0, e1));
    *ms = f;
    hipEventDestroy(e0); hipEventDestroy(e1);
    return DST_OK;
}

// ---- the shader clock under a field-arithmetic load: the same four multiplication chains per lane as mulmod_bench_kernel on every SIMD of the
// device, each wavefront noting s_memtime (shader cycles) and s_memrealtime (constant 100 MHz) around its loop.  The power management does not
// hold the nominal 2.4 GHz under this path's arithmetic (profiles/r6_power_clock.md): bench.py reports what the box of the run sustains.
__global__ void __launch_bounds__(PT) clock_probe_kernel(fe* out, unsigned long long* rec, uint32_t iters) {
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    fe a = fe_make(g * 2654435761u + 12345u, g ^ 0x9E3779B9u, g + 77u, 0x12345678u);
    fe b = fe_make(g + 1u, 0xABCDEF01u, g * 3u + 5u, 0x0FEDCBA9u);
    fe c2 = fe_make(0x11111111u + g, 0x22222222u, 0x33333333u, 0x04444444u);
    fe d = fe_make(0x55555555u, 0x66666666u + g, 0x77777777u, 0x08888888u);
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long r0 = wall_clock64(), t0 = __builtin_readcyclecounter();
#else
    const unsigned long long r0 = 0, t0 = 0;
#endif
    for (uint32_t i = 0; i < iters; i++) { a = fe_mul(a, b); b = fe_mul(b, c2); c2 = fe_mul(c2, d); d = fe_mul(d, a); }
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
#else
    const unsigned long long t1 = 2400, r1 = 100;
#endif
    out[g] = fe_add(fe_add(a, b), fe_add(c2, d));
    if ((threadIdx.x & 63u) == 0u) { rec[2 * (g >> 6)] = t1 - t0; rec[2 * (g >> 6) + 1] = r1 - r0; }
}
int k_bench_clock(dst_ctx* c, uint64_t lanes, uint32_t iters, double* mhz) {
    if (lanes * 2 > c->scratch_elems || lanes < PT) { c->err = "dst_bench_clock: lanes must be in [256, 2^20]"; return DST_ERR_ARG; }
    lanes = lanes / PT * PT;
    unsigned long long* rec = reinterpret_cast<unsigned long long*>(c->scratch + lanes);
    hipLaunchKernelGGL(clock_probe_kernel, dim3((unsigned)(lanes / PT)), dim3(PT), 0, c->stream, c->scratch, rec, 4u);          // warm-up
    hipLaunchKernelGGL(clock_probe_kernel, dim3((unsigned)(lanes / PT)), dim3(PT), 0, c->stream, c->scratch, rec, iters);
    std::vector<unsigned long long> h(2 * (lanes / 64));
    HIP_TRY(c, hipMemcpyAsync(h.data(), rec, h.size() * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    std::vector<double> r;
    for (size_t w = 0; w < lanes / 64; w++) if (h[2 * w + 1]) r.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 100.0);      // ticks per 10 ns -> MHz
    if (r.empty()) { c->err = "dst_bench_clock: no wave reported"; return DST_ERR_HIP; }
    std::sort(r.begin(), r.end());
    *mhz = r[r.size() / 2];
    return DST_OK;
}

// ---- peak of the 32x32+64 multiply-add (v_mad_u64_u32): eight accumulators per lane, 32 mads per iteration, nothing else in the loop.
// The integer-multiplier roofline of the path: every modular multiplication is 18 or 21 of these plus carry handling.
#define MAD_PEAK_PER_ITER 32
__global__ void __launch_bounds__(PT) mad_peak_kernel(uint64_t* out, uint32_t iters) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t a0 = tid, a1 = tid + 1, a2 = tid + 2, a3 = tid + 3, a4 = tid + 4, a5 = tid + 5, a6 = tid + 6, a7 = tid + 7;
    const uint32_t x = tid * 2654435761u + 12345u, y = tid ^ 0xDEADBEEFu;
    for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < MAD_PEAK_PER_ITER / 8; u++) {
            a0 = (uint64_t)x * (uint32_t)a7 + a0; a1 = (uint64_t)y * (uint32_t)a0 + a1; a2 = (uint64_t)x * (uint32_t)a1 + a2; a3 = (uint64_t)y * (uint32_t)a2 + a3;
            a4 = (uint64_t)x * (uint32_t)a3 + a4; a5 = (uint64_t)y * (uint32_t)a4 + a5; a6 = (uint64_t)x * (uint32_t)a5 + a6; a7 = (uint64_t)y * (uint32_t)a6 + a7;
        }
    }
    out[tid] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
int k_bench_mad(dst_ctx* c, uint64_t lanes, uint32_t iters, double* ms) {
    if (lanes > c->scratch_elems * 2 || lanes < PT) { c->err = "dst_bench_mad: lanes must be in [256, 2^22]"; return DST_ERR_ARG; }
    lanes = lanes / PT * PT;
    hipEvent_t e0, e1;
    HIP_TRY(c, hipEventCreate(&e0)); HIP_TRY(c, hipEventCreate(&e1));
    hipLaunchKernelGGL(mad_peak_kernel, dim3((unsigned)(lanes / PT)), dim3(PT), 0, c->stream, (uint64_t*)c->scratch, 4u);
    HIP_TRY(c, hipEventRecord(e0, c->stream));
    hipLaunchKernelGGL(mad_peak_kernel, dim3((unsigned)(lanes / PT)), dim3(PT), 0, c->stream, (uint64_t*)c->scratch, iters);
    HIP_TRY(c, hipEventRecord(e1, c->stream));
    HIP_TRY(c, hipEventSynchronize(e1));
    float f = 0; HIP_TRY(c, hipEventElapsedTime(&f, e0, e1));
    *ms = f;
    hipEventDestroy(e0); hipEventDestroy(e1);
    return DST_OK;
}

#endif  // DST_TEST_HOOKS

// ---- element-wise field operations on caller data (test hook behind dst_field_op) ---------------------------------------------------------
__global__ void field_op_kernel(int op, const fe* a, const fe* b, fe* out, size_t count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    fe x = a[i], y = b[i], r;
    switch (op) {
        case 0: r = fe_add(x, y); break;
        case 1: r = fe_sub(x, y); break;
        case 2: r = fe_mul(x, y); break;
        case 3: r = fe_mul_portable(x, y); break;
        case 4: r = fe_inv(x); break;
        case 5: r = fe_pow(x, y); break;
        case 6: {   // sum_{j<40} a[(i + j) % count] * b[(i + 7j) % count] + a[i], one reduction (fe_acc)
            fe_acc A; fe_acc_zero(A);
            for (size_t j = 0; j < 40; j++) fe_acc_mac(A, a[(i + j) % count], b[(i + 7 * j) % count]);
            fe_acc_add(A, x);
            r = fe_acc_reduce(A);
            break;
        }
        default: r = fe_zero();
    }
    out[i] = r;
}
int k_field_op(dst_ctx* c, int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t count) {
    if (3 * count > c->scratch_elems - 2048) { c->err = "dst_field_op: too many elements"; return DST_ERR_ARG; }
    fe* da = c->scratch; fe* db = da + count; fe* dout = db + count;
    HIP_TRY(c, hipMemcpyAsync(da, a, count * 16, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(db, b, count * 16, hipMemcpyHostToDevice, c->stream));
    { KScope ks_(c, "field_op_kernel", 48.0 * count); hipLaunchKernelGGL(field_op_kernel, dim3((unsigned)((count + PT - 1) / PT)), dim3(PT), 0, c->stream, op, (const fe*)da, (const fe*)db, dout, count); }
    HIP_TRY(c, hipMemcpyAsync(out, dout, count * 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DST_OK;
}

// ---- coset-major FRI fold (sharded mode): e[Bc][nd] -> out[Bc][nd/4]; row r = B*k + j0 + jl of the layer with stride 4^layer ------------
__global__ void __launch_bounds__(PT) fri_fold_cm_kernel(const fe* __restrict__ e, fe* __restrict__ out, size_t nd, uint32_t Bc, uint32_t log_b, uint32_t j0,
                                                        uint32_t log_stride, FoldArgs a) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t q = nd / 4;
    if (t >= q * Bc) return;
    if (a.alpha_dev) a.alpha = *a.alpha_dev;
    size_t jl = t / q, k = t % q;
    const fe* base = e + jl * nd + k;
    uint64_t r = ((uint64_t)k << log_b) + j0 + jl;
    out[jl * q + k] = fold_row(a, base[0], base[q], base[2 * q], base[3 * q], r << log_stride);
}
void k_fri_fold_at(dst_ctx* c, const fe* e, fe* out, size_t R, int layer, fe special_x, const fe* alpha_dev) {      // natural-order layer `layer` of 4R evaluations -> R
    FoldArgs a{};
    a.itw_lo = c->itw_lo; a.itw_hi = c->itw_hi; a.lo_bits = c->tw_lo_bits; a.log_N = c->log_N;
    a.alpha = special_x; a.iota = c->iota; a.quarter = c->four_inv; a.alpha_dev = alpha_dev;
    { KScope ks_(c, "fri_fold_kernel", 80.0 * R); hipLaunchKernelGGL(fri_fold_kernel, dim3((unsigned)((R + PT - 1) / PT)), dim3(PT), 0, c->stream, e, out, R, (uint32_t)(2 * layer), a); }
}
void k_fri_fold_cm(dst_ctx* c, const fe* e, fe* out, size_t nd, int layer, fe special_x, const fe* alpha_dev) {
    FoldArgs a{};
    a.itw_lo = c->itw_lo; a.itw_hi = c->itw_hi; a.lo_bits = c->tw_lo_bits; a.log_N = c->log_N;
    a.alpha = special_x; a.iota = c->iota; a.quarter = c->four_inv; a.alpha_dev = alpha_dev;
    size_t total = nd / 4 * c->Bc;
    { KScope ks_(c, "fri_fold_cm_kernel", 80.0 * total); hipLaunchKernelGGL(fri_fold_cm_kernel, dim3((unsigned)((total + PT - 1) / PT)), dim3(PT), 0, c->stream, e, out, nd, (uint32_t)c->Bc, c->log_b,
                       (uint32_t)c->j0, (uint32_t)(2 * layer), a); }
}
// plain 16-byte-vector copy, used to move shards between this library's buffers and device memory owned by another runtime
// instance in the same process (e.g. a torch tensor): the GPU sees one address space, the other runtime's API does not know our pointers
__global__ void copy_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t vecs) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < vecs; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
void k_copy(dst_ctx* c, void* dst, const void* src, size_t bytes) {
    size_t vecs = bytes / 16;
    if (!vecs) return;
    size_t blocks = (vecs + PT - 1) / PT; if (blocks > 8192) blocks = 8192;
    { KScope ks_(c, "copy_kernel", 2.0 * bytes); hipLaunchKernelGGL(copy_kernel, dim3((unsigned)blocks), dim3(PT), 0, c->stream, (uint4*)dst, (const uint4*)src, vecs); }
}
