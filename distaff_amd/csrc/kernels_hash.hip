// BLAKE3 leaf hashing and Merkle tree construction for gfx950.
//
// Replaces TraceTable::build_merkle_tree (/root/reference/src/stark/trace/trace_table.rs:174-185), MerkleTree::new /
// build_merkle_nodes (src/crypto/merkle.rs:25,269-294), evaluations_to_leaves (src/stark/prover.rs:180-187) and
// fri::utils::hash_values (src/stark/fri/utils.rs:16-22).  One lane hashes one leaf / one node; kernels that consume
// coset-major evaluations emit natural-order digests through an LDS tile transpose, so that global reads are
// KT*16-byte runs along k and global writes are JT*32-byte runs along the leaf index.
#include "ctx.h"
#include "blake3_dev.h"

#define HASH_THREADS 256

__device__ __forceinline__ void store_digest(digest* p, const uint32_t* cv) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
    q[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
}

// BLAKE3 of `count` field elements read through `get(i)`; count * 16 <= 2032 bytes (one or two chunks)
template <class Get>
__device__ __forceinline__ void hash_elements(Get get, uint32_t count, uint32_t* out8) {
    uint32_t cv[8], cv1[8];
    const uint32_t first = count < 64u ? count : 64u;       // elements in chunk 0 (1024 bytes = 64 elements)
    const bool two_chunks = count > 64u;
    for (int pass = 0; pass < 2; pass++) {
        const uint32_t base = pass == 0 ? 0u : 64u;
        const uint32_t cnt = pass == 0 ? first : count - 64u;
        uint32_t* c = pass == 0 ? cv : cv1;
        b3_iv(c);
        const uint32_t blocks = (cnt + 3u) / 4u;
        for (uint32_t b = 0; b < blocks; b++) {
            uint32_t m[16];
#pragma unroll
            for (uint32_t e = 0; e < 4; e++) {
                uint32_t i = b * 4u + e;
                fe v = i < cnt ? get(base + i) : fe_zero();
                m[4 * e] = v.v[0]; m[4 * e + 1] = v.v[1]; m[4 * e + 2] = v.v[2]; m[4 * e + 3] = v.v[3];
            }
            uint32_t rem = cnt - b * 4u;
            uint32_t block_len = rem >= 4u ? 64u : rem * 16u;
            uint32_t flags = (b == 0 ? B3_CHUNK_START : 0u) | (b == blocks - 1 ? (B3_CHUNK_END | (two_chunks ? 0u : B3_ROOT)) : 0u);
            b3_compress(c, m, (uint32_t)pass, 0, block_len, flags);
        }
        if (!two_chunks) break;
    }
    if (two_chunks) {
        uint32_t m[16];
#pragma unroll
        for (int i = 0; i < 8; i++) { m[i] = cv[i]; m[8 + i] = cv1[i]; }
        b3_iv(cv);
        b3_compress(cv, m, 0, 0, 64, B3_PARENT | B3_ROOT);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) out8[i] = cv[i];
}

// ---- trace leaves: leaf(B*k + j) = BLAKE3(reg_0 || ... || reg_{W-1}) at that row --------------------------------------------
// block = KT x JT lanes (kk fastest on the read side, jj fastest on the write side); grid = (n / KT, Bc / JT)
__global__ void __launch_bounds__(HASH_THREADS) trace_leaves_kernel(const fe* __restrict__ lde, digest* __restrict__ leaves,
                                                                   uint32_t W, size_t n, uint32_t Bc, uint32_t log_jt) {
    __shared__ digest tile[HASH_THREADS];
    const uint32_t JT = 1u << log_jt, KT = blockDim.x >> log_jt;          // the launcher shrinks the block for tiny traces
    const uint32_t kk = threadIdx.x % KT, jj = threadIdx.x / KT;
    const size_t k = (size_t)blockIdx.x * KT + kk;
    const uint32_t j = blockIdx.y * JT + jj;
    const fe* base = lde + (size_t)j * n + k;
    const size_t col_stride = (size_t)Bc * n;
    uint32_t h[8];
    hash_elements([&](uint32_t c) { return base[(size_t)c * col_stride]; }, W, h);
    store_digest(&tile[kk * JT + jj], h);
    __syncthreads();
    const uint32_t kk2 = threadIdx.x >> log_jt, jj2 = threadIdx.x & (JT - 1);
    const size_t out = ((size_t)blockIdx.x * KT + kk2) * Bc + blockIdx.y * JT + jj2;
    leaves[out] = tile[kk2 * JT + jj2];
}

// block size of the tile-transposing kernels: HASH_THREADS lanes = (HASH_THREADS >> log_t) indices k x 2^log_t cosets, fewer lanes when
// the array has fewer than that many k (tiny traces, or many ranks at a small blowup): the grid never comes out empty
static uint32_t tile_threads(size_t extent, uint32_t log_t) {
    const size_t want = extent << log_t;
    return (uint32_t)(want < HASH_THREADS ? want : HASH_THREADS);
}

void k_trace_leaves(dst_ctx* c) {
    uint32_t jt = c->Bc < 32 ? (uint32_t)c->Bc : 32u, log_jt = 0;
    while ((1u << log_jt) < jt) log_jt++;
    const uint32_t threads = tile_threads(c->n, log_jt), KT = threads >> log_jt;
    dim3 g((unsigned)(c->n / KT), (unsigned)(c->Bc >> log_jt));
    { KScope ks_(c, "trace_leaves_kernel", (16.0 * c->W + 32.0) * c->Bc * c->n, true); hipLaunchKernelGGL(trace_leaves_kernel, g, dim3(threads), 0, c->stream, (const fe*)c->lde, c->trace_leaves, (uint32_t)c->W, c->n, (uint32_t)c->Bc, log_jt); }
}

// ---- generic Merkle levels: out[i] = H(children[2i] || children[2i+1]) -------------------------------------------------------
__global__ void __launch_bounds__(HASH_THREADS) merkle_level_kernel(const digest* __restrict__ children, digest* __restrict__ out, size_t count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint4* p = reinterpret_cast<const uint4*>(children + 2 * i);
    uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    uint32_t m[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
    uint32_t h[8];
    b3_hash64(m, h);
    store_digest(out + i, h);
}

// two levels per launch for the wide levels: a lane turns four consecutive children into their two parents and the grandparent (three
// compressions, every lane busy), so the level in between is written but not read back and a tree needs half as many wide launches
__global__ void __launch_bounds__(HASH_THREADS) merkle_level2_kernel(const digest* __restrict__ children, digest* __restrict__ parents, digest* __restrict__ grand, size_t count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;          // grandparent index, count of them
    if (i >= count) return;
    uint32_t h0[8], h1[8], g[8];
    {
        const uint4* p = reinterpret_cast<const uint4*>(children + 4 * i);
        uint4 a = p[0], b = p[1], c = p[2], d = p[3];
        uint32_t m[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
        b3_hash64(m, h0);
        a = p[4]; b = p[5]; c = p[6]; d = p[7];
        uint32_t m2[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
        b3_hash64(m2, h1);
    }
    store_digest(parents + 2 * i, h0);
    store_digest(parents + 2 * i + 1, h1);
    uint32_t m[16];
#pragma unroll
    for (int w = 0; w < 8; w++) { m[w] = h0[w]; m[8 + w] = h1[w]; }
    b3_hash64(m, g);
    store_digest(grand + i, g);
}
#define MERKLE_LEVEL2_MIN ((size_t)1 << 19)        // grandparents per launch from which the fused form is used (below: the subtree kernel takes over)
// children[0 .. 4 * count) -> nodes[2 * count .. 4 * count) and nodes[count .. 2 * count)   (heap positions of the two levels above)
static bool merkle_two_levels(dst_ctx* c, const digest* children, digest* nodes, size_t count) {
    size_t min_count = MERKLE_LEVEL2_MIN;
    if (const char* e = c->sw("DISTAFF_MERKLE_LEVEL2_LOG")) min_count = (size_t)1 << (atoi(e) < 0 ? 0 : atoi(e) > 40 ? 40 : atoi(e));      // tests: the fused form on small trees
    if (count < min_count || count == 0 || c->sw("DISTAFF_MERKLE_LEVELS")) return false;
    KScope ks_(c, "merkle_level2_kernel", 96.0 * 3 * count);
    hipLaunchKernelGGL(merkle_level2_kernel, dim3((unsigned)((count + HASH_THREADS - 1) / HASH_THREADS)), dim3(HASH_THREADS), 0, c->stream, children, nodes + 2 * count, nodes + count, count);
    return true;
}

// the top of the tree in one workgroup: nodes[count .. 2*count) are already valid, fills nodes[1 .. count)
__global__ void __launch_bounds__(HASH_THREADS) merkle_top_kernel(digest* nodes, uint32_t count) {
    for (uint32_t cnt = count >> 1; cnt >= 1; cnt >>= 1) {
        for (uint32_t i = threadIdx.x; i < cnt; i += HASH_THREADS) {
            const uint4* p = reinterpret_cast<const uint4*>(nodes + 2 * (cnt + i));
            uint4 a = p[0], b = p[1], c = p[2], d = p[3];
            uint32_t m[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
            uint32_t h[8];
            b3_hash64(m, h);
            store_digest(nodes + cnt + i, h);
        }
        __threadfence_block();
        __syncthreads();
    }
    if (threadIdx.x < 8) nodes[0].w[threadIdx.x] = 0;          // merkle.rs:275
}

// nine levels in one launch: workgroup b hashes the 512 nodes nodes[count + 512 b ..) down to ONE node, keeping the intermediate levels
// in LDS and writing every level to its place in the heap.  Replaces nine launch-bound level kernels in the middle of a tree (between
// the wide levels, which are work-bound and keep one launch each, and the single-workgroup top).
__global__ void __launch_bounds__(HASH_THREADS) merkle_subtree_kernel(digest* nodes, size_t count) {
    __shared__ digest lvl[HASH_THREADS];
    const size_t b = blockIdx.x;
    uint32_t h[8];
    {
        const uint4* p = reinterpret_cast<const uint4*>(nodes + count + 512 * b + 2 * threadIdx.x);
        uint4 a = p[0], bb = p[1], cc = p[2], d = p[3];
        uint32_t m[16] = {a.x, a.y, a.z, a.w, bb.x, bb.y, bb.z, bb.w, cc.x, cc.y, cc.z, cc.w, d.x, d.y, d.z, d.w};
        b3_hash64(m, h);
        store_digest(nodes + (count >> 1) + 256 * b + threadIdx.x, h);
        store_digest(&lvl[threadIdx.x], h);
    }
    __syncthreads();
    size_t level = count >> 1;
    for (uint32_t width = 128; width >= 1; width >>= 1) {          // nodes of this workgroup on the level being built
        level >>= 1;
        uint32_t m[16];
        const bool active = threadIdx.x < width;
        if (active) {
            const uint4* p = reinterpret_cast<const uint4*>(&lvl[2 * threadIdx.x]);
            uint4 a = p[0], bb = p[1], cc = p[2], d = p[3];
            m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.w; m[4] = bb.x; m[5] = bb.y; m[6] = bb.z; m[7] = bb.w;
            m[8] = cc.x; m[9] = cc.y; m[10] = cc.z; m[11] = cc.w; m[12] = d.x; m[13] = d.y; m[14] = d.z; m[15] = d.w;
            b3_hash64(m, h);
            store_digest(nodes + level + (size_t)width * b + threadIdx.x, h);
        }
        __syncthreads();                                            // everyone has read its pair
        if (active) store_digest(&lvl[threadIdx.x], h);
        __syncthreads();
    }
}
#define MERKLE_SUBTREE_MAX ((size_t)1 << 19)       // wider levels are work-bound: one launch each

// builds nodes[1 .. count) from an already filled level nodes[count .. 2*count)
static void merkle_upper_levels(dst_ctx* c, digest* nodes, size_t count) {
    while (count > 1024) {
        if (count <= MERKLE_SUBTREE_MAX && count % 512 == 0 && !c->sw("DISTAFF_MERKLE_LEVELS")) {
            { KScope ks_(c, "merkle_subtree_kernel", 64.0 * count); hipLaunchKernelGGL(merkle_subtree_kernel, dim3((unsigned)(count / 512)), dim3(HASH_THREADS), 0, c->stream, nodes, count); }
            count /= 512;
            continue;
        }
        if (merkle_two_levels(c, nodes + count, nodes, count >> 2)) { count >>= 2; continue; }
        size_t cnt = count >> 1;
        { KScope ks_(c, "merkle_level_kernel", 96.0 * cnt); hipLaunchKernelGGL(merkle_level_kernel, dim3((unsigned)((cnt + HASH_THREADS - 1) / HASH_THREADS)), dim3(HASH_THREADS), 0, c->stream,
                           (const digest*)(nodes + count), nodes + cnt, cnt); }
        count = cnt;
    }
    { KScope ks_(c, "merkle_top_kernel", 96.0 * count); hipLaunchKernelGGL(merkle_top_kernel, dim3(1), dim3(HASH_THREADS), 0, c->stream, nodes, (uint32_t)count); }
}

void k_merkle_levels(dst_ctx* c, const digest* leaves, digest* nodes, size_t num_leaves) {
    if (merkle_two_levels(c, leaves, nodes, num_leaves >> 2)) { merkle_upper_levels(c, nodes, num_leaves >> 2); return; }
    size_t cnt = num_leaves >> 1;
    { KScope ks_(c, "merkle_level_kernel", 96.0 * cnt); hipLaunchKernelGGL(merkle_level_kernel, dim3((unsigned)((cnt + HASH_THREADS - 1) / HASH_THREADS)), dim3(HASH_THREADS), 0, c->stream,
                       leaves, nodes + cnt, cnt); }
    merkle_upper_levels(c, nodes, cnt);
}

// ---- constraint tree: leaves are raw evaluation pairs (prover.rs:180-187); the first node level hashes four consecutive
//      natural-order evaluations B*k + 4q .. 4q+3, i.e. cosets 4q..4q+3 at index k ---------------------------------------------
__global__ void __launch_bounds__(HASH_THREADS) constraint_level1_kernel(const fe* __restrict__ cevals, digest* __restrict__ out,
                                                                        size_t n, uint32_t Bc, uint32_t log_qt) {
    __shared__ digest tile[HASH_THREADS];
    const uint32_t QT = 1u << log_qt, KT = blockDim.x >> log_qt;
    const uint32_t kk = threadIdx.x % KT, qq = threadIdx.x / KT;
    const size_t k = (size_t)blockIdx.x * KT + kk;
    const uint32_t q = blockIdx.y * QT + qq;
    const fe* base = cevals + (size_t)(4 * q) * n + k;
    uint32_t m[16], h[8];
#pragma unroll
    for (uint32_t e = 0; e < 4; e++) {
        fe v = base[(size_t)e * n];
        m[4 * e] = v.v[0]; m[4 * e + 1] = v.v[1]; m[4 * e + 2] = v.v[2]; m[4 * e + 3] = v.v[3];
    }
    b3_hash64(m, h);
    store_digest(&tile[kk * QT + qq], h);
    __syncthreads();
    const uint32_t kk2 = threadIdx.x >> log_qt, qq2 = threadIdx.x & (QT - 1);
    const size_t o = ((size_t)blockIdx.x * KT + kk2) * (Bc / 4) + blockIdx.y * QT + qq2;
    out[o] = tile[kk2 * QT + qq2];
}

void k_constraint_tree(dst_ctx* c) {
    uint32_t qn = (uint32_t)(c->Bc / 4);
    uint32_t qt = qn < 32 ? qn : 32u, log_qt = 0;
    while ((1u << log_qt) < qt) log_qt++;
    const uint32_t threads = tile_threads(c->n, log_qt), KT = threads >> log_qt;
    size_t level1 = c->N / 4;          // leaves = N/2, first node level = N/4 entries at nodes[N/4 ..)
    dim3 g((unsigned)(c->n / KT), (unsigned)(qn >> log_qt));
    { KScope ks_(c, "constraint_level1_kernel", 96.0 * level1); hipLaunchKernelGGL(constraint_level1_kernel, g, dim3(threads), 0, c->stream, (const fe*)c->cevals, c->cnodes + level1, c->n, (uint32_t)c->Bc, log_qt); }
    merkle_upper_levels(c, c->cnodes, level1);
}

// ---- FRI leaves: leaf(r) = BLAKE3(e[r] || e[r+R] || e[r+2R] || e[r+3R]), R = N_d / 4 (quartic.rs:137, fri/utils.rs:16) ----------
// layer 0 reads the coset-major composition evaluations: r = B*k + j with k < n/4, and r + s*R = B*(k + s*n/4) + j
__global__ void __launch_bounds__(HASH_THREADS) fri_leaves0_kernel(const fe* __restrict__ comp, digest* __restrict__ leaves,
                                                                  size_t n, uint32_t Bc, uint32_t log_jt) {
    __shared__ digest tile[HASH_THREADS];
    const uint32_t JT = 1u << log_jt, KT = blockDim.x >> log_jt;          // the launcher shrinks the block for tiny traces
    const uint32_t kk = threadIdx.x % KT, jj = threadIdx.x / KT;
    const size_t k = (size_t)blockIdx.x * KT + kk;
    const uint32_t j = blockIdx.y * JT + jj;
    const fe* base = comp + (size_t)j * n + k;
    const size_t q = n / 4;
    uint32_t m[16], h[8];
#pragma unroll
    for (uint32_t e = 0; e < 4; e++) {
        fe v = base[(size_t)e * q];
        m[4 * e] = v.v[0]; m[4 * e + 1] = v.v[1]; m[4 * e + 2] = v.v[2]; m[4 * e + 3] = v.v[3];
    }
    b3_hash64(m, h);
    store_digest(&tile[kk * JT + jj], h);
    __syncthreads();
    const uint32_t kk2 = threadIdx.x >> log_jt, jj2 = threadIdx.x & (JT - 1);
    const size_t o = ((size_t)blockIdx.x * KT + kk2) * Bc + blockIdx.y * JT + jj2;
    leaves[o] = tile[kk2 * JT + jj2];
}

void k_fri_leaves_layer0(dst_ctx* c) {
    uint32_t jt = c->Bc < 32 ? (uint32_t)c->Bc : 32u, log_jt = 0;
    while ((1u << log_jt) < jt) log_jt++;
    size_t kq = c->n / 4;
    const uint32_t threads = tile_threads(kq, log_jt), KT = threads >> log_jt;
    dim3 g((unsigned)(kq / KT), (unsigned)(c->Bc >> log_jt));
    { KScope ks_(c, "fri_leaves0_kernel", 96.0 * kq * c->Bc); hipLaunchKernelGGL(fri_leaves0_kernel, g, dim3(threads), 0, c->stream, (const fe*)c->comp, c->fri_leaves[0], c->n, (uint32_t)c->Bc, log_jt); }
}

__global__ void __launch_bounds__(HASH_THREADS) fri_leaves_kernel(const fe* __restrict__ e, digest* __restrict__ leaves, size_t R) {
    size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    uint32_t m[16], h[8];
#pragma unroll
    for (uint32_t s = 0; s < 4; s++) {
        fe v = e[r + (size_t)s * R];
        m[4 * s] = v.v[0]; m[4 * s + 1] = v.v[1]; m[4 * s + 2] = v.v[2]; m[4 * s + 3] = v.v[3];
    }
    b3_hash64(m, h);
    store_digest(leaves + r, h);
}

void k_fri_leaves_at(dst_ctx* c, const fe* e, digest* leaves, size_t R) {        // natural-order layer of 4R evaluations
    { KScope ks_(c, "fri_leaves_kernel", 96.0 * R); hipLaunchKernelGGL(fri_leaves_kernel, dim3((unsigned)((R + HASH_THREADS - 1) / HASH_THREADS)), dim3(HASH_THREADS), 0, c->stream,
                       e, leaves, R); }
}
void k_fri_leaves(dst_ctx* c, int layer) { k_fri_leaves_at(c, c->fri_e[layer], c->fri_leaves[layer], c->fri_size[layer] / 4); }

// ---- coset-sharded trees (world > 1): a rank owns leaves B*k + j for its cosets j and keeps them as local index k*Bc + jl, so the
//      lowest log2(Bc) levels of every tree are rank-local.  `k_merkle_levels_to` builds the local heap down to `stop_count`
//      nodes (one per k); after the all-gather `k_upper_tree` interleaves the ranks' boundary nodes (node G*k + g = gathered[g][k])
//      and finishes the replicated upper part of the tree. ---------------------------------------------------------------------
void k_merkle_local_levels(dst_ctx* c, digest* nodes, size_t count, size_t stop_count) {      // nodes[count..2count) valid on entry
    while (count > stop_count) {
        if ((count >> 2) >= stop_count && merkle_two_levels(c, nodes + count, nodes, count >> 2)) { count >>= 2; continue; }
        size_t cnt = count >> 1;
        { KScope ks_(c, "merkle_level_kernel", 96.0 * cnt); hipLaunchKernelGGL(merkle_level_kernel, dim3((unsigned)((cnt + HASH_THREADS - 1) / HASH_THREADS)), dim3(HASH_THREADS), 0, c->stream,
                           (const digest*)(nodes + count), nodes + cnt, cnt); }
        count = cnt;
    }
}
void k_merkle_levels_to(dst_ctx* c, const digest* leaves, digest* nodes, size_t num_leaves, size_t stop_count) {
    if ((num_leaves >> 2) >= stop_count && merkle_two_levels(c, leaves, nodes, num_leaves >> 2)) { k_merkle_local_levels(c, nodes, num_leaves >> 2, stop_count); return; }
    size_t cnt = num_leaves >> 1;
    { KScope ks_(c, "merkle_level_kernel", 96.0 * cnt); hipLaunchKernelGGL(merkle_level_kernel, dim3((unsigned)((cnt + HASH_THREADS - 1) / HASH_THREADS)), dim3(HASH_THREADS), 0, c->stream,
                       leaves, nodes + cnt, cnt); }
    k_merkle_local_levels(c, nodes, cnt, stop_count);
}
__global__ void interleave_boundary_kernel(const digest* __restrict__ gathered, digest* __restrict__ out, size_t nb, uint32_t G) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nb * G) return;
    size_t k = t / G, g = t % G;
    out[t] = gathered[g * nb + k];
}
void k_upper_tree(dst_ctx* c, const digest* gathered, digest* upper, size_t nb, uint32_t G) {
    size_t count = nb * G;
    { KScope ks_(c, "interleave_boundary_kernel", 64.0 * count); hipLaunchKernelGGL(interleave_boundary_kernel, dim3((unsigned)((count + HASH_THREADS - 1) / HASH_THREADS)), dim3(HASH_THREADS), 0, c->stream, gathered, upper + count, nb, G); }
    merkle_upper_levels(c, upper, count);
}
void k_merkle_upper(dst_ctx* c, digest* nodes, size_t count) { merkle_upper_levels(c, nodes, count); }
// dst[i] = the digest at the head of record i (records `stride` bytes apart): the ranks' subtree roots out of the exchanged root + status records
__global__ void digests_from_records_kernel(const uint8_t* __restrict__ recs, size_t stride, digest* __restrict__ dst, uint32_t count) {
    const uint32_t i = threadIdx.x >> 3, w = threadIdx.x & 7;
    if (i < count) dst[i].w[w] = reinterpret_cast<const uint32_t*>(recs + (size_t)i * stride)[w];
}
void k_digests_from_records(dst_ctx* c, const void* recs, size_t stride, digest* dst, size_t count) {      // count <= 8
    KScope ks_(c, "digests_from_records_kernel", 0.0);
    hipLaunchKernelGGL(digests_from_records_kernel, dim3(1), dim3(64), 0, c->stream, (const uint8_t*)recs, stride, dst, (uint32_t)count);
}
// first node level of the constraint tree only (local), see k_constraint_tree
void k_constraint_level1(dst_ctx* c) {
    uint32_t qn = (uint32_t)(c->Bc / 4);
    uint32_t qt = qn < 32 ? qn : 32u, log_qt = 0;
    while ((1u << log_qt) < qt) log_qt++;
    const uint32_t threads = tile_threads(c->n, log_qt), KT = threads >> log_qt;
    size_t level1 = c->Bc * c->n / 4;
    dim3 g((unsigned)(c->n / KT), (unsigned)(qn >> log_qt));
    { KScope ks_(c, "constraint_level1_kernel", 96.0 * level1); hipLaunchKernelGGL(constraint_level1_kernel, g, dim3(threads), 0, c->stream, (const fe*)c->cevals, c->cnodes + level1, c->n, (uint32_t)c->Bc, log_qt); }
}
// FRI leaves of a coset-major layer e[Bc][nd]: leaf (k, jl), k < nd/4, local index k*Bc + jl
__global__ void __launch_bounds__(HASH_THREADS) fri_leaves_cm_kernel(const fe* __restrict__ e, digest* __restrict__ leaves, size_t nd, uint32_t Bc) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t q = nd / 4;
    if (t >= q * Bc) return;
    size_t jl = t / q, k = t % q;
    const fe* base = e + jl * nd + k;
    uint32_t m[16], h[8];
#pragma unroll
    for (uint32_t s = 0; s < 4; s++) {
        fe v = base[(size_t)s * q];
        m[4 * s] = v.v[0]; m[4 * s + 1] = v.v[1]; m[4 * s + 2] = v.v[2]; m[4 * s + 3] = v.v[3];
    }
    b3_hash64(m, h);
    store_digest(leaves + k * Bc + jl, h);
}
void k_fri_leaves_cm(dst_ctx* c, const fe* e, digest* leaves, size_t nd) {
    size_t total = nd / 4 * c->Bc;
    { KScope ks_(c, "fri_leaves_cm_kernel", 96.0 * total); hipLaunchKernelGGL(fri_leaves_cm_kernel, dim3((unsigned)((total + HASH_THREADS - 1) / HASH_THREADS)), dim3(HASH_THREADS), 0, c->stream, e, leaves, nd, (uint32_t)c->Bc); }
}
