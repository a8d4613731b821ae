// NTT / low-degree-extension kernels for gfx950.
//
// Replaces the recursive radix-2 `fft_in_place` (+ `permute`) of /root/reference/src/math/fft.rs:16-108 as it is used by
// TraceTable::extend (src/stark/trace/trace_table.rs:143-169), ConstraintTable::combine_polys (constraint_table.rs:54-88),
// ConstraintPoly::eval (constraint_poly.rs:28-37) and the composition evaluation (prover.rs:98-101).  Field results are
// unique, so the algorithm is free: every transform here is a "four-step" NTT in HBM passes over LDS tiles,
//   two passes (n = n1*n2, n < 2^21):
//     pass A: for a tile of T adjacent columns m2, the n1-point NTT over the stride-n2 dimension, in LDS, with the four-step
//             twiddle w_N^(m2*(B*k1+j)) on store (streamed from a table).  For an extension the transform is a coset DIT whose
//             stage twiddles carry the coset pre-scale w_N^(j*n2*m1); otherwise a DIF; two radix-2 stages per LDS round trip;
//     pass B: for a tile of T adjacent rows k1, the n2-point NTT over the contiguous dimension, in LDS, stored as
//             X[k1 + n1*k2] so that the output is in natural order;
//   three passes (n = n1*nm*n3, n >= 2^21): pass A, pass A again on every row of n/n1 points, pass B with tiles of adjacent k1.
// Workgroups are persistent over adjacent tiles; the 512-lane instances prefetch the next tile into registers while the current one
// is in LDS, the 1024-lane instances of 1024-point tiles (8 waves per SIMD) leave the overlap to the CU's other workgroup.
// The low-degree extension never materialises the zero-padded size-N input of the reference: the N = B*n evaluations
// are B coset transforms of size n (coset j holds the reference's indices B*k + j), stored coset-major; coset 0 of the trace
// extension is the trace itself and is copied.  HBM accesses are T*16-byte segments (T = 4 for 1024-point tiles, 16 for the
// 256-point tiles of three-pass plans).  A second, register-radix kernel family (ntt_reg_kernel) is kept as an independent
// implementation that the tests run at every tile length.
#include "ctx.h"
#include <mutex>
#include <type_traits>
#include "ntt_lds.h"

#if defined(NTT_NO_ROTATE)
#define NTT_NO_ROTATE_DEFINED 1
#else
#define NTT_NO_ROTATE_DEFINED 0
#endif
#define NTT_TILE_ELEMS 4096      // elements of an LDS tile (64 KiB); a workgroup of 512 lanes holds 8 per lane, one of 1024 lanes 4

struct NttArgs {
    const fe* src; fe* dst;
    size_t src_col_stride, src_coset_stride;   // elements
    size_t dst_col_stride, dst_coset_stride;
    const fe_tw* stage_tw;     // w_len^t, t < len/2 (forward or inverse), as table pairs
    const fe_tw* prescale;     // w_{B*n1}^t or nullptr
    const fe_tw* dit_last;     // coset DIT whose last-stage twiddles stay in global memory: [B][n1/2] pairs w_{B*n1}^(j + B*k), else nullptr
    const tw4_t* tw4;          // four-step twiddles in output order [coset][k1][m2] (coset stride tw4_coset_stride)
    size_t tw4_coset_stride;
    uint32_t log_n1, log_n2, tile /* log2 of the tile width */, log_N, log_b;
    uint32_t j0;               // global index of the first local coset
    uint32_t has_scale;        // 1: multiply the result by `scale` (1/n of inverse transforms)
    uint32_t tiles_per_block;  // adjacent tiles one workgroup walks through
    uint32_t debug;            // DISTAFF_NTT_DEBUG (test build): 1 = report the occupancy of every launch shape once
    // pass B addressing (two-pass plans: row stride n2, frequency stride n1, no batch); three-pass plans run the last pass once per
    // middle frequency k2 (batch index = low bits of the tile-group index): source rows (k1, k2, .) and destination k1 + n1 * (k2 + n2' * k3)
    size_t src_row_stride, dst_k_stride, src_batch_stride, dst_batch_stride;
    uint32_t batch_log;
    uint32_t dit;              // pass A of an extension: coset DIT (pre-scale folded into the stage twiddles, taken from `prescale`)
    uint32_t groups, cosets, cols;   // extent of the linear block index: tile groups x cosets x columns (registers)
    uint32_t coset_fast;       // block order of first passes over several cosets, see ntt_block
    // Register pre-stage (instances with PRE = 1): the pass's transform has TWICE the length of its LDS tile.  A workgroup owns one half h
    // of the frequencies (k = 2 k' + h; h = upper half of the tile-group index): it loads the elements m and m + len of a column,
    // forms u_h[m] in registers and runs the len-point LDS transform on it --
    //   coset DIT (extension):  u_h = x[m] + (-1)^h c x[m + len], c = g^len, followed by the coset transform with shift g * w^h
    //                           (virtual coset j + B h of the same pre-scale table);
    //   DIF (everything else):  u_0 = x[m] + x[m + len],  u_1 = (x[m] - x[m + len]) * w_{2 len}^m  (pre_tw).
    // n = 2^21 and 2^22 then stay on the two-pass plan with 1024 x 4 tiles (a third HBM pass costs more than the extra stage).
    uint32_t pre;              // 0 / 1 (read by the PRE instances only)
    const fe_tw* pre_tw;       // w_{2 len}^m, m < len (forward or inverse), DIF pre-stage
    fe_tw scale;
};

__device__ __forceinline__ fe load_fe(const fe* p) { return *p; }

__device__ __forceinline__ fe dom_pow(const fe* lo, const fe* hi, uint32_t lo_bits, uint64_t e) {
    uint32_t l = (uint32_t)e & ((1u << lo_bits) - 1u), h = (uint32_t)(e >> lo_bits);
    fe a = lo[l];
    if (h == 0) return a;
    return fe_mul(a, hi[h]);
}

// Linear block index -> (tile group, local coset, column).  Block b runs on XCD b % 8.  The COLUMN (register) is the fastest dimension and
// the blocks that differ only in it carry the same index modulo 8: the four-step twiddles of a (coset, tile) - the same for every
// register - are served to the other registers by that XCD's L2.  (Round 1 had the register as the slowest grid dimension: the table was
// streamed once per register.)  coset_fast (first passes of an extension): the coset is the next dimension and an XCD finishes a tile
// group before it starts the next, so the coefficient tiles of the group are shared by the cosets as well.
__device__ __forceinline__ void ntt_block(const NttArgs& a, uint32_t& group, uint32_t& jl, uint32_t& col) {
    const uint32_t b = blockIdx.x, units = a.groups * a.cosets;
    uint32_t u;
    if (a.coset_fast) {        // XCD x works through the tile groups x, x + 8, ...; within a group every (coset, register) pair before the next group,
        // the registers in chunks of four: the 64 workgroups resident on an XCD are then 16 cosets x 4 registers of one tile group (a chunk of all
        // 20 registers would leave 3 cosets to share a coefficient tile: measured 41 against 33 GB of L2 misses per proof in the first passes)
        const uint32_t s = b >> 3, pairs = a.cosets * a.cols, pair = s % pairs;
        group = (s / pairs) * 8u + (b & 7u);
        const uint32_t chunk = pair / (4u * a.cosets), in_chunk = pair - chunk * 4u * a.cosets;
        const uint32_t width = (chunk * 4u + 4u <= a.cols) ? 4u : a.cols - chunk * 4u;              // the last chunk may be narrower
        col = chunk * 4u + in_chunk % width; jl = in_chunk / width;
        return;
    }
    if ((units & 7u) == 0) { const uint32_t s = b >> 3; col = s % a.cols; u = (s / a.cols) * 8u + (b & 7u); }
    else { col = b % a.cols; u = b / a.cols; }
    jl = u / a.groups; group = u % a.groups;
}

// consumer of the last butterfly round of the second passes (ntt_lds.h): the results leave for HBM without another pass over LDS
struct OutB {                                          // natural-order store X[k1 + n1 * k2] from the last round, 1/n of inverse transforms on the way
    static constexpr bool active = true;
    fe* __restrict__ dst /* at the tile's first k1 */; uint32_t k_stride, log_n2; bool scale; fe_tw s;
    uint32_t pre_shift, h;                             // register pre-stage: the tile holds the frequencies k2 = (k' << pre_shift) + h
    struct Tok {};
    __device__ __forceinline__ Tok pre(uint32_t, uint32_t) const { return Tok{}; }
    __device__ __forceinline__ void put(uint32_t row, uint32_t t, const fe& v, const Tok&) const {
        const uint32_t k2 = ((log_n2 ? (__brev(row) >> (32 - log_n2)) : 0u) << pre_shift) + h;
#if NTT_STREAM_B_STORE
        fe_store_stream(dst + (k2 * k_stride + t), scale ? fe_mul_tw(v, s) : v);
#else
        dst[k2 * k_stride + t] = scale ? fe_mul_tw(v, s) : v;            // uniform base + 32-bit lane offset (at most 2^24 points)
#endif
    }
};

// Streaming hints on the arrays a pass touches exactly once (fe_store_stream / fe_load_stream, fe.h): the staging array between the passes
// (written by the first, read by the second) and the finished extension (read next by the leaf hashing).  Measured on one lease, four
// alternating rounds of the 2^20 proof (tools/ab_multi.sh, round 5): none 35.63 ms, first-pass stores 35.30, + second-pass stores 35.16,
// + second-pass loads 35.15 (first passes 11.84 -> 11.41 ms).  FETCH_SIZE of the first passes does not move (10.3 -> 10.1 GB raw per proof):
// what they re-read is not pushed out by their own output, it is the four-step table and the coefficient tiles in 64-byte row segments.
#ifndef NTT_STREAM_STORE
#define NTT_STREAM_STORE 1
#endif
#ifndef NTT_STREAM_B_STORE
#define NTT_STREAM_B_STORE 1
#endif
#ifndef NTT_STREAM_B_LOAD
#define NTT_STREAM_B_LOAD 1
#endif
// Laboratory switches of the first pass (tools/r6_lde_lab.sh; WRONG results on purpose, never defined in a build that ships): each takes one
// of the pass's three global streams out of the memory system while every instruction stays -- the bound of anything that could be done
// about that stream.  NTT_LAB_SRC_SMALL=1: the coefficient loads of every workgroup fall into a 64 KiB window (cache hits); =2: only those
// of the odd-half workgroups of a register pre-stage pass (the upper bound of SHARING the loads between the halves);
// NTT_LAB_TW4_SMALL: the four-step twiddles come from a 64 KiB window per coset; NTT_LAB_DST_DENSE: a tile is stored as one contiguous
// 64 KiB block instead of 64-byte row segments a row stride apart.
// (the window is indexed by (row, column) of the tile, so every row still carries different data: a window indexed by the masked ARRAY offset
// would make the columns periodic, their spectra sparse -- and the kernels measurably faster for that alone, their time depends on the data)
#if defined(NTT_LAB_SRC_SMALL)
#define NTT_LAB_SRC(row, t, off) ((((NTT_LAB_SRC_SMALL) == 1) || hh) ? ((((row) & 1023u) << 2) + ((t) & 3u)) : (off))
#else
#define NTT_LAB_SRC(row, t, off) (off)
#endif
#if defined(NTT_LAB_TW4_SMALL)
#define NTT_LAB_TW(row, t, off) ((((row) & 1023u) << 2) + ((t) & 3u))
#else
#define NTT_LAB_TW(row, t, off) (off)
#endif
#if defined(NTT_LAB_STAMPS)
// [pass 0/1][workgroup slot 256][wave 16][tile 2][stamp 16]: written by lane 0 of every wave of the workgroups blockIdx.x = 61 * slot
#define NTT_LAB_WG_STRIDE 61u
__device__ unsigned long long ntt_lab_buf[2 * 256 * 16 * 2 * 16];
extern "C" __attribute__((visibility("default"))) int dst_lab_stamps(unsigned long long* out, size_t count, int clear) {
    if (count > sizeof(ntt_lab_buf) / 8) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ntt_lab_buf), count * 8) != hipSuccess) return -2;
    if (clear) { static unsigned long long zero[2 * 256 * 16 * 2 * 16]; if (hipMemcpyToSymbol(HIP_SYMBOL(ntt_lab_buf), zero, sizeof zero) != hipSuccess) return -3; }
    return 0;
}
#define NTT_LAB_FLUSH(pass, it, st, nst) do { \
    if ((blockIdx.x % NTT_LAB_WG_STRIDE) == 0u && blockIdx.x / NTT_LAB_WG_STRIDE < 256u && (it) < 2u && (threadIdx.x & 63u) == 0u) { \
        unsigned long long* o_ = ntt_lab_buf + ((((size_t)(pass) * 256u + blockIdx.x / NTT_LAB_WG_STRIDE) * 16u + (threadIdx.x >> 6)) * 2u + (it)) * 16u; \
        for (int k_ = 0; k_ < (nst); k_++) o_[k_] = (st)[k_]; } } while (0)
#else
#define NTT_LAB_FLUSH(pass, it, st, nst) ((void)0)
#endif
extern __shared__ __attribute__((aligned(16))) unsigned char ntt_smem[];

// Both passes are persistent over `tiles_per_block` adjacent tiles.  PREFETCH instances (WPE = 4 waves per SIMD, 128 registers): the
// tile's elements for the NEXT iteration are fetched from HBM into registers before the butterfly stages of the current one start, so
// the HBM latency and most of the transfer overlap with the arithmetic (measured: a block of this occupancy that loads, computes and
// stores in sequence pays HBM time + ALU time, not their maximum).  <1024, 8, false>: two workgroups of 1024 lanes per CU, 8 waves per
// SIMD in 64 registers, no prefetch -- while one workgroup loads, the other one's 16 waves keep the SIMDs busy (ntt_launch picks it for
// 1024-point tiles).  The stage twiddles live in LDS behind the tile as table pairs (32 bytes each), so the stages issue no global loads
// that would have to wait behind the prefetch.  A 1024-point coset DIT with its whole table in LDS needs 64 KiB of tile + 32 KiB of
// twiddles: that instance runs as ONE workgroup of 1024 lanes per CU; everything that fits 80 KiB runs as two workgroups per CU.
// LOG_LEN / LOG_T != 0: the instance of ONE tile shape (2^LOG_LEN points x 2^LOG_T columns) -- its rounds are separate code with literal
// strides (ntt_lds.h) and the tile index arithmetic folds into immediates; 0: any shape, from the arguments.
template <int THREADS, int WPE = 4, bool PREFETCH = true, int LOG_LEN = 0, int LOG_T = 0, int PRE = 0>
__global__ void __launch_bounds__(THREADS, WPE) ntt_pass_a(NttArgs a, const fe* __restrict__ src_base, fe* __restrict__ dst_base) {
    constexpr int EPT = NTT_TILE_ELEMS / THREADS;
    fe* L = reinterpret_cast<fe*>(ntt_smem);
    const uint32_t log_n1 = LOG_LEN ? (uint32_t)LOG_LEN : a.log_n1;          // length of the LDS transform (half of the pass's with PRE)
    const uint32_t log_t = LOG_T ? (uint32_t)LOG_T : a.tile, T = 1u << log_t, n1 = 1u << log_n1;
    fe_tw* TW = reinterpret_cast<fe_tw*>(L + n1 * T);
    uint32_t group, jl, col;
    ntt_block(a, group, jl, col);
    uint32_t hh = 0;                                       // PRE: which half of the frequencies this workgroup produces
    if (PRE) { const uint32_t half = a.groups >> 1; hh = group >= half ? 1u : 0u; group -= hh * half; }
    const uint32_t jg = a.j0 + jl;
    const fe* __restrict__ src0 = src_base + (size_t)col * a.src_col_stride + (size_t)jl * a.src_coset_stride;
    fe* __restrict__ dst0 = dst_base + (size_t)col * a.dst_col_stride + (size_t)jl * a.dst_coset_stride;
    const uint32_t pmask = (1u << (a.log_b + log_n1 + PRE)) - 1u;
    const uint32_t jv = jg + (hh << a.log_b);              // PRE: the half-length transform of frequencies 2k' + h is the coset transform with shift g * w^h
    if (a.dit) {
        // stage twiddles of this coset: g^(n1/B) * w_B^k = w_{B_lde*n1}^((j + B_lde*k) * n1/B) straight from the pre-scale table
        const uint32_t in_lds = a.dit_last ? n1 / 2 : n1;                                     // entries + 1
        for (uint32_t i = threadIdx.x; i + 1 < in_lds; i += THREADS) {
            const uint32_t lb = 31u - (uint32_t)__clz(i + 1), k = i + 1 - (1u << lb);         // entry i: block size B = 2^(lb+1), index k
            TW[i] = a.prescale[((jv + (k << (a.log_b + PRE))) << (log_n1 - lb - 1)) & pmask];
        }
    } else
    for (uint32_t i = threadIdx.x; i < n1 / 2; i += THREADS) TW[dif_tw_slot(i)] = a.stage_tw[i];
    const bool scaled = !a.dit && a.prescale != nullptr && jg != 0;
    fe pre0 = fe_zero(), pre1 = fe_zero(), pre2 = fe_zero(), pre3 = fe_zero(), pre4 = fe_zero(), pre5 = fe_zero(), pre6 = fe_zero(), pre7 = fe_zero();
    const uint32_t count = n1 * T;
    // named registers, not an array: the prefetched elements must stay in VGPRs across the butterfly stages.  Branch-free fetch: a
    // lane past the tile re-reads element 0.
#define NTT_EACH(M) { M(0, pre0) M(1, pre1) M(2, pre2) M(3, pre3) M(4, pre4) M(5, pre5) M(6, pre6) M(7, pre7) }
    // uniform base (scalar registers) + 32-bit lane offset: the transform has at most 2^24 points
    // Element (row, column) of the tile that lane index idx carries.  A coset DIT enters LDS bit-reversed: the rows r .. r + 3 of one LDS
    // access group land 256-byte multiples apart, i.e. on the same banks; rotating the column by the row spreads them (the 64-byte HBM
    // segment of a row is read by the same four lanes as before, in another order).
    const bool rot_a = a.dit && log_t == 2 && !NTT_NO_ROTATE_DEFINED;
#define NTT_COL_A(idx) (rot_a ? (((idx) + ((idx) >> log_t)) & (T - 1)) : ((idx) & (T - 1)))
    // PRE: the element is formed from the rows m and m + n1 of the column as it arrives (see NttArgs)
    const fe_tw pre_c = (PRE && a.dit) ? a.prescale[(jg << log_n1) & pmask] : fe_tw{};       // c = g^len
    auto pre_stage_a = [&](const fe& x0, const fe& x1, uint32_t row) -> fe {
        if (a.dit) { const fe t = fe_mul_tw(x1, pre_c); return hh ? fe_sub(x0, t) : fe_add(x0, t); }
        return hh ? fe_mul_tw(fe_sub(x0, x1), a.pre_tw[row]) : fe_add(x0, x1);
    };
#define NTT_FETCH_A1(e, var) if constexpr ((e) < EPT) { uint32_t idx = lane + (e) * THREADS; idx = idx < count ? idx : 0u; \
        if constexpr (PRE != 0) { const uint32_t row_ = idx >> log_t; const fe x0_ = src[NTT_LAB_SRC(row_, NTT_COL_A(idx), (row_ << a.log_n2) + NTT_COL_A(idx))], x1_ = src[NTT_LAB_SRC(row_ + 512u, NTT_COL_A(idx), ((row_ + n1) << a.log_n2) + NTT_COL_A(idx))]; var = pre_stage_a(x0_, x1_, row_); } \
        else var = src[NTT_LAB_SRC(idx >> log_t, NTT_COL_A(idx), ((idx >> log_t) << a.log_n2) + NTT_COL_A(idx))]; }
#define NTT_FETCH_A(tile) { const fe* __restrict__ src = src0 + (tile) * T; NTT_EACH(NTT_FETCH_A1) }
    const tw4_t* __restrict__ tw4 = a.tw4 + (size_t)jl * a.tw4_coset_stride;
    const uint32_t tile0 = group * a.tiles_per_block;
    uint32_t lane = threadIdx.x;
    if (PREFETCH) NTT_FETCH_A(tile0)
    for (uint32_t it = 0; it < a.tiles_per_block; it++) {
        const uint32_t tile = tile0 + it;
        unsigned long long st[16]; (void)st;               // laboratory stamps (NTT_STAMP: nothing in the product build)
        NTT_STAMP_REAL(st, 14);
        NTT_STAMP(st, 8);
        lane = lds_opaque_lane();                          // per-tile index arithmetic is recomputed, not kept live (and spilled) across the loop
        if (!PREFETCH) NTT_FETCH_A(tile)
        NTT_STAMP(st, 9);
        __syncthreads();                                   // the previous tile has left LDS (and TW is complete)
        NTT_STAMP(st, 10);
        // DISTAFF_NTT_DIF: pre-scale + DIF instead of the coset DIT
#define NTT_PUT_A(e, var) if constexpr ((e) < EPT) { const uint32_t idx = lane + (e) * THREADS; if (idx < count) { if (a.dit) L[lds_slot(__brev(idx >> log_t) >> (32 - log_n1), NTT_COL_A(idx), log_t)] = var; else L[lds_slot(idx >> log_t, idx & (T - 1), log_t)] = scaled ? fe_mul_tw(var, a.prescale[(jg * (idx >> log_t)) & pmask]) : var; } }
        NTT_EACH(NTT_PUT_A)
#undef NTT_PUT_A
        NTT_STAMP(st, 11);
        __syncthreads();
        NTT_STAMP(st, 12);
        if (PREFETCH && it + 1 < a.tiles_per_block) NTT_FETCH_A(tile + 1)
        if constexpr (LOG_LEN != 0) {
            if (a.dit) lds_ntt_dit_fixed<THREADS, LOG_LEN, LOG_T>(L, TW, a.dit_last ? a.dit_last + (size_t)((jg << PRE) + hh) * (n1 / 2) : nullptr, NttKeepInLds(), st);
            else lds_ntt_dif_fixed<THREADS, LOG_LEN, LOG_T>(L, TW, NttKeepInLds(), st);
        } else {
            if (a.dit) lds_ntt_dit<THREADS>(L, TW, log_n1, log_t, 1u, log_n1 + 1u, a.dit_last ? a.dit_last + (size_t)((jg << PRE) + hh) * (n1 / 2) : nullptr);
            else lds_ntt_dif<THREADS>(L, TW, log_n1, log_t, 1u, log_n1 + 1u);
        }
        // read-out in batches of RB elements per lane: the twiddle loads of a batch are in flight together (the 512-lane instance also
        // holds eight prefetched elements: a batch of four would spill).  The second passes store from their last butterfly round instead
        // (OutB); here that does not pay: in the 128-register instances the four-step twiddles (and the last-stage pairs from global memory)
        // live in the last round spill (13.15 / 13.3 against 13.05 ms), in the spill-free fixed-shape instance it is within the noise
        // (18.85 against 18.95 ms of extension, with the twiddle requested before or after the butterfly).
        constexpr int RB = WPE > 4 ? 1 : THREADS == 512 ? 2 : 4;
#if defined(NTT_LAB_DST_DENSE)
        fe* __restrict__ dst = dst_base + ((size_t)blockIdx.x * a.tiles_per_block + it) * count;
#else
        fe* __restrict__ dst = dst0 + tile * T;          // uniform bases, 32-bit lane offsets
#endif
        const tw4_t* __restrict__ tw = tw4 + tile * T;
        for (uint32_t base = 0; base < count; base += RB * THREADS) {
            fe v[RB]; tw4_t w[RB]; uint32_t off[RB]; bool ok[RB];
            static_for<0, RB>([&](auto q_) {
                constexpr int q = decltype(q_)::value;
                uint32_t idx = base + q * THREADS + lane;
                ok[q] = idx < count;
                idx = ok[q] ? idx : 0u;
                const uint32_t t = idx & (T - 1), r = idx >> log_t;
                const uint32_t k1 = ((a.dit ? r : __brev(r) >> (32 - log_n1)) << PRE) + hh;        // DIT leaves the tile in natural order; PRE: frequencies 2 k' + h
                off[q] = (k1 << a.log_n2) + t;
                v[q] = L[idx];
                w[q] = tw[NTT_LAB_TW(r, t, off[q])];
#if defined(NTT_LAB_DST_DENSE)
                off[q] = idx;
#endif
            });
            static_for<0, RB>([&](auto q_) {
                constexpr int q = decltype(q_)::value;
#if NTT_TW4_PAIRS
                const fe out_q = fe_mul_tw(v[q], w[q]);
#else
                const fe out_q = fe_mul(v[q], w[q]);
#endif
#if NTT_STREAM_STORE
                if (ok[q]) fe_store_stream(dst + off[q], out_q);          // the staging array: read next by the second pass, not by this one
#else
                if (ok[q]) dst[off[q]] = out_q;
#endif
            });
        }
        NTT_STAMP(st, 13);
        NTT_STAMP_REAL(st, 15);
        NTT_LAB_FLUSH(0, it, st, 16);
    }
}

// WPE: waves per SIMD the instance is compiled for (its register budget); PREFETCH: the next tile travels in registers during the rounds
template <int THREADS, int WPE = 4, bool PREFETCH = true, int LOG_LEN = 0, int LOG_T = 0, int PRE = 0>
__global__ void __launch_bounds__(THREADS, WPE) ntt_pass_b(NttArgs a, const fe* __restrict__ src_base, fe* __restrict__ dst_base) {
    constexpr int EPT = NTT_TILE_ELEMS / THREADS;
    fe* L = reinterpret_cast<fe*>(ntt_smem);
    const uint32_t log_n2 = LOG_LEN ? (uint32_t)LOG_LEN : a.log_n2;          // length of the LDS transform (half of the pass's with PRE)
    const uint32_t log_t = LOG_T ? (uint32_t)LOG_T : a.tile, T = 1u << log_t, n2 = 1u << log_n2;
    fe_tw* TW = reinterpret_cast<fe_tw*>(L + n2 * T);
    uint32_t g, jl, col;
    ntt_block(a, g, jl, col);
    uint32_t hh = 0;                                       // PRE: which half of the frequencies this workgroup produces (two-pass plans: no batches)
    if (PRE) { const uint32_t half = a.groups >> 1; hh = g >= half ? 1u : 0u; g -= hh * half; }
    const uint32_t batch = g & ((1u << a.batch_log) - 1u), group = g >> a.batch_log;
    const fe* __restrict__ src = src_base + (size_t)col * a.src_col_stride + (size_t)jl * a.src_coset_stride + (size_t)batch * a.src_batch_stride;
    fe* __restrict__ dst = dst_base + (size_t)col * a.dst_col_stride + (size_t)jl * a.dst_coset_stride + (size_t)batch * a.dst_batch_stride;
    for (uint32_t i = threadIdx.x; i < n2 / 2; i += THREADS) TW[dif_tw_slot(i)] = a.stage_tw[i];
    fe pre0 = fe_zero(), pre1 = fe_zero(), pre2 = fe_zero(), pre3 = fe_zero(), pre4 = fe_zero(), pre5 = fe_zero(), pre6 = fe_zero(), pre7 = fe_zero();
    const uint32_t count = n2 * T;
    // contiguous along m2
    // Four-column tiles: lane index = (m2, row), row fastest -- a wavefront reads 16 consecutive elements (256 bytes) of each of the four
    // rows and writes 64 consecutive LDS slots.  (With m2 fastest the slots of neighbouring lanes are 64 bytes apart: the 16 lanes of an
    // access group share their banks four times over.)  Wider tiles keep m2 fastest.
    const bool colfast_b = log_t == 2 && !NTT_NO_ROTATE_DEFINED;
#define NTT_ROW_B(idx) (colfast_b ? ((idx) & (T - 1)) : ((idx) >> log_n2))
#define NTT_M2_B(idx) (colfast_b ? ((idx) >> log_t) : ((idx) & (n2 - 1)))
    // PRE (DIF pre-stage, see NttArgs): u_0 = y[m] + y[m + n2], u_1 = (y[m] - y[m + n2]) * w_{2 n2}^m
#if NTT_STREAM_B_LOAD
#define NTT_LOAD_B(p) fe_load_stream(p)
#else
#define NTT_LOAD_B(p) (*(p))
#endif
#define NTT_FETCH_B1(e, var) if constexpr ((e) < EPT) { uint32_t idx = lane + (e) * THREADS; idx = idx < count ? idx : 0u; \
        if constexpr (PRE != 0) { const uint32_t m_ = NTT_M2_B(idx); const fe* p_ = srct + NTT_ROW_B(idx) * row_stride + m_; const fe x0_ = p_[0], x1_ = p_[n2]; \
                                  var = hh ? fe_mul_tw(fe_sub(x0_, x1_), a.pre_tw[m_]) : fe_add(x0_, x1_); } \
        else var = NTT_LOAD_B(srct + (NTT_ROW_B(idx) * row_stride + NTT_M2_B(idx))); }
#define NTT_FETCH_B(tile) { const fe* __restrict__ srct = src + (size_t)((tile) * T) * a.src_row_stride; NTT_EACH(NTT_FETCH_B1) }
    const uint32_t row_stride = (uint32_t)a.src_row_stride;
    const uint32_t tile0 = group * a.tiles_per_block;
    uint32_t lane = threadIdx.x;
    if (PREFETCH) NTT_FETCH_B(tile0)
    for (uint32_t it = 0; it < a.tiles_per_block; it++) {
        const uint32_t k1_0 = (tile0 + it) * T;
        unsigned long long st[16]; (void)st;               // laboratory stamps (NTT_STAMP: nothing in the product build)
        NTT_STAMP_REAL(st, 14);
        NTT_STAMP(st, 8);
        lane = lds_opaque_lane();
        if (!PREFETCH) NTT_FETCH_B(tile0 + it)
        NTT_STAMP(st, 9);
        __syncthreads();
        NTT_STAMP(st, 10);
#define NTT_PUT_B(e, var) if constexpr ((e) < EPT) { const uint32_t idx = lane + (e) * THREADS; if (idx < count) L[lds_slot(NTT_M2_B(idx), NTT_ROW_B(idx), log_t)] = var; }
        NTT_EACH(NTT_PUT_B)
#undef NTT_PUT_B
        NTT_STAMP(st, 11);
        __syncthreads();
        NTT_STAMP(st, 12);
        if (PREFETCH && it + 1 < a.tiles_per_block) NTT_FETCH_B(tile0 + it + 1)
        const OutB out{dst + k1_0, (uint32_t)a.dst_k_stride, log_n2, a.has_scale != 0, a.scale, (uint32_t)PRE, hh};
        const fe_tw* Wuse = TW;
        if constexpr (LOG_LEN != 0) lds_ntt_dif_fixed<THREADS, LOG_LEN, LOG_T, OutB>(L, Wuse, out, st);
        else lds_ntt_dif<THREADS, OutB>(L, Wuse, log_n2, log_t, 1u, log_n2 + 1u, out);
        NTT_STAMP(st, 13);
        NTT_STAMP_REAL(st, 15);
        NTT_LAB_FLUSH(1, it, st, 16);
    }
}

// ---- register-radix transform (tile lengths 2^6 .. 2^12) -----------------------------------------------------------------------
// The same two passes, but every lane keeps 16 points in registers and runs 3-4 butterfly stages on them before the tile is
// exchanged through LDS, so a 1024-point tile needs two LDS exchanges and two barriers instead of ten.  Mixed-radix
// Cooley-Tukey over the digits (r1, r2[, r3]) of the tile length L = 2^(r1+r2+r3): with the time index written
// m = (m1, m2, m3) (m1 most significant) and the frequency k = k1 + R1*k2 + R1*R2*k3,
//   round i: for fixed other digits, the R_i-point DFT over m_i (radix-2 DIF in registers, roots w_16^j from the kernel
//            arguments, i.e. scalar registers), then the twiddle w_{L_i}^(k_i * mlow) from the stage table (L_i = remaining
//            length, mlow = the digits below i); the slot (.., m_i, ..) of the tile now holds (.., k_i, ..).
// Round 1 reads straight from HBM and the last round writes straight to HBM; lanes run over the T tile columns first, so
// every HBM access is a T*16-byte segment as before.  LDS slot = digits * T + t with an XOR swizzle that moves the 8-lane
// groups of the last round (stride R_q*T slots) onto distinct banks.
template <int LOGL> struct NttDigits;
template <> struct NttDigits<6>  { static constexpr int r1 = 3, r2 = 3, r3 = 0, log_t = 2; };
template <> struct NttDigits<7>  { static constexpr int r1 = 4, r2 = 3, r3 = 0, log_t = 2; };
template <> struct NttDigits<8>  { static constexpr int r1 = 4, r2 = 4, r3 = 0, log_t = 2; };
template <> struct NttDigits<9>  { static constexpr int r1 = 3, r2 = 3, r3 = 3, log_t = 2; };
template <> struct NttDigits<10> { static constexpr int r1 = 4, r2 = 3, r3 = 3, log_t = 2; };
template <> struct NttDigits<11> { static constexpr int r1 = 4, r2 = 4, r3 = 3, log_t = 2; };
template <> struct NttDigits<12> { static constexpr int r1 = 4, r2 = 4, r3 = 4, log_t = 1; };

struct NttRegArgs {
    const fe* src; fe* dst;
    size_t src_col_stride, src_coset_stride, dst_col_stride, dst_coset_stride;
    size_t in_stride_m, in_stride_t;    // element strides of the time index and of the tile column on input
    size_t out_stride_k;                // element stride of the frequency index on output (tile columns are contiguous)
    const fe_tw* stage_tw;              // w_L^t, t < L/2 (table pairs)
    const fe* tw_lo; const fe* tw_hi;   // two-level table of w_N (pass A four-step twiddle) or nullptr
    const fe_tw* prescale;              // w_{B*L}^t or nullptr
    uint32_t lo_bits, log_N, log_b, j0, coset_twiddle, has_scale;
    fe scale;
    fe c16[8];                          // w_16^j (forward or inverse), j < 8
};

constexpr __host__ __device__ int ntt_brev(int v, int bits) { int r = 0; for (int i = 0; i < bits; i++) r |= ((v >> i) & 1) << (bits - 1 - i); return r; }

// One out-of-line copy of the 128-bit modular multiplication for the register-radix kernels: with every call site inlined a
// 1024-point kernel is ~190 KB of straight-line code, several times the 64 KB instruction cache a CU pair shares, and the waves
// stall on instruction fetch; as a call the kernel is a few thousand instructions.
__device__ __attribute__((noinline)) fe fe_mul_call(fe a, fe b) { return fe_mul(a, b); }
__device__ __attribute__((noinline)) fe fe_mul_tw_call(fe a, fe p, fe q) { return fe_mul_tw(a, p, q); }

// radix-2 DIF over x[0 .. 2^r): x[rho] <- X[brev_r(rho)]
template <int r>
__device__ __forceinline__ void dft_regs(fe* x, const fe* c16) {
    constexpr int R = 1 << r;
    static_for<0, r>([&](auto s_) {
        constexpr int s = decltype(s_)::value, half = R >> (s + 1);
        static_for<0, R / 2>([&](auto i_) {
            constexpr int i = decltype(i_)::value, b = (i / half) * 2 * half, j = i % half;
            fe u = x[b + j], v = x[b + j + half];
            x[b + j] = fe_add(u, v);
            fe d = fe_sub(u, v);
            if constexpr (j == 0) x[b + j + half] = d;
            else x[b + j + half] = fe_mul_call(d, c16[(j << s) << (4 - r)]);
        });
    });
}

// v * w_L^e from the half table: w_L^(e + L/2) = -w_L^e
template <int LOGL>
__device__ __forceinline__ fe mul_stage_twiddle(const fe& v, const fe_tw* __restrict__ tw, uint32_t e) {
    constexpr uint32_t H = 1u << (LOGL - 1);
    const fe_tw w = tw[e & (H - 1)];
    const fe r = fe_mul_tw_call(v, w.p, w.q);
    return (e & H) ? fe_neg(r) : r;
}

template <int LOGL>
__global__ void __launch_bounds__((1 << (LOGL + NttDigits<LOGL>::log_t)) / 16, (LOGL + NttDigits<LOGL>::log_t >= 13) ? 1 : 2) ntt_reg_kernel(NttRegArgs a) {
    using D = NttDigits<LOGL>;
    constexpr int r1 = D::r1, r2 = D::r2, r3 = D::r3, LT = D::log_t, T = 1 << LT;
    constexpr int rq = r3 ? r3 : r2;                               // last digit
    constexpr int L = 1 << LOGL, E = 16, NT = L * T / E;
    constexpr int SWZ_SHIFT = LT + rq, SWZ_MASK = 8 / T - 1;
    __shared__ fe tile[L * T];
    auto phys = [](uint32_t s) -> uint32_t { return s ^ (((s >> SWZ_SHIFT) & SWZ_MASK) << LT); };
    const uint32_t tid = threadIdx.x;
    const uint32_t jl = blockIdx.y, jg = a.j0 + jl;
    const fe* src = a.src + (size_t)blockIdx.z * a.src_col_stride + (size_t)jl * a.src_coset_stride + (size_t)blockIdx.x * T * a.in_stride_t;
    fe* dst = a.dst + (size_t)blockIdx.z * a.dst_col_stride + (size_t)jl * a.dst_coset_stride + (size_t)blockIdx.x * T;
    fe x[E];

    // ---- round 1: HBM -> registers -> LDS
    {
        constexpr int R = 1 << r1, LO = L >> r1, SETS = E / R;
        static_for<0, E>([&](auto i_) {
            constexpr int i = decltype(i_)::value, u = i / R, d = i % R;
            const uint32_t w = tid + NT * u, t = w & (T - 1), p = w >> LT;
            x[i] = src[(size_t)(d * LO + p) * a.in_stride_m + (size_t)t * a.in_stride_t];
        });
        if (a.prescale != nullptr && jg != 0) {
            const uint32_t pmask = (1u << (a.log_b + LOGL)) - 1u;
            static_for<0, E>([&](auto i_) {
                constexpr int i = decltype(i_)::value, u = i / R, d = i % R;
                const uint32_t p = (tid + NT * u) >> LT;
                const fe_tw w = a.prescale[(jg * (uint32_t)(d * LO + p)) & pmask];
                x[i] = fe_mul_tw_call(x[i], w.p, w.q);
            });
        }
        static_for<0, SETS>([&](auto u_) {
            constexpr int u = decltype(u_)::value;
            const uint32_t w = tid + NT * u, t = w & (T - 1), p = w >> LT;
            dft_regs<r1>(x + u * R, a.c16);
            static_for<0, R>([&](auto rho_) {
                constexpr int rho = decltype(rho_)::value, k = ntt_brev(rho, r1);
                fe v = x[u * R + rho];
                if constexpr (k != 0) v = mul_stage_twiddle<LOGL>(v, a.stage_tw, (uint32_t)k * p);
                tile[phys((((uint32_t)k * LO + p) << LT) + t)] = v;
            });
        });
    }
    __syncthreads();
    // ---- round 2 of 3: LDS -> registers -> LDS (slots are owned by one lane, so only the rounds are separated by barriers)
    if constexpr (r3 != 0) {
        constexpr int R = 1 << r2, LO = L >> (r1 + r2), SETS = E / R;
        static_for<0, SETS>([&](auto u_) {
            constexpr int u = decltype(u_)::value;
            const uint32_t w = tid + NT * u, t = w & (T - 1), rest = w >> LT, low = rest & (LO - 1), high = rest / LO;
            const uint32_t base = (high * (R * LO) + low);
            static_for<0, R>([&](auto d_) {
                constexpr int d = decltype(d_)::value;
                x[u * R + d] = tile[phys(((base + d * LO) << LT) + t)];
            });
            dft_regs<r2>(x + u * R, a.c16);
            static_for<0, R>([&](auto rho_) {
                constexpr int rho = decltype(rho_)::value, k = ntt_brev(rho, r2);
                fe v = x[u * R + rho];
                if constexpr (k != 0) v = mul_stage_twiddle<LOGL>(v, a.stage_tw, ((uint32_t)k * low) << r1);
                tile[phys(((base + k * LO) << LT) + t)] = v;
            });
        });
        __syncthreads();
    }
    // ---- last round: LDS -> registers -> HBM
    {
        constexpr int R = 1 << rq, SETS = E / R;
        const uint64_t nmask = (1ull << a.log_N) - 1ull;
        static_for<0, SETS>([&](auto u_) {
            constexpr int u = decltype(u_)::value;
            const uint32_t w = tid + NT * u, t = w & (T - 1), high = w >> LT;
            static_for<0, R>([&](auto d_) {
                constexpr int d = decltype(d_)::value;
                x[u * R + d] = tile[phys(((high * R + d) << LT) + t)];
            });
            dft_regs<rq>(x + u * R, a.c16);
            // frequency of the digits above: k1 + R1 * k2 (three rounds: high = k1 * R2 + k2) or k1 (two rounds)
            const uint32_t klow = r3 ? ((high >> r2) + ((high & ((1u << r2) - 1u)) << r1)) : high;
            const uint32_t m2 = blockIdx.x * T + t;
            static_for<0, R>([&](auto rho_) {
                constexpr int rho = decltype(rho_)::value;
                const uint32_t k = klow + ((uint32_t)ntt_brev(rho, rq) << (LOGL - rq));
                fe v = x[u * R + rho];
                if (a.tw_lo != nullptr) {
                    uint64_t e = ((uint64_t)m2 * (((uint64_t)k << a.log_b) + (a.coset_twiddle ? jg : 0u))) & nmask;
                    if (e != 0) {
                        const uint32_t el = (uint32_t)e & ((1u << a.lo_bits) - 1u), eh = (uint32_t)(e >> a.lo_bits);
                        v = fe_mul_call(v, a.tw_lo[el]);
                        if (eh != 0) v = fe_mul_call(v, a.tw_hi[eh]);
                    }
                }
                if (a.has_scale) v = fe_mul_call(v, a.scale);
                dst[(size_t)k * a.out_stride_k + t] = v;
            });
        });
    }
}

template <int LOGL>
static void launch_ntt_reg(dst_ctx* c, const NttRegArgs& a, size_t tiles, size_t cosets, size_t cols, const char* name, double bytes) {
    constexpr int NT = (1 << (LOGL + NttDigits<LOGL>::log_t)) / 16;
    dim3 g((unsigned)(tiles >> NttDigits<LOGL>::log_t), (unsigned)cosets, (unsigned)cols);
    KScope ks_(c, name, bytes, true);
    hipLaunchKernelGGL(ntt_reg_kernel<LOGL>, g, dim3(NT), 0, c->stream, a);
}
static void dispatch_ntt_reg(dst_ctx* c, uint32_t log_len, const NttRegArgs& a, size_t tiles, size_t cosets, size_t cols, const char* name, double bytes) {
    switch (log_len) {
        case 6: launch_ntt_reg<6>(c, a, tiles, cosets, cols, name, bytes); break;
        case 7: launch_ntt_reg<7>(c, a, tiles, cosets, cols, name, bytes); break;
        case 8: launch_ntt_reg<8>(c, a, tiles, cosets, cols, name, bytes); break;
        case 9: launch_ntt_reg<9>(c, a, tiles, cosets, cols, name, bytes); break;
        case 10: launch_ntt_reg<10>(c, a, tiles, cosets, cols, name, bytes); break;
        case 11: launch_ntt_reg<11>(c, a, tiles, cosets, cols, name, bytes); break;
        default: launch_ntt_reg<12>(c, a, tiles, cosets, cols, name, bytes); break;
    }
}

static void launch_pass_reg(dst_ctx* c, bool pass_b, const fe* src, size_t src_col_stride, size_t src_coset_stride,
                            fe* dst, size_t dst_col_stride, size_t dst_coset_stride, size_t cosets, size_t cols, bool inverse, bool lde, uint32_t skip) {
    const NttPlan& p = c->plan;
    const size_t n1 = (size_t)1 << p.log_n1, n2 = (size_t)1 << p.log_n2;
    NttRegArgs a{};
    a.log_N = c->log_N; a.log_b = c->log_b; a.lo_bits = c->tw_lo_bits;
    a.j0 = lde ? (uint32_t)c->j0 + skip : 0u; a.coset_twiddle = lde ? 1u : 0u;
    for (int j = 0; j < 8; j++) a.c16[j] = inverse ? c->c16i[j] : c->c16f[j];
    a.src = src; a.src_col_stride = src_col_stride; a.src_coset_stride = src_coset_stride;
    a.dst = dst; a.dst_col_stride = dst_col_stride; a.dst_coset_stride = dst_coset_stride;
    a.scale = c->n_inv;
    if (!pass_b) {      // n1-point transforms over the stride-n2 dimension
        a.in_stride_m = n2; a.in_stride_t = 1; a.out_stride_k = n2;
        a.stage_tw = inverse ? c->w1i : c->w1f;
        a.tw_lo = inverse ? c->itw_lo : c->tw_lo; a.tw_hi = inverse ? c->itw_hi : c->tw_hi;
        a.prescale = lde ? c->prescale : nullptr;
        a.has_scale = 0;
        dispatch_ntt_reg(c, p.log_n1, a, n2, cosets, cols, "ntt_pass_a", 16.0 * c->n * cols * (lde ? (1 + cosets) : 2 * cosets));
    } else {            // n2-point transforms over the contiguous dimension, natural-order output
        a.in_stride_m = 1; a.in_stride_t = n2; a.out_stride_k = n1;
        a.stage_tw = inverse ? c->w2i : c->w2f;
        a.tw_lo = nullptr; a.tw_hi = nullptr; a.prescale = nullptr;
        a.has_scale = inverse ? 1u : 0u;
        dispatch_ntt_reg(c, p.log_n2, a, n1, cosets, cols, "ntt_pass_b", 32.0 * c->n * cols * cosets);
    }
}

// tiles a workgroup walks through: as many as keep >= 2048 workgroups in the launch (8 per CU), at most 8
static uint32_t ntt_tiles_per_block(uint32_t tiles, size_t arrays) {
    uint32_t k = 1;
    while (k < 8 && tiles % (2 * k) == 0 && (size_t)(tiles / (2 * k)) * arrays >= 2048) k *= 2;
    return k;
}
// Block order of an extension's first pass (ntt_block): every (coset, register) of a tile group before the next group, so that the
// coefficient tiles (the same for every coset) and the four-step twiddles (the same for every register) are re-read while they are still
// in the XCD's L2: FETCH_SIZE of the 2^20 launches 1.29 against 1.57 GB, time unchanged.  DISTAFF_NTT_ORDER=0: coset-slow order (tests).
static uint32_t ntt_coset_fast(const dst_ctx* c, size_t groups, size_t cosets) {
    const char* e = c->sw("DISTAFF_NTT_ORDER");
    return (cosets > 1 && groups % 8 == 0 && !(e && e[0] == '0')) ? 1u : 0u;
}
static void ntt_raise_lds_limit(dst_ctx* c) {                  // tile + stage twiddles exceed the 64 KiB default
    // contexts of several ranks may run as threads of one process (dst_prove_sharded_local): the once-per-device marks are guarded
    static std::mutex mu;
    static bool raised[64] = {};
    std::lock_guard<std::mutex> lock(mu);
    if (c->device < 0 || c->device >= 64 || raised[c->device]) return;
    (void)hipFuncSetAttribute((const void*)ntt_pass_a<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ntt_pass_a<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ntt_pass_b<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ntt_pass_b<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ntt_pass_a<1024, 8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ntt_pass_a<1024, 8, false, 10, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ntt_pass_a<1024, 8, false, 8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ntt_pass_b<1024, 8, false, 8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ntt_pass_a<512, 4, true, 8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ntt_pass_b<512, 4, true, 8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ntt_pass_b<1024, 8, true, 10, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ntt_pass_b<1024, 8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ntt_pass_a<1024, 8, false, 10, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ntt_pass_b<1024, 8, false, 10, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ntt_pass_a<512, 4, true, 0, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ntt_pass_b<512, 4, true, 0, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);

    raised[c->device] = true;
}
// DISTAFF_NTT_DEBUG=1 prints, once per distinct launch shape, how many workgroups of the instance are resident per CU
static void ntt_report_occupancy(const char* name, bool pass_b, int threads, bool eight, size_t lds) {
    static std::mutex mu;
    static std::map<std::string, bool> seen;
    char key[128]; snprintf(key, sizeof key, "%s/%d/%d/%zu", name, threads, (int)eight, lds);
    std::lock_guard<std::mutex> lock(mu);
    if (seen[key]) return; seen[key] = true;
    int nb = -1;
    hipError_t e;
    if (eight) e = pass_b ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ntt_pass_b<1024, 8, false>, 1024, lds) : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ntt_pass_a<1024, 8, false>, 1024, lds);
    else if (threads == 512) e = pass_b ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ntt_pass_b<512>, 512, lds) : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ntt_pass_a<512>, 512, lds);
    else e = pass_b ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ntt_pass_b<1024>, 1024, lds) : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ntt_pass_a<1024>, 1024, lds);
    fprintf(stderr, "[distaff] %s: %d lanes, %zu B LDS -> %d workgroups per CU (%s)\n", name, threads, lds, nb, hipGetErrorString(e));
}
// two workgroups of 512 lanes per CU while tile + twiddles fit 80 KiB, else one of 1024 lanes (same waves per SIMD)
#define NTT_LDS_TWO_PER_CU (80 * 1024)
static void ntt_launch(dst_ctx* c, bool pass_b, NttArgs& a, size_t groups, size_t cosets, size_t cols, size_t lds, const char* name, double bytes) {
    ntt_raise_lds_limit(c);
    a.groups = (uint32_t)groups; a.cosets = (uint32_t)cosets; a.cols = (uint32_t)cols;
    const dim3 grid((unsigned)(groups * cosets * cols));
    // multiply-adds of the launch: 18 per table-pair multiplication (fe_mul_tw); multiplications per element: every DIT stage 1/2, DIF
    // two-stage rounds 1 each except the last (1/4: the distance-1 stage has no twiddles), plus four-step twiddle / pre-scale / 1/n
    const uint32_t stages = pass_b ? a.log_n2 : a.log_n1;
    double mults = (!pass_b && a.dit) ? 0.5 * stages : ((stages & 1u) ? 0.5 * (stages - 1) : (stages >= 2 ? 0.5 * stages - 0.75 : 0.0));
    if (!pass_b) mults += ((!a.dit && a.prescale != nullptr) ? 1.0 : 0.0) + (NTT_TW4_PAIRS ? 1.0 : 21.0 / 18.0); else if (a.has_scale) mults += 1.0;     // the four-step product: 18 or 21 mads
    if (a.pre) mults += (!pass_b && a.dit) ? 1.0 : 0.5;        // register pre-stage: c * x[m + len] in both halves of a coset DIT, the twiddle of the odd half otherwise
    const double elements = (double)groups * a.tiles_per_block * ((size_t)1 << a.tile) * ((size_t)1 << stages) * cosets * cols;
    KScope ks_(c, name, bytes, true, 18.0 * mults * elements);
    const char* wv = c->sw("DISTAFF_NTT_WAVES");
    const bool two = lds <= NTT_LDS_TWO_PER_CU, eight = two && (wv ? wv[0] == '8' : stages >= 10);
    const bool any_shape = c->sw_is("DISTAFF_NTT_FIXED", "0");
    const bool fixed = stages == 10 && a.tile == 2 && !any_shape;      // 1024 x 4 tiles (n = 2^20) have their own instances,
    const bool fixed84 = stages == 8 && a.tile == 4 && !any_shape;     // and so have 256 x 16 tiles (n = 2^16, the first two passes of three-pass plans)
    if (a.debug & 1u) ntt_report_occupancy(name, pass_b, eight || !two ? 1024 : 512, eight, lds);
    if (a.pre) {
        // register pre-stage instances (NttArgs::pre): the 1024 x 4 shape of n = 2^21 / 2^22 compiled for the shape, any other shape (tests) from the arguments
        if (pass_b && fixed) hipLaunchKernelGGL((ntt_pass_b<1024, 8, false, 10, 2, 1>), grid, dim3(1024), lds, c->stream, a, a.src, a.dst);
        else if (!pass_b && fixed) hipLaunchKernelGGL((ntt_pass_a<1024, 8, false, 10, 2, 1>), grid, dim3(1024), lds, c->stream, a, a.src, a.dst);
        else if (pass_b) hipLaunchKernelGGL((ntt_pass_b<512, 4, true, 0, 0, 1>), grid, dim3(512), lds, c->stream, a, a.src, a.dst);
        else hipLaunchKernelGGL((ntt_pass_a<512, 4, true, 0, 0, 1>), grid, dim3(512), lds, c->stream, a, a.src, a.dst);
        return;
    }
    if (two) {
        // 1024-point tiles (five LDS rounds per tile): two workgroups of 1024 lanes = 8 waves per SIMD, 64 registers, no register prefetch -- the
        // other workgroup's rounds cover a workgroup's loads (measured 20.2 against 20.55 ms of extension per 2^20 proof, same box); shorter
        // tiles stay with 512 lanes + prefetch (2^16: 0.96 against 1.03 ms).  DISTAFF_NTT_WAVES=4|8 forces one (the tests run both).
        // the fixed-shape second pass needs 44 registers: its 60-register form with the register prefetch still runs at 8 waves (9.03 -> 8.85 ms
        // per proof); the first pass with the prefetch spills (12.0 -> 12.3 ms) and stays without
        if (pass_b && eight && fixed) hipLaunchKernelGGL((ntt_pass_b<1024, 8, true, 10, 2>), grid, dim3(1024), lds, c->stream, a, a.src, a.dst);
        else if (!pass_b && eight && fixed) hipLaunchKernelGGL((ntt_pass_a<1024, 8, false, 10, 2>), grid, dim3(1024), lds, c->stream, a, a.src, a.dst);
        else if (pass_b && eight && fixed84) hipLaunchKernelGGL((ntt_pass_b<1024, 8, false, 8, 4>), grid, dim3(1024), lds, c->stream, a, a.src, a.dst);
        else if (!pass_b && eight && fixed84) hipLaunchKernelGGL((ntt_pass_a<1024, 8, false, 8, 4>), grid, dim3(1024), lds, c->stream, a, a.src, a.dst);
        else if (pass_b && fixed84) hipLaunchKernelGGL((ntt_pass_b<512, 4, true, 8, 4>), grid, dim3(512), lds, c->stream, a, a.src, a.dst);
        else if (!pass_b && fixed84) hipLaunchKernelGGL((ntt_pass_a<512, 4, true, 8, 4>), grid, dim3(512), lds, c->stream, a, a.src, a.dst);
        else if (pass_b && eight) hipLaunchKernelGGL((ntt_pass_b<1024, 8, false>), grid, dim3(1024), lds, c->stream, a, a.src, a.dst);
        else if (pass_b) hipLaunchKernelGGL(ntt_pass_b<512>, grid, dim3(512), lds, c->stream, a, a.src, a.dst);
        else if (eight) hipLaunchKernelGGL((ntt_pass_a<1024, 8, false>), grid, dim3(1024), lds, c->stream, a, a.src, a.dst);
        else hipLaunchKernelGGL(ntt_pass_a<512>, grid, dim3(512), lds, c->stream, a, a.src, a.dst);
    } else {
        if (pass_b) hipLaunchKernelGGL(ntt_pass_b<1024>, grid, dim3(1024), lds, c->stream, a, a.src, a.dst);
        else hipLaunchKernelGGL(ntt_pass_a<1024>, grid, dim3(1024), lds, c->stream, a, a.src, a.dst);
    }
}
// First pass of an extension: coset DIT (the pre-scale costs nothing, but the workgroup holds n1 - 1 twiddle pairs of ITS coset) while
// tile + pairs fit the 80 KiB that let two workgroups share a CU; otherwise pre-scale + DIF with the n1/2 shared stage twiddles (one
// more multiplication per element, two workgroups per CU: measured 14.5 against 15.0 ms per proof for the 1024-point tiles of n = 2^20).
// DISTAFF_NTT_DIF=1 / 0 forces one or the other (the tests run both).
// returns 0: pre-scale + DIF, 1: coset DIT with the whole table in LDS, 2: coset DIT whose last-stage twiddles (half of the table) are read
// from global memory (tile + the other half fit 80 KiB)
static int ntt_first_pass_mode(const dst_ctx* c, size_t n1, size_t tile) {
    if (const char* e = c->sw("DISTAFF_NTT_DIF")) return e[0] == '0' ? 1 : e[0] == '2' ? 2 : 0;
    if (n1 * tile * sizeof(fe) + n1 * sizeof(fe_tw) <= NTT_LDS_TWO_PER_CU) return 1;
    if (n1 * tile * sizeof(fe) + (n1 / 2) * sizeof(fe_tw) <= NTT_LDS_TWO_PER_CU) return 2;
    return 0;
}
static NttArgs ntt_common_args(dst_ctx* c, bool inverse, bool lde, uint32_t skip) {
    NttArgs a{};
    a.log_N = c->log_N; a.log_b = c->log_b;
    a.j0 = lde ? (uint32_t)c->j0 + skip : 0u;
    a.scale = c->n_inv_tw;
    { const char* dbg = c->sw("DISTAFF_NTT_DEBUG"); a.debug = dbg ? (uint32_t)atoi(dbg) : 0u; }
    (void)inverse;
    return a;
}
static void launch_pass_lds(dst_ctx* c, bool pass_b, const fe* src, size_t src_col_stride, size_t src_coset_stride,
                            fe* dst, size_t dst_col_stride, size_t dst_coset_stride, size_t cosets, size_t cols, bool inverse, bool lde, uint32_t skip) {
    const NttPlan& p = c->plan;
    NttArgs a = ntt_common_args(c, inverse, lde, skip);
    a.log_n1 = p.log_n1; a.log_n2 = p.log_n2;
    a.prescale = lde ? c->prescale : nullptr;
    a.has_scale = inverse ? 1u : 0u;
    a.src = src; a.src_col_stride = src_col_stride; a.src_coset_stride = src_coset_stride;
    a.dst = dst; a.dst_col_stride = dst_col_stride; a.dst_coset_stride = dst_coset_stride;
    if (!pass_b) {
        a.tw4 = lde ? c->tw4_lde + (size_t)skip * c->n : (inverse ? c->tw4_inv : c->tw4_fwd); a.tw4_coset_stride = lde ? c->n : 0;
        a.stage_tw = inverse ? c->w1i : c->w1f; a.tile = (uint32_t)__builtin_ctz(p.tile_a);
        a.pre = p.pre_a; a.pre_tw = inverse ? c->w1pi : c->w1pf;
        a.log_n1 = p.log_n1 - p.pre_a;                                 // the kernel's LDS transform; with the pre-stage the pass covers twice that
        const size_t n1 = (size_t)1 << a.log_n1;
        int mode = lde ? ntt_first_pass_mode(c, n1, p.tile_a) : 0;
        if (p.pre_a && lde && mode == 0) mode = 2;                     // the pre-stage is written for the coset DIT (tiles of at most 1024 x 4: always fits)
        a.dit = mode ? 1u : 0u; a.dit_last = mode == 2 ? c->dit_last : nullptr;
        const size_t lds_a = n1 * p.tile_a * sizeof(fe) + (mode == 1 ? n1 : n1 / 2) * sizeof(fe_tw);
        const uint32_t tiles = (1u << p.log_n2) / p.tile_a;
        a.tiles_per_block = ntt_tiles_per_block(tiles, (cosets * cols) << p.pre_a);
        a.coset_fast = ntt_coset_fast(c, (size_t)(tiles / a.tiles_per_block) << p.pre_a, cosets);
        ntt_launch(c, false, a, (size_t)(tiles / a.tiles_per_block) << p.pre_a, cosets, cols, lds_a, "ntt_pass_a", 16.0 * c->n * cols * (lde ? (1 + cosets) : 2 * cosets));
    } else {
        a.stage_tw = inverse ? c->w2i : c->w2f; a.tile = (uint32_t)__builtin_ctz(p.tile_b);
        a.pre = p.pre_b; a.pre_tw = inverse ? c->w2pi : c->w2pf;
        a.log_n2 = p.log_n2 - p.pre_b;
        a.src_row_stride = (size_t)1 << p.log_n2; a.dst_k_stride = (size_t)1 << p.log_n1; a.batch_log = 0; a.src_batch_stride = a.dst_batch_stride = 0;
        const size_t n2 = (size_t)1 << a.log_n2;
        const size_t lds_b = n2 * p.tile_b * sizeof(fe) + (n2 / 2) * sizeof(fe_tw);
        const uint32_t tiles = (1u << p.log_n1) / p.tile_b;
        a.tiles_per_block = ntt_tiles_per_block(tiles, (cosets * cols) << p.pre_b);
        ntt_launch(c, true, a, (size_t)(tiles / a.tiles_per_block) << p.pre_b, cosets, cols, lds_b, "ntt_pass_b", 32.0 * c->n * cols * cosets);
    }
}

// n = n1 * nm * n3 (each <= 2^8) in three HBM passes with 16-column tiles, all on the LDS-family kernels:
//   pass 1 (ntt_pass_a, shape n1 x n/n1): as the first pass of a two-pass plan (coset pre-scale, twiddle w_N^(m'*(B*k1+j)));
//   pass 2 (ntt_pass_a, shape 2^8 x n3 on each of the cosets*2^8 rows of n/2^8 points): twiddle w_{n/2^8}^(k2*m3);
//   pass 3 (ntt_pass_b, n3 points, tile = 16 adjacent k1, one batch per k2): natural-order store at k1 + 2^8*(k2 + 2^8*k3).
static void launch_three_pass(dst_ctx* c, const fe* src, size_t src_col_stride, size_t src_coset_stride,
                              fe* dst, size_t dst_col_stride, size_t dst_coset_stride,
                              size_t cosets, size_t cols, bool inverse, bool lde, uint32_t skip) {
    const NttPlan& p = c->plan;
    const uint32_t log_mid = p.log_n2 - p.log_n3;                      // 8
    const size_t n = c->n, nrow = (size_t)1 << p.log_n2, n3 = (size_t)1 << p.log_n3, n1 = (size_t)1 << p.log_n1;
    NttArgs a = ntt_common_args(c, inverse, lde, skip);
    // pass 1: src -> tmp
    a.log_n1 = p.log_n1; a.log_n2 = p.log_n2; a.tile = (uint32_t)__builtin_ctz(p.tile_a);
    a.prescale = lde ? c->prescale : nullptr; a.has_scale = 0;
    a.tw4 = lde ? c->tw4_lde + (size_t)skip * n : (inverse ? c->tw4_inv : c->tw4_fwd); a.tw4_coset_stride = lde ? n : 0;
    a.stage_tw = inverse ? c->w1i : c->w1f;
    const int mode = lde ? ntt_first_pass_mode(c, n1, p.tile_a) : 0;
    a.dit = mode ? 1u : 0u; a.dit_last = mode == 2 ? c->dit_last : nullptr;
    a.src = src; a.src_col_stride = src_col_stride; a.src_coset_stride = src_coset_stride;
    a.dst = c->tmp; a.dst_col_stride = n * cosets; a.dst_coset_stride = n;
    {
        const uint32_t tiles = (uint32_t)(nrow / p.tile_a);
        a.tiles_per_block = ntt_tiles_per_block(tiles, cosets * cols);
        a.coset_fast = ntt_coset_fast(c, tiles / a.tiles_per_block, cosets);
        const size_t lds = n1 * p.tile_a * sizeof(fe) + (mode == 1 ? n1 : n1 / 2) * sizeof(fe_tw);
        ntt_launch(c, false, a, tiles / a.tiles_per_block, cosets, cols, lds, "ntt_pass_a", 16.0 * n * cols * (lde ? (1 + cosets) : 2 * cosets));
    }
    // pass 2: tmp -> tmp2, every (coset, k1) row of nrow points is an array of shape 2^log_mid x n3
    a.log_n1 = log_mid; a.log_n2 = p.log_n3; a.tile = (uint32_t)__builtin_ctz(p.tile_m);
    a.j0 = 0; a.prescale = nullptr; a.dit = 0; a.coset_fast = 0;
    a.tw4 = inverse ? c->tw4_row_inv : c->tw4_row_fwd; a.tw4_coset_stride = 0;
    a.stage_tw = inverse ? c->w2i : c->w2f;
    a.src = c->tmp; a.src_col_stride = n * cosets; a.src_coset_stride = nrow;
    a.dst = c->tmp2; a.dst_col_stride = n * cosets; a.dst_coset_stride = nrow;
    {
        const uint32_t tiles = (uint32_t)(n3 / p.tile_m);
        const size_t rows = cosets * n1;
        a.tiles_per_block = ntt_tiles_per_block(tiles, rows * cols);
        const size_t lds = ((size_t)1 << log_mid) * p.tile_m * sizeof(fe) + (((size_t)1 << log_mid) / 2) * sizeof(fe_tw);
        ntt_launch(c, false, a, tiles / a.tiles_per_block, rows, cols, lds, "ntt_pass_mid", 32.0 * n * cols * cosets);
    }
    // pass 3: tmp2 -> dst
    a.log_n1 = p.log_n1; a.log_n2 = p.log_n3; a.tile = (uint32_t)__builtin_ctz(p.tile_b);
    a.tw4 = nullptr; a.has_scale = inverse ? 1u : 0u;
    a.stage_tw = inverse ? c->w3i : c->w3f;
    a.src = c->tmp2; a.src_col_stride = n * cosets; a.src_coset_stride = n;
    a.dst = dst; a.dst_col_stride = dst_col_stride; a.dst_coset_stride = dst_coset_stride;
    a.src_row_stride = nrow; a.src_batch_stride = n3; a.batch_log = log_mid;
    a.dst_k_stride = n1 << log_mid; a.dst_batch_stride = n1;
    {
        const uint32_t tiles = (uint32_t)(n1 / p.tile_b);
        a.tiles_per_block = ntt_tiles_per_block(tiles, ((size_t)cosets << log_mid) * cols);
        const size_t lds = n3 * p.tile_b * sizeof(fe) + (n3 / 2) * sizeof(fe_tw);
        ntt_launch(c, true, a, (size_t)(tiles / a.tiles_per_block) << log_mid, cosets, cols, lds, "ntt_pass_b", 32.0 * n * cols * cosets);
    }
}

// skip: number of leading local cosets left out (their outputs are produced elsewhere); dst points at the first coset computed
static void launch_two_pass(dst_ctx* c, const fe* src, size_t src_col_stride, size_t src_coset_stride,
                            fe* dst, size_t dst_col_stride, size_t dst_coset_stride,
                            size_t cosets, size_t cols, bool inverse, bool lde, uint32_t skip = 0) {
    if (c->plan.log_n3) { launch_three_pass(c, src, src_col_stride, src_coset_stride, dst, dst_col_stride, dst_coset_stride, cosets, cols, inverse, lde, skip); return; }
    // pass A: src -> tmp, pass B: tmp -> dst
    (c->plan.reg_a ? launch_pass_reg : launch_pass_lds)(c, false, src, src_col_stride, src_coset_stride, c->tmp, c->n * cosets, c->n, cosets, cols, inverse, lde, skip);
    (c->plan.reg_b ? launch_pass_reg : launch_pass_lds)(c, true, c->tmp, c->n * cosets, c->n, dst, dst_col_stride, dst_coset_stride, cosets, cols, inverse, lde, skip);
}

// ---- four-step twiddle tables -------------------------------------------------------------------------------------------------------
// out[coset][k1][m2] = w_N^(m2 * ((k1 << log_b) + j)) for a transform of 2^log_n points whose inner dimension has 2^log_n2 points;
// a sub-transform of length n' = n / 2^s is expressed with log_b + s (its root is w_N^(B * 2^s))
__global__ void twiddle_table_kernel(tw4_t* out, const fe* tw_lo, const fe* tw_hi, uint32_t lo_bits, uint32_t log_n2, uint32_t log_n, uint32_t log_b,
                                     uint32_t log_N, uint32_t j0, uint32_t coset_twiddle) {
    const size_t n = (size_t)1 << log_n;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t k1 = i >> log_n2, m2 = i & (((size_t)1 << log_n2) - 1);
    const uint64_t jg = coset_twiddle ? j0 + blockIdx.y : 0;
    const uint64_t e = (m2 * ((k1 << log_b) + jg)) & (((uint64_t)1 << log_N) - 1);
#if NTT_TW4_PAIRS
    out[(size_t)blockIdx.y * n + i] = fe_tw_make(dom_pow(tw_lo, tw_hi, lo_bits, e));
#else
    out[(size_t)blockIdx.y * n + i] = dom_pow(tw_lo, tw_hi, lo_bits, e);
#endif
}
int k_build_twiddle_tables(dst_ctx* c) {
    const NttPlan& p = c->plan;
    dim3 g((unsigned)((c->n + 255) / 256), (unsigned)c->Bc);
    hipLaunchKernelGGL(twiddle_table_kernel, g, dim3(256), 0, c->stream, c->tw4_lde, c->tw_lo, c->tw_hi, c->tw_lo_bits, p.log_n2, c->log_n, c->log_b, c->log_N, (uint32_t)c->j0, 1u);
    g.y = 1;
    hipLaunchKernelGGL(twiddle_table_kernel, g, dim3(256), 0, c->stream, c->tw4_fwd, c->tw_lo, c->tw_hi, c->tw_lo_bits, p.log_n2, c->log_n, c->log_b, c->log_N, 0u, 0u);
    hipLaunchKernelGGL(twiddle_table_kernel, g, dim3(256), 0, c->stream, c->tw4_inv, c->itw_lo, c->itw_hi, c->tw_lo_bits, p.log_n2, c->log_n, c->log_b, c->log_N, 0u, 0u);
    if (p.log_n3) {       // middle pass of a three-pass plan: rows of n' = n2 points, w_{n'}^(k2*m3) = w_N^(m3 * (k2 << (log_b + log_n1)))
        dim3 gr((unsigned)((((size_t)1 << p.log_n2) + 255) / 256), 1u);
        hipLaunchKernelGGL(twiddle_table_kernel, gr, dim3(256), 0, c->stream, c->tw4_row_fwd, c->tw_lo, c->tw_hi, c->tw_lo_bits, p.log_n3, p.log_n2, c->log_b + p.log_n1, c->log_N, 0u, 0u);
        hipLaunchKernelGGL(twiddle_table_kernel, gr, dim3(256), 0, c->stream, c->tw4_row_inv, c->itw_lo, c->itw_hi, c->tw_lo_bits, p.log_n3, p.log_n2, c->log_b + p.log_n1, c->log_N, 0u, 0u);
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    return DST_OK;
}

// how many (coset x column) size-n arrays fit in c->tmp
static size_t tmp_capacity_arrays(const dst_ctx* c) { return c->Bc * c->tmp_regs; }

void k_intt_columns(dst_ctx* c, const fe* src, size_t src_stride, fe* dst, size_t ncols) {
    size_t cap = tmp_capacity_arrays(c);
    for (size_t done = 0; done < ncols;) {
        size_t cols = ncols - done < cap ? ncols - done : cap;
        launch_two_pass(c, src + done * src_stride, src_stride, 0, dst + done * c->n, c->n, 0, 1, cols, true, false);
        done += cols;
    }
}

void k_lde_columns(dst_ctx* c, const fe* polys, fe* lde, size_t ncols) {
    // coset 0 of the extension is the trace itself (T(w_n^k) for the interpolant T): when this rank owns it and the polynomials are
    // the interpolated trace, it is copied instead of transformed (1/B of the extension work)
    const bool of_trace = polys >= c->polys && polys < c->polys + c->W * c->n;          // a group of the interpolated trace registers
    const uint32_t skip = (c->j0 == 0 && of_trace && c->Bc > 1 && !c->trace_owned_only) ? 1u : 0u;      // owned-only: this rank holds 1/world of the trace registers, coset 0 is transformed like the others
    if (skip && c->trace != c->lde)         // a context that owns coset 0 normally keeps the trace in those slots already (ctx.h: trace_stride)
        (void)hipMemcpy2DAsync(lde, c->Bc * c->n * sizeof(fe), c->trace + ((polys - c->polys) / c->n) * c->trace_stride, c->trace_stride * sizeof(fe), c->n * sizeof(fe), ncols, hipMemcpyDeviceToDevice, c->stream);
    // launch granularity: `bcols` registers x `bcos` cosets per pair of passes (the staging buffer holds tmp_capacity_arrays arrays)
    size_t bcols = tmp_capacity_arrays(c) / c->Bc, bcos = c->Bc - skip;
    if (const char* e = c->sw("DISTAFF_LDE_BATCH")) { unsigned x = 0, y = 0; if (sscanf(e, "%u,%u", &x, &y) == 2 && x >= 1 && y >= 1 && (size_t)x * y <= tmp_capacity_arrays(c)) { bcols = x; bcos = y; } }
    for (size_t done = 0; done < ncols;) {
        const size_t cols = ncols - done < bcols ? ncols - done : bcols;
        for (size_t s = skip; s < c->Bc;) {
            const size_t cc = c->Bc - s < bcos ? c->Bc - s : bcos;
            launch_two_pass(c, polys + done * c->n, c->n, 0, lde + done * c->Bc * c->n + s * c->n, c->Bc * c->n, c->n, cc, cols, false, true, (uint32_t)s);
            s += cc;
        }
        done += cols;
    }
}

// ---- 8n coefficients -> coset-major evaluations -------------------------------------------------------------------------
// d_j[m0] = w_N^(j*m0) * sum_{m1<8} c[m0 + n*m1] * w_B^(j*m1); followed by a plain size-n NTT per coset.
// one lane per m0 walks through all local cosets: the eight coefficients are read once, the factors w_B^(j*m1) of every coset come
// from LDS, and w_N^(j*m0) advances by one multiplication per coset
__global__ void __launch_bounds__(256) fold8_kernel(const fe* __restrict__ poly, fe* __restrict__ out, const fe* tw_lo, const fe* tw_hi, uint32_t lo_bits,
                                                    uint32_t log_n, uint32_t log_N, uint32_t j0, uint32_t cosets) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fold_smem[];
    fe* cj = reinterpret_cast<fe*>(fold_smem);                      // [cosets][8]
    const size_t n = (size_t)1 << log_n;
    const uint64_t nmask = ((uint64_t)1 << log_N) - 1;
    for (uint32_t i = threadIdx.x; i < cosets * 8; i += blockDim.x)
        cj[i] = dom_pow(tw_lo, tw_hi, lo_bits, ((uint64_t)(j0 + (i >> 3)) * (i & 7) << log_n) & nmask);
    __syncthreads();
    const size_t m0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m0 >= n) return;
    fe c[8];
#pragma unroll
    for (int m1 = 0; m1 < 8; m1++) c[m1] = poly[m0 + n * m1];
    const fe step = dom_pow(tw_lo, tw_hi, lo_bits, m0 & nmask);                         // w_N^m0
    fe t = dom_pow(tw_lo, tw_hi, lo_bits, ((uint64_t)j0 * m0) & nmask);                 // w_N^(j*m0) for the first local coset
    for (uint32_t jl = 0; jl < cosets; jl++) {
        fe_acc A; fe_acc_zero(A);                        // eight products, one reduction
        fe_acc_add(A, c[0]);
#pragma unroll
        for (int m1 = 1; m1 < 8; m1++) fe_acc_mac(A, c[m1], cj[jl * 8 + m1]);
        out[(size_t)jl * n + m0] = fe_mul(fe_acc_reduce(A), t);
        t = fe_mul(t, step);
    }
}

// The same folding when a context owns ALL B = 8 S cosets (one GPU): with j = j1 S + js, w_B^(j m1) = w_8^(j1 m1) * w_B^(js m1), so for each of the
// S residues js the eight coefficients are scaled by w_B^(js m1) and ONE 8-point DFT over m1 gives the sums of the eight cosets j1 S + js:
// 5 + 7 multiplications per eight cosets instead of 56 multiply-accumulates and 8 reductions (fold8_kernel 0.44 -> 0.26 ms at 2^20, blowup 32).
template <int S>
__global__ void __launch_bounds__(256) fold8_dft_kernel(const fe* __restrict__ poly, fe* __restrict__ out, const fe* tw_lo, const fe* tw_hi, uint32_t lo_bits,
                                                        uint32_t log_n, uint32_t log_N) {
    __shared__ fe pre[S * 8];                                        // w_B^(js * m1) = w_N^(n * js * m1)
    const size_t n = (size_t)1 << log_n;
    const uint64_t nmask = ((uint64_t)1 << log_N) - 1;
    for (uint32_t i = threadIdx.x; i < S * 8; i += blockDim.x) pre[i] = dom_pow(tw_lo, tw_hi, lo_bits, ((uint64_t)(i >> 3) * (i & 7) << log_n) & nmask);
    __syncthreads();
    const size_t m0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m0 >= n) return;
    fe c[8];
#pragma unroll
    for (int m1 = 0; m1 < 8; m1++) c[m1] = poly[m0 + n * m1];
    const fe r1 = dom_pow(tw_lo, tw_hi, lo_bits, (uint64_t)1 << (log_N - 3));          // w_8
    const fe r2 = fe_sqr(r1), r3 = fe_mul(r2, r1);
    const fe step = dom_pow(tw_lo, tw_hi, lo_bits, m0 & nmask);                         // w_N^m0
    const fe step_s = dom_pow(tw_lo, tw_hi, lo_bits, ((uint64_t)S * m0) & nmask);       // w_N^(S m0): from coset j to j + S
    fe t_js = fe_one();                                                                 // w_N^(js m0)
#pragma unroll 1
    for (uint32_t js = 0; js < S; js++) {
        fe x[8];
        x[0] = c[0];
#pragma unroll
        for (int m1 = 1; m1 < 8; m1++) x[m1] = js ? fe_mul(c[m1], pre[js * 8 + m1]) : c[m1];
        // 8-point DFT with root w_8, radix-2 DIF: X[j1] = sum_m1 x[m1] w_8^(j1 m1); stage outputs in bit-reversed positions
        fe a0, a1, a2, a3, a4, a5, a6, a7;
        fe_addsub(x[0], x[4], a0, a4); fe_addsub(x[1], x[5], a1, a5); fe_addsub(x[2], x[6], a2, a6); fe_addsub(x[3], x[7], a3, a7);
        a5 = fe_mul(a5, r1); a6 = fe_mul(a6, r2); a7 = fe_mul(a7, r3);
        fe b0, b1, b2, b3, b4, b5, b6, b7;
        fe_addsub(a0, a2, b0, b2); fe_addsub(a1, a3, b1, b3); fe_addsub(a4, a6, b4, b6); fe_addsub(a5, a7, b5, b7);
        b3 = fe_mul(b3, r2); b7 = fe_mul(b7, r2);
        fe X[8];
        fe_addsub(b0, b1, X[0], X[4]); fe_addsub(b2, b3, X[2], X[6]); fe_addsub(b4, b5, X[1], X[5]); fe_addsub(b6, b7, X[3], X[7]);
        fe t = t_js;
#pragma unroll
        for (uint32_t j1 = 0; j1 < 8; j1++) {
            out[((size_t)j1 * S + js) * n + m0] = fe_mul(X[j1], t);
            if (j1 < 7) t = fe_mul(t, step_s);
        }
        t_js = fe_mul(t_js, step);
    }
}

void k_lde_fold8(dst_ctx* c, const fe* poly8n, fe* out) {
    // stage the folded inputs in `out` itself, then transform each coset in place (pass A reads out, pass B writes out)
    dim3 g((unsigned)((c->n + 255) / 256));
    const char* e = c->sw("DISTAFF_FOLD8_DFT");
    const bool all_cosets = c->Bc == c->B && c->j0 == 0 && !(e && e[0] == '0');
    const double bytes = 16.0 * c->n * (8 + c->Bc);
#define FOLD8_DFT(S_) { KScope ks_(c, "fold8_dft_kernel", bytes); hipLaunchKernelGGL(fold8_dft_kernel<S_>, g, dim3(256), 0, c->stream, poly8n, out, c->tw_lo, c->tw_hi, c->tw_lo_bits, c->log_n, c->log_N); }
    if (all_cosets && c->B == 16) FOLD8_DFT(2)
    else if (all_cosets && c->B == 32) FOLD8_DFT(4)
    else if (all_cosets && c->B == 64) FOLD8_DFT(8)
    else if (all_cosets && c->B == 128) FOLD8_DFT(16)
    else if (all_cosets && c->B == 256) FOLD8_DFT(32)
    else { KScope ks_(c, "fold8_kernel", bytes); hipLaunchKernelGGL(fold8_kernel, g, dim3(256), c->Bc * 8 * sizeof(fe), c->stream, poly8n, out, c->tw_lo, c->tw_hi, c->tw_lo_bits, c->log_n, c->log_N, (uint32_t)c->j0, (uint32_t)c->Bc); }
#undef FOLD8_DFT
    launch_two_pass(c, out, 0, c->n, out, 0, c->n, c->Bc, 1, false, false);
}

// ---- inverse transform of size 8n from coset-major input ------------------------------------------------------------------
// v[q][k] = V(w_8n^(8k+q)); c[m0 + n*m1] = (1/8n) sum_q w_8^(-q*m1) * w_8n^(-q*m0) * (sum_k v[q][k] w_n^(-k*m0)).
// X[m1] = (1/8) sum_q w_8^(-q*m1) * w_8n^(-q*m0) * work[q][m0], m1 < 8: the 8n coefficients out[m0 + n*m1] of residue class m0
__device__ __forceinline__ void cross8_point(const fe* __restrict__ work, size_t m0, size_t n, const fe* itw_lo, const fe* itw_hi, uint32_t lo_bits,
                                             uint32_t log_N, uint32_t log_b, const fe& eight_inv, fe (&X)[8]) {
    const uint64_t nmask = ((uint64_t)1 << log_N) - 1;
    fe x[8];
#pragma unroll
    for (uint32_t q = 0; q < 8; q++) {
        fe v = work[(size_t)q * n + m0];
        uint64_t e = (((uint64_t)q * m0) << (log_b - 3)) & nmask;          // w_8n = w_N^(B/8)
        x[q] = e ? fe_mul(v, dom_pow(itw_lo, itw_hi, lo_bits, e)) : v;
    }
    // 8-point DFT with root r = w_8^-1 = w_N^-(N/8): X[m1] = sum_q x[q] r^(q*m1)
    fe r1 = dom_pow(itw_lo, itw_hi, lo_bits, (uint64_t)1 << (log_N - 3));
    fe r2 = fe_sqr(r1), r3 = fe_mul(r2, r1);
    // DIF radix-2, three stages
    fe a0 = fe_add(x[0], x[4]), a4 = fe_sub(x[0], x[4]);
    fe a1 = fe_add(x[1], x[5]), a5 = fe_mul(fe_sub(x[1], x[5]), r1);
    fe a2 = fe_add(x[2], x[6]), a6 = fe_mul(fe_sub(x[2], x[6]), r2);
    fe a3 = fe_add(x[3], x[7]), a7 = fe_mul(fe_sub(x[3], x[7]), r3);
    fe b0 = fe_add(a0, a2), b2 = fe_sub(a0, a2);
    fe b1 = fe_add(a1, a3), b3 = fe_mul(fe_sub(a1, a3), r2);
    fe b4 = fe_add(a4, a6), b6 = fe_sub(a4, a6);
    fe b5 = fe_add(a5, a7), b7 = fe_mul(fe_sub(a5, a7), r2);
    X[0] = fe_mul(fe_add(b0, b1), eight_inv); X[4] = fe_mul(fe_sub(b0, b1), eight_inv);
    X[2] = fe_mul(fe_add(b2, b3), eight_inv); X[6] = fe_mul(fe_sub(b2, b3), eight_inv);
    X[1] = fe_mul(fe_add(b4, b5), eight_inv); X[5] = fe_mul(fe_sub(b4, b5), eight_inv);
    X[3] = fe_mul(fe_add(b6, b7), eight_inv); X[7] = fe_mul(fe_sub(b6, b7), eight_inv);
}
__global__ void cross8_kernel(const fe* __restrict__ work, fe* __restrict__ out, const fe* itw_lo, const fe* itw_hi, uint32_t lo_bits,
                              uint32_t log_n, uint32_t log_N, uint32_t log_b, fe eight_inv) {
    const size_t n = (size_t)1 << log_n;
    size_t m0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m0 >= n) return;
    fe X[8];
    cross8_point(work, m0, n, itw_lo, itw_hi, lo_bits, log_N, log_b, eight_inv, X);
#pragma unroll
    for (int m1 = 0; m1 < 8; m1++) out[m0 + (size_t)m1 * n] = X[m1];
}

// ---- combine_polys in one pass (constraint_table.rs:54-88) -----------------------------------------------------------------------------
// The constraint polynomial is  I(x) / (x - 1)  +  F(x) / (x - x_last)  +  T(x) * (x - x_last) / (x^n - 1)  (8n coefficients).
//  * T comes from the inverse-transformed evaluation cosets `work[8][n]`: the 8-point step across cosets gives, per residue class m0,
//    the eight coefficients t[m0 + n m1]; the division by (x^n - 1) / (x - x_last) (polynom.rs:202-236) needs the stride-n suffix sums
//    S_r[k] = sum_{s >= k} t[r + n s] of the classes m0 and m0 - 1:  out[m0 + n m1] = S_{m0-1}[m1 + 1] - x_last * S_{m0}[m1 + 1]
//    (class n - 1 one level down for m0 = 0; nothing above 7n) -- a lane's left neighbour hands its sums over through LDS.
//  * I = A + x^p A', F = C + x^p C' with p = 6n + 2 and A, A', C, C' of n coefficients each (linear combinations of the trace polynomials,
//    dst_internal_boundary_quotients): the quotients of the sparse 8n-coefficient polynomials follow from the n-coefficient quotients.
//    With sA[1 + i] = sum_{t > i} A_t, sA'[0] = sum A' (exclusive suffix sums of the arrays extended by a leading zero) and
//    eC[1 + i] = sum_{t > i} C_t b^(t - i - 1), eC'[0] = C'(b) for b = x_last:
//        i < n:            sA[1 + i] + sA'[0]  +  eC[1 + i] + b^(p - 1 - i) eC'[0]
//        n <= i < p:       sA'[0]              +  b^(p - 1 - i) eC'[0]
//        p <= i < p + n:   sA'[1 + i - p]      +  eC'[1 + i - p]
//    and b^(p - 1 - i) = w_n^(m0 - 1) for every i of class m0 (b has order n).
// The reference's sequence -- two more 8n-point inverse transforms, three divisions and two additions over 8n coefficients each -- is
// DISTAFF_COMBINE=steps (api.hip; the tests compare both).
struct CombineArgs {
    const fe* work; fe* cpoly;
    const fe *sA, *sAp, *eC, *eCp;                 // n + 1 entries each, or all null: transition part only
    const fe *itw_lo, *itw_hi, *tw_lo, *tw_hi;
    uint32_t lo_bits, log_n, log_N, log_b;
    fe eight_inv, x_last;
};
#define COMBINE_THREADS 256
__global__ void __launch_bounds__(COMBINE_THREADS) combine_fused_kernel(CombineArgs a) {
    __shared__ fe S[COMBINE_THREADS][8];
    const size_t n = (size_t)1 << a.log_n;
    const uint32_t t = threadIdx.x;
    // lane t of block b: class m0 = 255 b + t - 1; lane 0 only serves lane 1 as its left neighbour (class n - 1 for the first block)
    const size_t m_plus = (size_t)blockIdx.x * (COMBINE_THREADS - 1) + t;
    const bool inside = m_plus <= n;                             // m0 = m_plus - 1 in [-1, n - 1]
    const size_t m0 = inside ? (m_plus + n - 1) & (n - 1) : 0;
    fe X[8], Sx[9];
    cross8_point(a.work, m0, n, a.itw_lo, a.itw_hi, a.lo_bits, a.log_N, a.log_b, a.eight_inv, X);
    Sx[8] = fe_zero();
#pragma unroll
    for (int k = 7; k >= 0; k--) Sx[k] = k == 7 ? X[7] : fe_add(X[k], Sx[k + 1]);
#pragma unroll
    for (int k = 0; k < 8; k++) S[t][k] = Sx[k];
    __syncthreads();
    if (t == 0 || m_plus > n) return;
    const bool first = m0 == 0;
    fe term = fe_zero();                                         // what every i < p of this class receives from the boundary quotients
    if (a.sA) {
        const fe P = dom_pow(a.tw_lo, a.tw_hi, a.lo_bits, (uint64_t)((m0 + n - 1) & (n - 1)) << a.log_b);      // w_n^(m0 - 1)
        term = fe_add(a.sAp[0], fe_mul(P, a.eCp[0]));
    }
#pragma unroll
    for (int m1 = 0; m1 < 8; m1++) {
        fe v;
        if (m1 < 7) v = fe_sub(first ? S[t - 1][m1] : S[t - 1][m1 + 1], fe_mul(a.x_last, Sx[m1 + 1]));
        else v = first ? S[t - 1][7] : fe_zero();                // i = 7n is the last coefficient of the quotient
        if (a.sA) {
            if (m1 == 0) v = fe_add(v, fe_add(fe_add(a.sA[1 + m0], a.eC[1 + m0]), term));
            else if (m1 < 6 || (m1 == 6 && m0 < 2)) v = fe_add(v, term);
            else if (m1 == 6) v = fe_add(v, fe_add(a.sAp[m0 - 1], a.eCp[m0 - 1]));                  // i - p = m0 - 2
            else if (m0 < 2) v = fe_add(v, fe_add(a.sAp[m0 + n - 1], a.eCp[m0 + n - 1]));          // i - p = m0 + n - 2
        }
        a.cpoly[m0 + (size_t)m1 * n] = v;
    }
}
void k_combine_fused(dst_ctx* c, const fe* work, const fe* q4, size_t q_stride, fe* cpoly) {
    CombineArgs a{};
    a.work = work; a.cpoly = cpoly;
    if (q4) { a.sA = q4; a.sAp = q4 + q_stride; a.eC = q4 + 2 * q_stride; a.eCp = q4 + 3 * q_stride; }
    a.itw_lo = c->itw_lo; a.itw_hi = c->itw_hi; a.tw_lo = c->tw_lo; a.tw_hi = c->tw_hi;
    a.lo_bits = c->tw_lo_bits; a.log_n = c->log_n; a.log_N = c->log_N; a.log_b = c->log_b;
    a.eight_inv = c->eight_inv; a.x_last = c->x_last;
    const size_t blocks = (c->n + 1 + (COMBINE_THREADS - 2)) / (COMBINE_THREADS - 1);
    KScope ks_(c, "combine_fused_kernel", 16.0 * c->n * 16);
    hipLaunchKernelGGL(combine_fused_kernel, dim3((unsigned)blocks), dim3(COMBINE_THREADS), 0, c->stream, a);
}

// the two halves of k_intt8_cosets for the sharded prover: a rank inverts the size-n transforms of the evaluation cosets it owns BEFORE the
// exchange (1/world of them instead of all eight on every rank), the 8-point step across cosets follows on the gathered arrays
void k_intt_cosets_local(dst_ctx* c, fe* vals, fe* out, size_t cosets) { launch_two_pass(c, vals, 0, c->n, out, 0, c->n, cosets, 1, true, false); }
void k_cross8(dst_ctx* c, const fe* work, fe* out8n) {
    { KScope ks_(c, "cross8_kernel", 256.0 * c->n); hipLaunchKernelGGL(cross8_kernel, dim3((unsigned)((c->n + 255) / 256)), dim3(256), 0, c->stream, work, out8n,
                       c->itw_lo, c->itw_hi, c->tw_lo_bits, c->log_n, c->log_N, c->log_b, c->eight_inv); }
}
void k_intt8_cosets(dst_ctx* c, fe* vals, fe* out8n, fe* work) {
    launch_two_pass(c, vals, 0, c->n, work, 0, c->n, 8, 1, true, false);
    { KScope ks_(c, "cross8_kernel", 256.0 * c->n); hipLaunchKernelGGL(cross8_kernel, dim3((unsigned)((c->n + 255) / 256)), dim3(256), 0, c->stream, (const fe*)work, out8n,
                       c->itw_lo, c->itw_hi, c->tw_lo_bits, c->log_n, c->log_N, c->log_b, c->eight_inv); }
}

// ---- layout conversion (inspection; replicated FRI tail of the sharded path) ------------------------------------------------------------------------------------
__global__ void coset_to_natural_kernel(const fe* __restrict__ src, fe* __restrict__ dst, size_t n, size_t cosets) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * cosets) return;
    size_t k = i / cosets, j = i % cosets;
    dst[i] = src[j * n + k];
}
void k_coset_to_natural_len(dst_ctx* c, const fe* src, size_t cosets, size_t len, fe* dst) {      // [cosets][len] -> natural [len * cosets]
    size_t total = len * cosets;
    { KScope ks_(c, "coset_to_natural_kernel", 32.0 * total); hipLaunchKernelGGL(coset_to_natural_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, src, dst, len, cosets); }
}
void k_coset_to_natural(dst_ctx* c, const fe* src, size_t cosets, fe* dst) { k_coset_to_natural_len(c, src, cosets, c->n, dst); }
