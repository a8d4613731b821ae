// In-LDS butterfly stages of the LDS-family transforms (kernels_ntt.hip; also timed in isolation by tools/felab).
#pragma once
#include <type_traits>
#include "fe.h"

// compile-time loop: the bodies hold fully inlined 128-bit multiplications, far beyond the size the loop unroller accepts, so the
// unrolling is structural and every register-array index is a constant
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

// LDS slot of tile element (row i, column t).  NTT_SWIZZLE (laboratory switch, off: it did not pay in the full kernels) permutes the
// 16-byte slots of four-column tiles so that 64-byte-stride access patterns fall on distinct banks.
__device__ __forceinline__ uint32_t lds_slot(uint32_t i, uint32_t t, uint32_t log_t) {
#if defined(NTT_SWIZZLE)
    if (log_t == 2) return ((i ^ ((i >> 2) & 3u)) << 2) + (t ^ ((i >> 1) & 3u));
#endif
    return (i << log_t) + t;
}

// Column of the tile that lane w = (q, t) works on.  In the round whose butterflies span four ADJACENT rows (distance 2 and 1) the
// neighbouring butterflies q, q + 1, .. of a four-column tile start 256 bytes apart: with every lane of a butterfly on "its" column t the
// 16 lanes of an LDS access group hit the same banks four times over (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.22 / 0.35 in the two
// passes, profiles/r2).  Which lane takes which column is free: rotating the column by q spreads the group over the 256 bytes.
__device__ __forceinline__ uint32_t lds_column(uint32_t w, uint32_t q, uint32_t log_t, bool adjacent_rows) {
    const uint32_t T = 1u << log_t;
#if !defined(NTT_NO_ROTATE)
    if (adjacent_rows && log_t == 2) return (w + q) & (T - 1);
#endif
    return w & (T - 1);
}

// Rounds whose butterflies stay inside the rows of one wavefront run without workgroup barriers (-DNTT_NO_WAVE_LOCAL: a barrier after
// every round, for comparison; measured at 2^20 with the column rotation below: second passes 8.74 -> 8.58 -> 8.51 ms per proof, first
// passes unchanged -- neither bank conflicts nor barriers are what the transforms wait for)
__device__ __forceinline__ constexpr bool lds_wave_local() {
#if defined(NTT_NO_WAVE_LOCAL)
    return false;
#else
    return true;
#endif
}

// ordering point between two rounds of ONE wavefront: no instruction on the device (a wavefront's LDS accesses execute in order; the
// compiler must not move accesses across it), a workgroup barrier in the host-emulated test build (its work-items are fibers)
__device__ __forceinline__ void lds_wave_sync() {
#if defined(__HIP_DEVICE_COMPILE__)
#if !defined(__gfx950__)
#error "the rows a wavefront owns (own_rows in the rounds below) are counted for the 64-lane wavefronts of gfx950"
#endif
    // wave_barrier keeps the scheduler from moving instructions across this point but is no memory fence to the compiler: the empty
    // statement with a memory clobber is (no load is hoisted above it, no store sunk below it, nothing forwarded across it)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
#else
    __syncthreads();
#endif
}

// Slot of DIF stage twiddle w_len^e in LDS.  Stage s reads the entries e = pos * 2^(s-1) of neighbouring butterflies: from the third stage
// on that is a stride of a multiple of 256 bytes, i.e. every lane of an LDS access group on the same banks (measured on the rounds in
// isolation: 11 % of their time).  XOR-ing bits 3..5 and 6..8 of the index into its low three bits spreads strides 4, 16 and 64 over
// the banks and leaves unit stride alone.  The workgroup fills the table through the same map.
__device__ __forceinline__ uint32_t dif_tw_slot(uint32_t e) {
    return e ^ ((e >> 3) & 7u) ^ ((e >> 6) & 7u);
}

// The same transform for a tile shape known at compile time: the rounds are separate code with literal strides.  The lane index goes
// through an empty asm statement per call: the slot addresses of all rounds are invariant over the tiles a workgroup walks through, the
// compiler would compute them once, keep them live across the loop -- and spill them (measured: 1 GB of scratch reloads per launch).
__device__ __forceinline__ uint32_t lds_opaque_lane() {
    uint32_t lane = threadIdx.x;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(lane));
#endif
    return lane;
}
// Laboratory instrument (-DNTT_LAB_STAMPS, tools/r6_pass_stamps.py; never defined in a build that ships): a wavefront notes the shader
// clock (s_memtime) at the boundaries of a pass's phases -- load issue, barriers, LDS fill, every butterfly round, read-out -- into scalar
// registers and writes them out once per tile.  The product build compiles the macro to nothing.
#if defined(NTT_LAB_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
#define NTT_STAMP(arr, k) ((arr)[k] = __builtin_readcyclecounter())
#define NTT_STAMP_REAL(arr, k) ((arr)[k] = wall_clock64())          // s_memrealtime: constant 100 MHz -- what an s_memtime tick is worth is measured, not assumed
#else
#define NTT_STAMP(arr, k) ((void)0)
#define NTT_STAMP_REAL(arr, k) ((void)0)
#endif

// Consumer of the LAST round's results.  With the default the rounds leave the transformed tile in LDS.  A pass hands in an object
// whose `pre(row, t)` may start a global load for the element (row, t) before the butterfly's arithmetic (the four-step twiddle: its
// latency hides behind ~300 instructions) and whose `put(row, t, value, token)` finishes and stores the element: the last round then
// writes straight to HBM -- one LDS write + read per element, one barrier and the read-out's index arithmetic less per tile.
struct NttKeepInLds {
    static constexpr bool active = false;
    struct Tok {};
    __device__ __forceinline__ Tok pre(uint32_t, uint32_t) const { return Tok{}; }
    __device__ __forceinline__ void put(uint32_t, uint32_t, const fe&, const Tok&) const {}
};

// in-LDS DIF over the first index of L[len][T]; output position r holds frequency bitrev(r).  W: stage twiddles w_len^t in LDS.
// Two radix-2 stages are fused into one radix-4 round (one LDS round trip, one barrier and one index computation per two
// stages; the arithmetic is exactly the two radix-2 stages); an odd stage count ends with a plain radix-2 stage.
// lds_dif_round = stages s and s + 1, lds_dif_tail = the distance-1 stage of an odd stage count; every argument but the pointers is
// uniform, and with literal arguments (lds_ntt_dif_fixed) the index arithmetic folds into immediates.
template <int THREADS, class Out>
__device__ __forceinline__ void lds_dif_round(fe* L, const fe_tw* W, uint32_t log_len, uint32_t log_t, uint32_t s, const Out& out, uint32_t lane) {
    const uint32_t T = 1u << log_t;
    const uint32_t ld = log_len - s;             // log2 of the first stage's butterfly distance d
    const uint32_t d = 1u << ld, hd = d >> 1;
    const bool last = (s + 1 == log_len);        // second stage has distance 1: its twiddles are 1
    const bool fin = Out::active && last;        // the results of this round leave through `out`
    for (uint32_t w = lane; w < ((1u << log_len) >> 2) * T; w += THREADS) {
        const uint32_t q = w >> log_t;
        const uint32_t t = lds_column(w, q, log_t, hd == 1);
        const uint32_t pos = q & (hd - 1), blk = q >> (ld - 1);
        const uint32_t i0 = (blk << (ld + 1)) + pos;
        typename Out::Tok k0, k1, k2, k3;
        if (fin) { k0 = out.pre(i0, t); k1 = out.pre(i0 + hd, t); k2 = out.pre(i0 + d, t); k3 = out.pre(i0 + d + hd, t); }
        fe* p0 = L + lds_slot(i0, t, log_t); fe* p1 = L + lds_slot(i0 + hd, t, log_t); fe* p2 = L + lds_slot(i0 + d, t, log_t); fe* p3 = L + lds_slot(i0 + d + hd, t, log_t);
        const fe x0 = *p0, x1 = *p1, x2 = *p2, x3 = *p3;
        // stage s: (x0, x2) with w_2d^pos, (x1, x3) with w_2d^(pos + d/2)
        fe a0, a1, a2, a3;
        fe_addsub(x0, x2, a0, a2);
        fe_addsub(x1, x3, a1, a3);
#define NTT_TW_AT(idx) W[dif_tw_slot(idx)]
        if (hd != 1) a2 = fe_mul_tw(a2, NTT_TW_AT(pos << (s - 1)));          // hd == 1: pos == 0 in every lane
        a3 = fe_mul_tw(a3, NTT_TW_AT((pos + hd) << (s - 1)));
        // stage s + 1: (a0, a1) and (a2, a3) with w_d^pos
        fe y0, y1, y2, y3;
        fe_addsub(a0, a1, y0, y1);
        fe_addsub(a2, a3, y2, y3);
        if (!last && hd != 1) { const fe_tw tw = NTT_TW_AT(pos << s); y1 = fe_mul_tw(y1, tw); y3 = fe_mul_tw(y3, tw); }
#undef NTT_TW_AT
        if (fin) { out.put(i0, t, y0, k0); out.put(i0 + hd, t, y1, k1); out.put(i0 + d, t, y2, k2); out.put(i0 + d + hd, t, y3, k3); }
        else { *p0 = y0; *p1 = y1; *p2 = y2; *p3 = y3; }
    }
    // A wavefront's 64 lanes hold (64 / T) butterflies of 4 rows each.  Once a block of this round (2d rows) is no larger than that, the
    // rows a wavefront writes here are exactly the rows it reads in every later round: its own LDS accesses are ordered, no other
    // wavefront touches them, and the workgroup barrier between the rounds is not needed.
    const bool own_rows = lds_wave_local() && (2u << ld) <= ((64u >> log_t) << 2) && s + 3 <= log_len;
    if (!fin) { if (own_rows) lds_wave_sync(); else __syncthreads(); }
}
template <int THREADS, class Out>
__device__ __forceinline__ void lds_dif_tail(fe* L, uint32_t log_len, uint32_t log_t, const Out& out, uint32_t lane) {       // distance-1 stage, no twiddles
    const uint32_t T = 1u << log_t;
    for (uint32_t w = lane; w < ((1u << log_len) >> 1) * T; w += THREADS) {
        const uint32_t t = w & (T - 1), q = w >> log_t;
        typename Out::Tok k0, k1;
        if (Out::active) { k0 = out.pre(q << 1, t); k1 = out.pre((q << 1) + 1, t); }
        fe* p0 = L + lds_slot(q << 1, t, log_t); fe* p1 = L + lds_slot((q << 1) + 1, t, log_t);
        const fe a = *p0, b = *p1;
        fe sum, dif;
        fe_addsub(a, b, sum, dif);
        if (Out::active) { out.put(q << 1, t, sum, k0); out.put((q << 1) + 1, t, dif, k1); }
        else { *p0 = sum; *p1 = dif; }
    }
    if (!Out::active) __syncthreads();
}
template <int THREADS, class Out = NttKeepInLds>
__device__ __forceinline__ void lds_ntt_dif(fe* L, const fe_tw* W, uint32_t log_len, uint32_t log_t, uint32_t s_from, uint32_t s_to, const Out& out = Out()) {
    uint32_t s = s_from;                             // stages [s_from, s_to), s_from odd
    const uint32_t lane = lds_opaque_lane();         // see lds_ntt_dif_fixed: nothing derived from the lane index stays live across the tile loop
    for (; s + 1 <= log_len && s < s_to; s += 2) lds_dif_round<THREADS, Out>(L, W, log_len, log_t, s, out, lane);
    if (s == log_len && s < s_to) lds_dif_tail<THREADS, Out>(L, log_len, log_t, out, lane);
}
template <int THREADS, int LOG_LEN, int LOG_T, class Out = NttKeepInLds>
__device__ __forceinline__ void lds_ntt_dif_fixed(fe* L, const fe_tw* W, const Out& out = Out(), unsigned long long* stamps = nullptr) {
    const uint32_t lane = lds_opaque_lane();
    static_for<0, LOG_LEN / 2>([&](auto r_) { constexpr uint32_t s = 1u + 2u * decltype(r_)::value; lds_dif_round<THREADS, Out>(L, W, (uint32_t)LOG_LEN, (uint32_t)LOG_T, s, out, lane); (void)stamps; NTT_STAMP(stamps, decltype(r_)::value); });
    if constexpr (LOG_LEN & 1) lds_dif_tail<THREADS, Out>(L, (uint32_t)LOG_LEN, (uint32_t)LOG_T, out, lane);
}

// in-LDS DIT over the first index of L[len][T] for a COSET transform: X[k] = sum_m x[m] * g^m * w_len^(m*k).  The input sits in
// bit-reversed order (position brev(m) holds x[m]), the output is in natural order.  The sub-transforms over the even and
// odd indices are coset transforms with g^2, so the stage that merges blocks of size B multiplies by g^(len/B) * w_B^k: the
// pre-scale by g^m costs nothing -- it is part of twiddles that had to be applied anyway.  W holds them per stage at offset
// B/2 - 1 (len - 1 entries for the workgroup's coset).  Two stages per LDS round trip, as in lds_ntt_dif.
// `Wlast` != nullptr: the len/2 twiddles of the LAST stage (half of the coset's table) are not in LDS but read from this global array
// (contiguous per coset, L2-resident): tile + the other len/2 - 1 pairs then fit the 80 KiB that let two workgroups share a CU.
template <int THREADS, class Out>
__device__ __forceinline__ void lds_dit_round(fe* L, const fe_tw* W, uint32_t log_len, uint32_t log_t, uint32_t s, const fe_tw* __restrict__ Wlast, const Out& out, uint32_t lane) {
    const uint32_t T = 1u << log_t;
    const uint32_t B = 1u << s, half = B >> 1;                     // first stage merges blocks of size B / 2 into B, second B into 2B
    const bool fin = Out::active && s + 1 == log_len;              // the results of this round leave through `out`
    for (uint32_t w = lane; w < ((1u << log_len) >> 2) * T; w += THREADS) {
        const uint32_t q = w >> log_t;
        const uint32_t t = lds_column(w, q, log_t, half == 1);
        const uint32_t k = q & (half - 1), base = (q >> (s - 1)) << (s + 1);
        typename Out::Tok k0, k1, k2, k3;
        if (fin) { k0 = out.pre(base + k, t); k1 = out.pre(base + k + half, t); k2 = out.pre(base + k + B, t); k3 = out.pre(base + k + B + half, t); }
        fe* p0 = L + lds_slot(base + k, t, log_t); fe* p1 = L + lds_slot(base + k + half, t, log_t); fe* p2 = L + lds_slot(base + k + B, t, log_t); fe* p3 = L + lds_slot(base + k + B + half, t, log_t);
        const fe_tw tb = W[half - 1 + k];
        const fe x0 = *p0, x1 = fe_mul_tw(*p1, tb), x2 = *p2, x3 = fe_mul_tw(*p3, tb);
        fe a0, a1, b2, b3;
        fe_addsub(x0, x1, a0, a1);
        fe_addsub(x2, x3, b2, b3);
        const bool from_global = Wlast != nullptr && s + 1 == log_len;        // the second stage of this round is the last stage
        const fe_tw t2 = from_global ? Wlast[k] : W[B - 1 + k], t3 = from_global ? Wlast[k + half] : W[B - 1 + k + half];
        const fe a2 = fe_mul_tw(b2, t2), a3 = fe_mul_tw(b3, t3);
        fe y0, y1, y2, y3;
        fe_addsub(a0, a2, y0, y2);
        fe_addsub(a1, a3, y1, y3);
        if (fin) { out.put(base + k, t, y0, k0); out.put(base + k + B, t, y2, k2); out.put(base + k + half, t, y1, k1); out.put(base + k + B + half, t, y3, k3); }
        else { *p0 = y0; *p2 = y2; *p1 = y1; *p3 = y3; }
    }
    // the NEXT round merges blocks of 8B rows: while those fit the rows of one wavefront (see lds_dif_round) no barrier is needed
    const bool own_rows = lds_wave_local() && (8u << s) <= ((64u >> log_t) << 2) && s + 3 <= log_len;
    if (!fin) { if (own_rows) lds_wave_sync(); else __syncthreads(); }
}
template <int THREADS, class Out>
__device__ __forceinline__ void lds_dit_tail(fe* L, const fe_tw* W, uint32_t log_len, uint32_t log_t, const fe_tw* __restrict__ Wlast, const Out& out, uint32_t lane) {   // last single stage: blocks of len / 2 into len
    const uint32_t T = 1u << log_t;
    const uint32_t half = 1u << (log_len - 1);
    for (uint32_t w = lane; w < half * T; w += THREADS) {
        const uint32_t t = w & (T - 1), k = w >> log_t;
        typename Out::Tok k0, k1;
        if (Out::active) { k0 = out.pre(k, t); k1 = out.pre(k + half, t); }
        fe* p0 = L + lds_slot(k, t, log_t); fe* p1 = L + lds_slot(k + half, t, log_t);
        const fe u = *p0, v = fe_mul_tw(*p1, Wlast != nullptr ? Wlast[k] : W[half - 1 + k]);
        fe sum, dif;
        fe_addsub(u, v, sum, dif);
        if (Out::active) { out.put(k, t, sum, k0); out.put(k + half, t, dif, k1); }
        else { *p0 = sum; *p1 = dif; }
    }
    if (!Out::active) __syncthreads();
}
template <int THREADS, class Out = NttKeepInLds>
__device__ __forceinline__ void lds_ntt_dit(fe* L, const fe_tw* W, uint32_t log_len, uint32_t log_t, uint32_t s_from, uint32_t s_to, const fe_tw* __restrict__ Wlast = nullptr, const Out& out = Out()) {
    uint32_t s = s_from;                             // stages [s_from, s_to), s_from odd
    const uint32_t lane = lds_opaque_lane();
    for (; s + 1 <= log_len && s < s_to; s += 2) lds_dit_round<THREADS, Out>(L, W, log_len, log_t, s, Wlast, out, lane);
    if (s == log_len && s < s_to) lds_dit_tail<THREADS, Out>(L, W, log_len, log_t, Wlast, out, lane);
}
template <int THREADS, int LOG_LEN, int LOG_T, class Out = NttKeepInLds>
__device__ __forceinline__ void lds_ntt_dit_fixed(fe* L, const fe_tw* W, const fe_tw* __restrict__ Wlast = nullptr, const Out& out = Out(), unsigned long long* stamps = nullptr) {
    const uint32_t lane = lds_opaque_lane();
    static_for<0, LOG_LEN / 2>([&](auto r_) { constexpr uint32_t s = 1u + 2u * decltype(r_)::value; lds_dit_round<THREADS, Out>(L, W, (uint32_t)LOG_LEN, (uint32_t)LOG_T, s, Wlast, out, lane); (void)stamps; NTT_STAMP(stamps, decltype(r_)::value); });
    if constexpr (LOG_LEN & 1) lds_dit_tail<THREADS, Out>(L, W, (uint32_t)LOG_LEN, (uint32_t)LOG_T, Wlast, out, lane);
}
