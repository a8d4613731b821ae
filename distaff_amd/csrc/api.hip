// C-ABI of libdistaff_hip.so (include/distaff_hip.h): context, tables, the prover phases and proof assembly.
// The phase order and every Fiat-Shamir dependency follow stark::prove (/root/reference/src/stark/prover.rs:17-168).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include "ctx.h"
#include "host_proof.h"
#include "host_util.h"
#include "host_vm.h"

using namespace dsth;

static std::string g_create_error;

static double wall_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---- small host field helpers on limbs (fe.h compiles for the host too) ---------------------------------------------------------
static fe h_root_of_unity(uint32_t log_order) {               // field.rs:228: G^(2^(40 - log_order))
    const u128 G = (((u128)0x120532E7B364080Aull) << 64) | 0x86B8723E1920F4AAull;      // field.rs:14
    u128 r = G;
    for (uint32_t i = log_order; i < 40; i++) r = hf_mul(r, r);
    return fe_from_u128(r);
}
static std::vector<fe> h_powers(fe base, size_t count) {
    std::vector<fe> v(count);
    u128 b = fe_to_u128(base), cur = 1;
    for (size_t i = 0; i < count; i++) { v[i] = fe_from_u128(cur); cur = hf_mul(cur, b); }
    return v;
}
static std::vector<fe_tw> h_powers_tw(fe base, size_t count) {     // table pairs (w, w * 2^64 mod p) of the powers
    std::vector<fe_tw> v(count);
    u128 b = fe_to_u128(base), cur = 1;
    for (size_t i = 0; i < count; i++) { v[i] = fe_tw_make(fe_from_u128(cur)); cur = hf_mul(cur, b); }
    return v;
}
static fe h_inv(fe a) { return fe_from_u128(hf_pow(fe_to_u128(a), FIELD_P - 2)); }
static fe h_pow(fe a, u128 e) { return fe_from_u128(hf_pow(fe_to_u128(a), e)); }

template <class T>
static int dev_alloc(dst_ctx* c, T** p, size_t count) {
    HIP_TRY(c, hipMalloc((void**)p, count * sizeof(T) > 0 ? count * sizeof(T) : 16));
    return DST_OK;
}
template <class T>
static int dev_upload(dst_ctx* c, T** p, const std::vector<T>& v) {
    int r = dev_alloc(c, p, v.size());
    if (r) return r;
    HIP_TRY(c, hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return DST_OK;
}

// periodic constants of the AIR over a cycle of 16*8 steps: interpolate each 16-entry row, evaluate at w_128^s
// (constraints/utils.rs:87-113; decoder/mod.rs:95-100,219-223; stack/mod.rs:67-70)
static std::vector<fe> build_periodic_table() {
    const size_t cyc = 128;
    std::vector<fe> out(cyc * AIR_PERIODIC_STRIDE);
    u128 w16 = fe_to_u128(h_root_of_unity(4)), w128 = fe_to_u128(h_root_of_unity(7));
    u128 w16_inv = hf_pow(w16, 15), inv16 = hf_pow(16, FIELD_P - 2);
    static const uint8_t masks[3][16] = {{0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1}, {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0}, {0, 1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1}};
    for (int row = 0; row < 23; row++) {
        u128 vals[16], coef[16];
        for (int i = 0; i < 16; i++) vals[i] = row < 8 ? limbs(SPONGE_ARK[row][i]) : row < 20 ? limbs(HASHER_ARK[row - 8][i]) : (u128)masks[row - 20][i];
        for (int j = 0; j < 16; j++) {
            u128 acc = 0;
            for (int i = 0; i < 16; i++) acc = hf_add(acc, hf_mul(vals[i], hf_pow(w16_inv, (u128)((i * j) % 16))));
            coef[j] = hf_mul(acc, inv16);
        }
        for (size_t s = 0; s < cyc; s++) {
            u128 x = hf_pow(w128, s), acc = 0, pw = 1;
            for (int j = 0; j < 16; j++) { acc = hf_add(acc, hf_mul(coef[j], pw)); pw = hf_mul(pw, x); }
            out[s * AIR_PERIODIC_STRIDE + row] = fe_from_u128(acc);
        }
    }
    // columns 23..28: the cubes of the first hasher constants (rows 8..13) -- what (item + constant)^3 is for a stack item that is not
    // part of the trace (zero): the constraint kernel of a shallow stack reads them instead of cubing
    for (size_t s = 0; s < cyc; s++)
        for (int i = 0; i < 6; i++) { const u128 v = fe_to_u128(out[s * AIR_PERIODIC_STRIDE + 8 + i]); out[s * AIR_PERIODIC_STRIDE + 23 + i] = fe_from_u128(hf_mul(hf_mul(v, v), v)); }
    return out;
}

static void free_all(dst_ctx* c) {
    void* ptrs[] = {c->tw_lo, c->tw_hi, c->itw_lo, c->itw_hi, c->w1f, c->w2f, c->w1i, c->w2i, c->w1pf, c->w1pi, c->w2pf, c->w2pi, c->prescale, c->dit_last, c->tw4_lde, c->tw4_fwd, c->tw4_inv, c->tw4_row_fwd, c->tw4_row_inv, c->w3f, c->w3i, c->tmp2, c->periodic, c->trace == c->lde ? nullptr : c->trace, c->polys, c->lde, c->tmp,
                    c->trace_leaves, c->trace_nodes, c->air_consts, c->ceval, c->cwork, c->cpoly, c->cevals, c->cnodes, c->comp_poly, c->comp, c->scratch, c->d_u64, c->d_stage, c->d_fri_chain};
    for (void* p : ptrs) if (p) hipFree(p);
    for (auto& e : c->kpending) { hipEventDestroy(e.e0); hipEventDestroy(e.e1); }
    for (hipEvent_t e : c->event_pool) hipEventDestroy(e);
    if (c->gather_buf) hipFree(c->gather_buf);
    if (c->trace_upper) hipFree(c->trace_upper);
    if (c->c_upper) hipFree(c->c_upper);
    for (int d = 0; d < DST_MAX_FRI_LAYERS; d++) if (c->fri_upper[d]) hipFree(c->fri_upper[d]);
    if (c->fri_nat0) hipFree(c->fri_nat0);
    for (int d = 0; d < DST_MAX_FRI_LAYERS; d++) {
        if (d > 0 && c->fri_e[d]) hipFree(c->fri_e[d]);
        if (c->fri_leaves[d]) hipFree(c->fri_leaves[d]);
        if (c->fri_nodes[d]) hipFree(c->fri_nodes[d]);
    }
    for (hipEvent_t e : c->upload_done) hipEventDestroy(e);
    if (c->upload_stream) hipStreamDestroy(c->upload_stream);
    for (hipEvent_t e : c->comm_events) hipEventDestroy(e);
    if (c->comm_stream) hipStreamDestroy(c->comm_stream);
    if (c->d_status) hipFree(c->d_status);
    if (c->h_stage) hipHostFree(c->h_stage);
    for (hipEvent_t e : c->ph_ev) if (e) hipEventDestroy(e);
    for (hipEvent_t e : c->sh_ev) if (e) hipEventDestroy(e);
    for (auto& ce : c->coll_ev) { hipEventDestroy(ce.e0); hipEventDestroy(ce.e1); }
    if (c->stream) hipStreamDestroy(c->stream);
}

static int ctx_init(dst_ctx* c) {
    c->read_switches();                                  // the DISTAFF_* variables as they are NOW; nothing reads the environment after this
    const dst_params& p = c->prm;
    if (p.log_trace_length < 4 || p.log_trace_length > 24) { c->err = "log_trace_length must be in [4, 24]"; return DST_ERR_ARG; }      // lib.rs:82 MIN_TRACE_LENGTH = 16
    if (p.log_blowup < 4 || p.log_blowup > 8) { c->err = "extension factor must be in [16, 256]"; return DST_ERR_ARG; }
    if (p.ctx_depth > 16 || p.loop_depth > 8) { c->err = "context / loop depth out of range"; return DST_ERR_ARG; }
    if (p.width >= 128 || p.width <= 15 + p.ctx_depth + p.loop_depth) { c->err = "register count out of range"; return DST_ERR_ARG; }
    if (p.num_queries == 0 || p.num_queries > 128 || p.grinding_factor > 32) { c->err = "invalid proof options"; return DST_ERR_ARG; }
    if (p.world == 0 || p.rank >= p.world || (p.world & (p.world - 1)) || p.world > 8 || p.world > (1u << p.log_blowup) / 2) {
        // a rank needs >= 1 of the 8 evaluation cosets and >= 2 LDE cosets (one constraint-tree leaf = a pair of neighbouring cosets)
        c->err = "invalid rank / world: world must be a power of two <= min(8, blowup / 2)"; return DST_ERR_ARG;
    }
    if (p.log_trace_length + p.log_blowup > 40) { c->err = "LDE domain exceeds 2^40"; return DST_ERR_ARG; }
    c->log_n = p.log_trace_length; c->log_b = p.log_blowup; c->log_N = c->log_n + c->log_b;
    c->n = (size_t)1 << c->log_n; c->B = (size_t)1 << c->log_b; c->N = c->n * c->B; c->W = p.width;
    c->Bc = c->B / p.world; c->j0 = c->Bc * p.rank;
    c->stack_depth = p.width - 15 - p.ctx_depth - p.loop_depth;
    if (c->stack_depth > 32) { c->err = "user stack deeper than 32 registers"; return DST_ERR_ARG; }
    c->device = p.device;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamCreate(&c->stream));
    HIP_TRY(c, hipHostMalloc((void**)&c->h_stage, HS_TOTAL, hipHostMallocDefault));
    for (hipEvent_t& e : c->ph_ev) HIP_TRY(c, hipEventCreate(&e));

    // NTT plan: n = n1 * n2 in two HBM passes, tiles bounded by 64 KiB of LDS; from n = 2^21 (measured cross-over) three passes n = n1 * nm * n3 with
    // 16-column tiles (256-byte HBM segments) instead of 4096-point tiles that hold one or two columns
    NttPlan& pl = c->plan;
    const char* force = c->sw("DISTAFF_NTT");
    // n = 2^21, 2^22: still two passes -- their 2048-point factors run as a register pre-stage + 1024-point LDS tiles (NttArgs::pre);
    // DISTAFF_NTT=pre forces the pre-stages onto smaller transforms (tests), DISTAFF_NTT=3pass keeps the three-pass plan from 2^21 on
    const bool force_3 = force && !strcmp(force, "3pass"), force_pre = force && !strcmp(force, "pre") && c->log_n >= 10 && c->log_n <= 22;
    const bool pre_plan = force_pre || ((c->log_n == 21 || c->log_n == 22) && !force);
    const bool three = !pre_plan && ((c->log_n >= 21 && !(force && (!strcmp(force, "reg") || !strcmp(force, "lds")))) || (force_3 && c->log_n >= 12));
    pl.log_n = c->log_n;
    if (three) {
        // shape n1 * nm * n3 with n1 >= nm >= n3 as balanced as possible, at most 2^8 each; DISTAFF_NTT_SHAPE=a,b overrides n1, nm (tests)
        uint32_t a = (c->log_n + 2) / 3, b = (c->log_n - a + 1) / 2;
        if (a > 8) { a = 8; b = 8; }
        if (const char* sh = c->sw("DISTAFF_NTT_SHAPE")) { unsigned x = 0, y = 0; if (sscanf(sh, "%u,%u", &x, &y) == 2 && x >= 4 && y >= 4 && x <= 8 && y <= 8 && x + y + 4 <= c->log_n && c->log_n - x - y <= 8) { a = x; b = y; } }
        pl.log_n1 = a; pl.log_n2 = c->log_n - a; pl.log_n3 = c->log_n - a - b;
    }
    else { pl.log_n1 = (c->log_n + 1) / 2; pl.log_n2 = c->log_n / 2; pl.log_n3 = 0; }
    if (pre_plan) { pl.pre_a = (force_pre || pl.log_n1 > 10) ? 1u : 0u; pl.pre_b = (force_pre || pl.log_n2 > 10) ? 1u : 0u; }
    auto tile_for = [](uint32_t log_len, uint32_t other_len_log, uint32_t cap) {
        uint32_t t = cap;
        while (t > 1 && (((size_t)1 << log_len) * t * sizeof(fe) > 65536 || t > (1u << other_len_log))) t >>= 1;
        return t;
    };
    pl.tile_a = tile_for(pl.log_n1 - pl.pre_a, pl.log_n2, three ? 16 : 4);
    pl.tile_b = three ? tile_for(pl.log_n3, pl.log_n1, 16) : tile_for(pl.log_n2 - pl.pre_b, pl.log_n1, 4);
    pl.tile_m = three ? tile_for(pl.log_n2 - pl.log_n3, pl.log_n3, 16) : 1;
    // kernel choice per pass (measured): the LDS radix-2 kernels win while a tile holds >= 2 columns in 64 KiB of LDS; the
    // register-radix kernels take over for 4096-point tiles.  DISTAFF_NTT=reg|lds forces one two-pass family (tests run both).
    {
        const bool reg_ok = !three && !pre_plan && pl.log_n2 >= 6 && pl.log_n1 <= 12;
        pl.reg_a = reg_ok && pl.log_n1 >= 12; pl.reg_b = reg_ok && pl.log_n2 >= 12;      // 4096-point tiles: the LDS family is down to one column (16-byte segments)
        if (force && !strcmp(force, "reg") && reg_ok) pl.reg_a = pl.reg_b = true;
        if (force && !strcmp(force, "lds")) { pl.reg_a = pl.reg_a && pl.log_n1 >= 12; pl.reg_b = pl.reg_b && pl.log_n2 >= 12; }   // a 4096-point coset DIT (64 KiB tile + 128 KiB of twiddle pairs) does not fit LDS
    }
    {
        fe w16 = h_root_of_unity(4), w16i = h_inv(w16);
        std::vector<fe> f = h_powers(w16, 8), b = h_powers(w16i, 8);
        for (int j = 0; j < 8; j++) { c->c16f[j] = f[j]; c->c16i[j] = b[j]; }
    }

    // twiddle tables
    fe wN = h_root_of_unity(c->log_N), wN_inv = h_inv(wN);
    c->tw_lo_bits = (c->log_N + 1) / 2;
    uint32_t hi_bits = c->log_N - c->tw_lo_bits;
    int r;
    if ((r = dev_upload(c, &c->tw_lo, h_powers(wN, (size_t)1 << c->tw_lo_bits)))) return r;
    if ((r = dev_upload(c, &c->tw_hi, h_powers(h_pow(wN, (u128)1 << c->tw_lo_bits), (size_t)1 << hi_bits)))) return r;
    if ((r = dev_upload(c, &c->itw_lo, h_powers(wN_inv, (size_t)1 << c->tw_lo_bits)))) return r;
    if ((r = dev_upload(c, &c->itw_hi, h_powers(h_pow(wN_inv, (u128)1 << c->tw_lo_bits), (size_t)1 << hi_bits)))) return r;
    const uint32_t log_second = pl.log_n3 ? pl.log_n2 - pl.log_n3 : pl.log_n2 - pl.pre_b;        // three-pass: w2* serve the middle pass
    const uint32_t log_first = pl.log_n1 - pl.pre_a;                                             // length of the first pass's LDS transform
    fe w1 = h_root_of_unity(log_first), w2 = h_root_of_unity(log_second);
    if ((r = dev_upload(c, &c->w1f, h_powers_tw(w1, (size_t)1 << (log_first - 1))))) return r;
    if ((r = dev_upload(c, &c->w2f, h_powers_tw(w2, (size_t)1 << (log_second - 1))))) return r;
    if ((r = dev_upload(c, &c->w1i, h_powers_tw(h_inv(w1), (size_t)1 << (log_first - 1))))) return r;
    if ((r = dev_upload(c, &c->w2i, h_powers_tw(h_inv(w2), (size_t)1 << (log_second - 1))))) return r;
    if (pl.pre_a) {
        fe wp = h_root_of_unity(pl.log_n1);
        if ((r = dev_upload(c, &c->w1pf, h_powers_tw(wp, (size_t)1 << (pl.log_n1 - 1))))) return r;
        if ((r = dev_upload(c, &c->w1pi, h_powers_tw(h_inv(wp), (size_t)1 << (pl.log_n1 - 1))))) return r;
    }
    if (pl.pre_b) {
        fe wp = h_root_of_unity(pl.log_n2);
        if ((r = dev_upload(c, &c->w2pf, h_powers_tw(wp, (size_t)1 << (pl.log_n2 - 1))))) return r;
        if ((r = dev_upload(c, &c->w2pi, h_powers_tw(h_inv(wp), (size_t)1 << (pl.log_n2 - 1))))) return r;
    }
    if (pl.log_n3) {
        fe w3 = h_root_of_unity(pl.log_n3);
        if ((r = dev_upload(c, &c->w3f, h_powers_tw(w3, (size_t)1 << (pl.log_n3 - 1))))) return r;
        if ((r = dev_upload(c, &c->w3i, h_powers_tw(h_inv(w3), (size_t)1 << (pl.log_n3 - 1))))) return r;
        if ((r = dev_alloc(c, &c->tw4_row_fwd, (size_t)1 << pl.log_n2))) return r;
        if ((r = dev_alloc(c, &c->tw4_row_inv, (size_t)1 << pl.log_n2))) return r;
    }
    {
        const std::vector<fe_tw> pre = h_powers_tw(h_root_of_unity(c->log_b + pl.log_n1), (size_t)1 << (c->log_b + pl.log_n1));
        if ((r = dev_upload(c, &c->prescale, pre))) return r;
        // last-stage twiddles of the (half-length, with a pre-stage) coset DITs: [B][R][len / 2] entries pre[j + B * (h + R * k)]
        const size_t R = (size_t)1 << pl.pre_a, half = (size_t)1 << (pl.log_n1 - pl.pre_a - 1);
        std::vector<fe_tw> last(c->B * R * half);
        for (size_t j = 0; j < c->B; j++) for (size_t h = 0; h < R; h++) for (size_t k = 0; k < half; k++) last[(j * R + h) * half + k] = pre[j + c->B * (h + R * k)];
        if ((r = dev_upload(c, &c->dit_last, last))) return r;
    }
    if ((r = dev_alloc(c, &c->tw4_lde, c->Bc * c->n))) return r;
    if ((r = dev_alloc(c, &c->tw4_fwd, c->n))) return r;
    if ((r = dev_alloc(c, &c->tw4_inv, c->n))) return r;
    if ((r = dev_upload(c, &c->periodic, build_periodic_table()))) return r;
    c->n_inv = fe_from_u128(hf_pow((u128)c->n, FIELD_P - 2));
    c->n_inv_tw = fe_tw_make(c->n_inv);
    c->eight_inv = fe_from_u128(hf_pow(8, FIELD_P - 2));
    c->four_inv = fe_from_u128(hf_pow(4, FIELD_P - 2));
    c->iota = h_root_of_unity(2);
    c->g_trace = h_root_of_unity(c->log_n);
    c->x_last = h_inv(c->g_trace);

    // data buffers
    const size_t n = c->n, Nl = c->Bc * n;
    // sharded contexts: the coefficient vectors are all-gathered in rounds of `world` registers (dst_prove_sharded), so the array holds
    // a whole number of rounds
    if ((r = dev_alloc(c, &c->polys, (c->W + c->prm.world - 1) / c->prm.world * c->prm.world * n))) return r;
    if ((r = dev_alloc(c, &c->lde, c->W * Nl))) return r;
    if (c->j0 == 0 && c->Bc > 1 && !c->sw_is("DISTAFF_TRACE_BUFFER", "1")) { c->trace = c->lde; c->trace_stride = Nl; }     // see ctx.h; DISTAFF_TRACE_BUFFER=1: separate buffer + copy (tests)
    else { if ((r = dev_alloc(c, &c->trace, c->W * n))) return r; c->trace_stride = n; }
    // staging buffer of the two-pass transforms: tmp_regs registers x Bc cosets per pair of launches.  Every launch ends with a partly
    // filled last wave of workgroups (4 registers x 31 cosets at n = 2^20: 7.75 waves of 512 resident workgroups), so as many registers per
    // launch as 12 GiB of staging hold (all 20 at n = 2^20: 39.75 waves, one tail instead of five); at least 4.  DISTAFF_TMP_REGS overrides.
    // The cap follows the free memory of the device at creation (several contexts may share one GPU: thread-ranks, tests): at most 12 GiB
    // and at most a quarter of what is free after the buffers above.
    size_t stage_cap = (size_t)12 << 30;
    { size_t free_b = 0, total_b = 0; if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b / 4 < stage_cap) stage_cap = free_b / 4; }
    c->tmp_regs = 4;
    while (c->tmp_regs < c->W && (c->tmp_regs + 1) * (pl.log_n3 ? 2 : 1) * c->Bc * n * sizeof(fe) <= stage_cap) c->tmp_regs++;
    if (const char* e = c->sw("DISTAFF_TMP_REGS")) { const long k = atol(e); if (k >= 4 && k <= (long)c->W) c->tmp_regs = (size_t)k; }
    if ((r = dev_alloc(c, &c->tmp, c->Bc * c->tmp_regs * n))) return r;
    if (pl.log_n3 && (r = dev_alloc(c, &c->tmp2, c->Bc * c->tmp_regs * n))) return r;
    if ((r = dev_alloc(c, &c->trace_leaves, Nl))) return r;
    if ((r = dev_alloc(c, &c->trace_nodes, Nl))) return r;
    if ((r = dev_alloc(c, &c->ceval, 3 * 8 * n))) return r;
    if ((r = dev_alloc(c, &c->cwork, 8 * 8 * n))) return r;
    if ((r = dev_alloc(c, &c->cpoly, 8 * n))) return r;
    if ((r = dev_alloc(c, &c->cevals, Nl))) return r;
    if ((r = dev_alloc(c, &c->cnodes, Nl / 2))) return r;
    if ((r = dev_alloc(c, &c->comp_poly, 8 * n))) return r;
    if ((r = dev_alloc(c, &c->comp, Nl))) return r;
    c->scratch_elems = (size_t)1 << 21;
    if ((r = dev_alloc(c, &c->scratch, c->scratch_elems))) return r;
    if ((r = dev_alloc(c, &c->d_u64, 64))) return r;
    if ((r = dev_alloc(c, &c->d_fri_chain, DST_MAX_FRI_LAYERS * 48))) return r;
    c->stage_bytes = (size_t)8 << 20;
    if ((r = dev_alloc(c, &c->d_stage, c->stage_bytes))) return r;
    // FRI layers: sizes N, N/4, ... while > 256; the last one (<= 256) is the remainder (fri/prover.rs:21, fri/mod.rs:13)
    size_t sz = c->N;
    int d = 0;
    for (;; d++) {
        if (d >= DST_MAX_FRI_LAYERS) { c->err = "too many FRI layers"; return DST_ERR_ARG; }
        c->fri_size[d] = sz;
        if (d == 0) c->fri_e[0] = c->comp;
        else if ((r = dev_alloc(c, &c->fri_e[d], sz))) return r;
        if ((r = dev_alloc(c, &c->fri_leaves[d], sz / 4))) return r;
        if ((r = dev_alloc(c, &c->fri_nodes[d], sz / 4))) return r;
        if (sz <= 256) break;
        sz /= 4;
    }
    c->num_fri_layers = d + 1;
    return k_build_twiddle_tables(c);
}

static const fe* as_fe(const uint8_t* p) { return reinterpret_cast<const fe*>(p); }
static std::vector<fe> copy_fe(const uint8_t* p, size_t count) { std::vector<fe> v(count); memcpy(v.data(), p, count * 16); return v; }

extern "C" int dst_internal_build_proof(dst_ctx* c, const uint64_t* positions, uint32_t num_positions, uint64_t pow_nonce, std::vector<uint8_t>& proof);   // shard.hip
extern "C" int dst_internal_shard_buffers(dst_ctx* c);      // shard.hip: exchange buffers of a sharded context

extern "C" {

int dst_ctx_create(const dst_params* params, dst_ctx** out) {
    if (!params || !out) { g_create_error = "null argument"; return DST_ERR_ARG; }
    dst_ctx* c = new dst_ctx();
    c->prm = *params;
    int r = ctx_init(c);
    if (r == DST_OK && c->prm.world > 1) r = dst_internal_shard_buffers(c);       // a rank can join every collective from its first proof on
    if (r != DST_OK) { g_create_error = c->err; free_all(c); delete c; *out = nullptr; return r; }
    *out = c;
    return DST_OK;
}
void dst_ctx_destroy(dst_ctx* c) { if (!c) return; hipSetDevice(c->device); free_all(c); delete c; }
const char* dst_last_error(const dst_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }
int dst_phase_ms(const dst_ctx* c, double out_ms[9]) { if (!c || !out_ms) return DST_ERR_ARG; for (int i = 0; i < 9; i++) out_ms[i] = c->phase_ms[i]; return DST_OK; }

// an asynchronous upload still in flight writes the same buffer from upload_stream: drain it before another upload starts
static int drain_pending_upload(dst_ctx* c) {
    if (c->upload_pending && c->upload_stream) HIP_TRY(c, hipStreamSynchronize(c->upload_stream));
    c->upload_pending = false;
    return DST_OK;
}

int dst_trace_upload(dst_ctx* c, const uint8_t* const* cols) {
    if (!c || !cols) return DST_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    { int rd = drain_pending_upload(c); if (rd) return rd; }
    for (size_t i = 0; i < c->W; i++) HIP_TRY(c, hipMemcpyAsync(c->trace + i * c->trace_stride, cols[i], c->n * 16, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->upload_pending = false; c->trace_owned_only = false;
    c->have_trace = true; c->committed = c->constraints_done = c->composed = false;
    return DST_OK;
}
// Sharded contexts: uploads only the registers this rank interpolates, r = rank (mod world) -- 1/world of the trace per GPU instead of
// all of it.  cols[r] of the other registers is not read (may be NULL).  Only dst_prove_sharded accepts a context in this state.
int dst_trace_upload_owned(dst_ctx* c, const uint8_t* const* cols) {
    if (!c || !cols) return DST_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    { int rd = drain_pending_upload(c); if (rd) return rd; }
    for (size_t i = c->prm.rank; i < c->W; i += c->prm.world) {
        if (!cols[i]) { c->err = "dst_trace_upload_owned: register " + std::to_string(i) + " belongs to this rank but has no data"; return DST_ERR_ARG; }
        HIP_TRY(c, hipMemcpyAsync(c->trace + i * c->trace_stride, cols[i], c->n * 16, hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->upload_pending = false; c->trace_owned_only = c->prm.world > 1;
    c->have_trace = true; c->committed = c->constraints_done = c->composed = false;
    return DST_OK;
}
int dst_trace_upload_contiguous(dst_ctx* c, const uint8_t* cols) {
    if (!c || !cols) return DST_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    { int rd = drain_pending_upload(c); if (rd) return rd; }
    for (size_t i = 0; i < c->W; i++) HIP_TRY(c, hipMemcpy(c->trace + i * c->trace_stride, cols + i * c->n * 16, c->n * 16, hipMemcpyHostToDevice));
    c->upload_pending = false; c->trace_owned_only = false;
    c->have_trace = true; c->committed = c->constraints_done = c->composed = false;
    return DST_OK;
}

// Pinned host memory for the trace: the reference hands stark::prove a host-resident TraceTable (prover.rs:17, trace_table.rs:10); from
// pinned memory the upload runs as asynchronous DMA and overlaps with the extension of the registers that have already arrived.
int dst_pinned_alloc(size_t bytes, void** out) {
    if (!out || bytes == 0) return DST_ERR_ARG;
    return hipHostMalloc(out, bytes, hipHostMallocDefault) == hipSuccess ? DST_OK : DST_ERR_HIP;
}
int dst_pinned_free(void* p) { return (!p || hipHostFree(p) == hipSuccess) ? DST_OK : DST_ERR_HIP; }

// Starts the upload of the W register columns (host pointers, ideally from dst_pinned_alloc) on a copy stream and returns at once.  The
// next dst_commit_trace / dst_prove interpolates and extends the registers group by group as their copies complete, so only the first
// group's transfer is exposed.  The host buffers must stay valid until that call returns.
int dst_trace_upload_async(dst_ctx* c, const uint8_t* const* cols) {
    if (!c || !cols) return DST_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->upload_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->upload_stream, hipStreamNonBlocking));
    { int rd = drain_pending_upload(c); if (rd) return rd; }
    // the previous proof's kernels may still read the buffer on c->stream (coset 0 of the extension IS the trace buffer)
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    // groups of 4 registers (one extension launch, see k_lde_columns) -- except that the first group is ONE register: its transfer is the
    // only one nothing overlaps with
    c->upload_bounds.clear();
    for (size_t first = 0; first < c->W;) { c->upload_bounds.push_back(first); first += (first == 0 && c->W > 4) ? (c->W % 4 ? c->W % 4 : 1) : 4; }
    c->upload_bounds.push_back(c->W);
    const size_t groups = c->upload_bounds.size() - 1;
    while (c->upload_done.size() < groups) { hipEvent_t e; HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming)); c->upload_done.push_back(e); }
    for (size_t g = 0; g < groups; g++) {
        for (size_t i = c->upload_bounds[g]; i < c->upload_bounds[g + 1]; i++)
            HIP_TRY(c, hipMemcpyAsync(c->trace + i * c->trace_stride, cols[i], c->n * 16, hipMemcpyHostToDevice, c->upload_stream));
        HIP_TRY(c, hipEventRecord(c->upload_done[g], c->upload_stream));
    }
    c->upload_pending = true; c->trace_owned_only = false;
    c->have_trace = true; c->committed = c->constraints_done = c->composed = false;
    return DST_OK;
}

// ---- steps 1-2 ------------------------------------------------------------------------------------------------------------------
int dst_commit_trace(dst_ctx* c, uint8_t trace_root[32]) {
    if (!c || !trace_root) return DST_ERR_ARG;
    if (!c->have_trace) { c->err = "dst_commit_trace: no trace uploaded"; return DST_ERR_STATE; }
    if (c->trace_owned_only) { c->err = "dst_commit_trace: only this rank's registers were uploaded (dst_trace_upload_owned): use dst_prove_sharded"; return DST_ERR_STATE; }
    HIP_TRY(c, hipSetDevice(c->device));
    c->sharded_layout = false;
    double t0 = wall_ms();
    HIP_TRY(c, hipEventRecord(c->ph_ev[4], c->stream));
    if (c->upload_pending) {
        // registers arrive in groups (dst_trace_upload_async): each group is interpolated and extended as soon as its copy has landed
        for (size_t g = 0; g + 1 < c->upload_bounds.size(); g++) {
            const size_t first = c->upload_bounds[g], cnt = c->upload_bounds[g + 1] - first;
            HIP_TRY(c, hipStreamWaitEvent(c->stream, c->upload_done[g], 0));
            k_intt_columns(c, c->trace + first * c->trace_stride, c->trace_stride, c->polys + first * c->n, cnt);
            k_lde_columns(c, c->polys + first * c->n, c->lde + first * c->Bc * c->n, cnt);
        }
        c->upload_pending = false;
    } else {
        k_intt_columns(c, c->trace, c->trace_stride, c->polys, c->W);   // interpolate_fft_twiddles (trace_table.rs:159)
        k_lde_columns(c, c->polys, c->lde, c->W);                    // eval_fft_twiddles over the LDE domain (trace_table.rs:166)
    }
    HIP_TRY(c, hipEventRecord(c->ph_ev[5], c->stream));         // end of the extension: the host does not wait here
    k_trace_leaves(c);                                           // trace_table.rs:174-185
    k_merkle_levels(c, c->trace_leaves, c->trace_nodes, c->Bc * c->n);
    HIP_TRY(c, hipMemcpyAsync(c->trace_root, c->trace_nodes + 1, 32, hipMemcpyDeviceToHost, c->stream));
    // last state of the un-extended trace: op counter and program hash (evaluator.rs:37,73-74)
    fe last[3];
    for (int i = 0; i < 3; i++) HIP_TRY(c, hipMemcpyAsync(&last[i], c->trace + (size_t)i * c->trace_stride + (c->n - 1), 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    c->op_count = (uint64_t)fe_to_u128(last[0]);
    c->program_hash[0] = last[1]; c->program_hash[1] = last[2];
    memcpy(trace_root, c->trace_root, 32);
    {
        // extension = what the stream spent up to the event; the rest of the call's wall time is the tree (and the host's share)
        float lde_ms = 0;
        const double total = wall_ms() - t0;
        if (hipEventElapsedTime(&lde_ms, c->ph_ev[4], c->ph_ev[5]) != hipSuccess || lde_ms > total) lde_ms = 0;
        c->phase_ms[0] = lde_ms; c->phase_ms[1] = total - lde_ms;
    }
    c->committed = true; c->constraints_done = c->composed = false;
    return DST_OK;
}

// ---- steps 3-5 ------------------------------------------------------------------------------------------------------------------
// constraint degrees in constraint-index order (decoder/mod.rs:31-47, stack/mod.rs:40-41) and the coefficient each
// constraint receives when they are visited in degree-group order (evaluator.rs:335-358,385-406; coefficients.rs:140-185)
// the 344 constraint coefficients and the compacted transition coefficients -> device, queued from the page-locked staging area (the
// callers' vectors go out of scope while the copies may still be pending: the evaluation's verdict is not waited for)
int dst_internal_upload_draws(dst_ctx* c, const fe* draws344, const std::vector<fe>& tc, fe* d_coef, fe* d_tc) {
    if ((344 + tc.size()) * sizeof(fe) > HS_DRAWS_BYTES) { c->err = "too many transition coefficients for the staging area"; return DST_ERR_ARG; }
    fe* h = reinterpret_cast<fe*>(c->h_stage + HS_DRAWS);
    memcpy(h, draws344, 344 * sizeof(fe));
    memcpy(h + 344, tc.data(), tc.size() * sizeof(fe));
    HIP_TRY(c, hipMemcpyAsync(d_coef, h, 344 * sizeof(fe), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(d_tc, h + 344, tc.size() * sizeof(fe), hipMemcpyHostToDevice, c->stream));
    return DST_OK;
}
void dst_internal_transition_coefficients(const dst_ctx* c, const fe* draws344, std::vector<fe>& tc) {
    const size_t cl = c->prm.ctx_depth > 1 ? c->prm.ctx_depth : 1, ll = c->prm.loop_depth > 1 ? c->prm.loop_depth : 1;
    const size_t sl = c->stack_depth > 8 ? c->stack_depth : 8;
    std::vector<int> deg = {2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 8, 8, 6, 4, 6, 7, 6, 6, 4};
    deg.resize(20 + cl + ll, 4);
    deg.resize(20 + cl + ll + 2 + c->stack_depth, 7);
    // compacted coefficient list (build_transition_coefficients)
    const fe* t = draws344 + 188;
    std::vector<fe> cc;
    auto take = [&](size_t from, size_t cnt) { for (size_t i = 0; i < cnt; i++) cc.push_back(t[from + i]); };
    take(0, 40); take(40, 2 * cl); take(72, 2 * ll); take(88, 4); take(92, 2 * sl);
    const size_t nc = deg.size();
    tc.assign(2 * nc, fe_zero());
    size_t i = 0;
    for (int d = 0; d <= 8; d++)
        for (size_t k = 0; k < nc; k++)
            if (deg[k] == d) { tc[k] = cc[2 * i]; tc[nc + k] = cc[2 * i + 1]; i++; }
}

// The two boundary combinations (evaluator.rs:181-326) in COEFFICIENT form.  With v_k(x) = T_k(x) - const_k for the constrained
// registers k, the reference evaluates I(x) = sum_k v_k(x) * (cc_k + cc'_k * x^p), p = 6n + 2, on the 8n-point domain and interpolates
// it again (constraint_table.rs:54-62).  I has degree < 7n + 2 < 8n, so the interpolant is I itself, and its coefficients are two
// linear combinations of the trace polynomials: A = sum_k cc_k v_k at [0, n) and A' = sum_k cc'_k v_k at [p, p + n).  Nothing is
// evaluated and nothing is interpolated: ip / fp (8n coefficients each, before the divisions) are written directly.
// DISTAFF_BOUNDARY=eval keeps the evaluate-and-interpolate route (the tests compare its evaluation vectors with the oracle's).
bool dst_internal_boundary_by_evaluation(const dst_ctx* c) { return DST_TEST_HOOKS && c->sw_is("DISTAFF_BOUNDARY", "eval"); }
int dst_internal_boundary_polys(dst_ctx* c, const fe* draws344, fe* ip, fe* fp, fe* o0, fe* o1, fe* o2, fe* o3) {
    const size_t n = c->n, D = 8 * n, W = c->W, p = 6 * n + 2;
    const uint32_t ctx_depth = c->prm.ctx_depth, loop_depth = c->prm.loop_depth;
    const size_t sd = c->stack_depth;
    const size_t cl = ctx_depth > 1 ? ctx_depth : 1, ll = loop_depth > 1 ? loop_depth : 1;
    // host: weights per trace register (plain, degree-adjusted) and the constant terms, for the first-step and last-step combinations
    std::vector<u128> w(4 * W, 0);                       // [pass][adj][register]
    u128 g[4] = {0, 0, 0, 0};                            // [pass][adj]
    const u128 one = 1;
    for (int pass = 0; pass < 2; pass++) {
        const fe* cc = draws344 + pass * 94;
        auto term = [&](int col, u128 constant, size_t idx) {       // value = T_col(x) - constant (col < 0: no register, value = -constant)
            for (int adj = 0; adj < 2; adj++) {
                const u128 k = fe_to_u128(cc[idx + adj]);
                if (col >= 0) w[(pass * 2 + adj) * W + col] = hf_add(w[(pass * 2 + adj) * W + col], k);
                if (constant != 0) g[pass * 2 + adj] = hf_add(g[pass * 2 + adj], hf_mul(k, constant));
            }
        };
        term(0, pass ? (u128)c->op_count : 0, 0);
        if (pass == 0) { for (int i = 0; i < 4; i++) term(1 + i, 0, 2 + 2 * i); }
        else { for (int i = 0; i < 2; i++) term(1 + i, fe_to_u128(c->program_hash[i]), 2 + 2 * i); }
        for (int i = 0; i < 3; i++) term(5 + i, pass ? one : 0, 10 + 2 * i);
        for (int i = 0; i < 5; i++) term(8 + i, pass ? one : 0, 16 + 2 * i);
        for (int i = 0; i < 2; i++) term(13 + i, pass ? one : 0, 26 + 2 * i);
        for (size_t i = 0; i < cl; i++) if (i < ctx_depth) term(15 + (int)i, 0, 30 + 2 * i);
        for (size_t i = 0; i < ll; i++) if (i < loop_depth) term(15 + (int)ctx_depth + (int)i, 0, 62 + 2 * i);
        const uint32_t nio = pass ? c->pub.num_outputs : c->pub.num_inputs;
        for (uint32_t i = 0; i < nio && i < 8; i++) {
            u128 v; memcpy(&v, pass ? c->pub.outputs[i] : c->pub.inputs[i], 16);
            term(i < sd ? 15 + (int)ctx_depth + (int)loop_depth + (int)i : -1, v, 78 + 2 * i);
        }
    }
    std::vector<fe> up(4 * W + 4);
    for (size_t i = 0; i < 4 * W; i++) up[i] = fe_from_u128(w[i]);
    for (int i = 0; i < 4; i++) up[4 * W + i] = fe_from_u128(g[i]);
    fe* d_w = (fe*)c->d_stage;                           // staging area is free until the openings
    // through the page-locked staging area: the copy is queued and the host moves on
    fe* h_w = reinterpret_cast<fe*>(c->h_stage + HS_WEIGHTS);
    memcpy(h_w, up.data(), up.size() * sizeof(fe));
    HIP_TRY(c, hipMemcpyAsync(d_w, h_w, up.size() * sizeof(fe), hipMemcpyHostToDevice, c->stream));
    fe* outs[4] = {o0, o1, o2, o3};                       // [pass][adj]
    if (ip) {                                            // the 8n-coefficient polynomials themselves (DISTAFF_COMBINE=steps)
        HIP_TRY(c, hipMemsetAsync(ip, 0, D * sizeof(fe), c->stream));
        HIP_TRY(c, hipMemsetAsync(fp, 0, D * sizeof(fe), c->stream));
        outs[0] = ip; outs[1] = ip + p; outs[2] = fp; outs[3] = fp + p;
    }
    k_lincomb4(c, c->polys, W, n, d_w, outs[0], outs[1], outs[2], outs[3]);
    for (int q = 0; q < 4; q++) k_sub_at0(c, outs[q], d_w + 4 * W + q);
    return DST_OK;
}

// DISTAFF_COMBINE=steps: combine_polys and the DEEP composition as the reference's sequence of whole-array steps (boundary polynomials of
// 8n coefficients, their divisions, additions; copy / division / multiply-adds of the composition) instead of the fused passes.  Tests
// run both; the boundary-by-evaluation route implies it.
bool dst_internal_combine_by_steps(const dst_ctx* c) { return DST_TEST_HOOKS && (c->sw_is("DISTAFF_COMBINE", "steps") || dst_internal_boundary_by_evaluation(c)); }

// What the fused combination (k_combine_fused) reads of the two boundary constraints: I = A + x^p A', F = C + x^p C' (see
// dst_internal_boundary_polys), p = 6n + 2, each of A, A', C, C' a linear combination of the trace polynomials with n coefficients.  Written
// behind a leading zero and divided in place -- A, A' by (x - 1), C, C' by (x - x_last) -- so that q4[k][0] is the sum / the value at
// x_last and q4[k][1 + i] the quotient coefficient i.  Four divisions over n + 1 coefficients instead of two over 8n.
int dst_internal_boundary_quotients(dst_ctx* c, const fe* draws344, fe* q4, size_t stride) {
    const size_t n = c->n;
    HIP_TRY(c, hipMemsetAsync(q4, 0, 4 * stride * sizeof(fe), c->stream));
    // ip = q4[0] (A at offset 0) ... dst_internal_boundary_polys writes A, A', C, C' at (ip, ip + p, fp, fp + p): hand it views whose
    // "+ p" lands on the next array
    int r = dst_internal_boundary_polys(c, draws344, nullptr, nullptr, q4 + 1, q4 + stride + 1, q4 + 2 * stride + 1, q4 + 3 * stride + 1);
    if (r) return r;
    fe* arrays[4] = {q4, q4 + stride, q4 + 2 * stride, q4 + 3 * stride};
    const fe divisors[4] = {fe_one(), fe_one(), c->x_last, c->x_last};
    k_syn_div_batch(c, arrays, divisors, 4, n + 1);            // one set of launches for the four
    return DST_OK;
}

// milliseconds between two phase-boundary events of the last proof (recorded on the stream: no host wait at the boundary itself)
static double event_ms(dst_ctx* c, int from, int to) {
    float ms = 0;
    return hipEventElapsedTime(&ms, c->ph_ev[from], c->ph_ev[to]) == hipSuccess ? (double)ms : 0.0;
}

int dst_eval_constraints(dst_ctx* c, const dst_public* pub, const uint8_t* coeffs, uint8_t constraint_root[32], int64_t* bad_step) {
    if (!c || !pub || !coeffs || !constraint_root) return DST_ERR_ARG;
    if (!c->committed) { c->err = "dst_eval_constraints: trace not committed"; return DST_ERR_STATE; }
    if (pub->num_inputs > 8 || pub->num_outputs > 8) { c->err = "too many public inputs / outputs"; return DST_ERR_ARG; }
    if (c->prm.world != 1) { c->err = "dst_eval_constraints: multi-GPU combination is driven by the host (see distaff_amd/sharded.py)"; return DST_ERR_ARG; }
    HIP_TRY(c, hipSetDevice(c->device));
    c->pub = *pub;
    const double t0 = wall_ms();
    HIP_TRY(c, hipEventRecord(c->ph_ev[0], c->stream));
    std::vector<fe> draws = copy_fe(coeffs, 344), tc;
    dst_internal_transition_coefficients(c, draws.data(), tc);
    fe* d_coef = c->scratch + c->scratch_elems - 1024;           // tail of the scratch area
    fe* d_tc = d_coef + 344;
    if (int ru = dst_internal_upload_draws(c, draws.data(), tc, d_coef, d_tc)) return ru;
    // The host does not wait for the evaluation's verdict (evaluator.rs:152-158) before it queues the combination: the flag travels back
    // with the constraint root, and a trace that fails is reported then (the work queued behind it is wasted only in that case).
    const bool steps = dst_internal_combine_by_steps(c);
    int r = k_eval_constraints(c, d_coef, d_tc, bad_step, /*defer_check=*/!steps);   // prover.rs:53-64
    if (r == DST_ERR_AIR) { c->err = "transition constraints were not satisfied"; return r; }
    if (r != DST_OK) return r;
    HIP_TRY(c, hipEventRecord(c->ph_ev[1], c->stream));
    // combine_polys (constraint_table.rs:54-88)
    const size_t n = c->n, D = 8 * n;
    fe* work = c->cwork + 3 * D;
    if (steps) {
        fe* ip = c->cwork; fe* fp = c->cwork + D; fe* tp = c->cwork + 2 * D;
        if (dst_internal_boundary_by_evaluation(c)) {
            k_intt8_cosets(c, c->ceval, ip, work);
            k_intt8_cosets(c, c->ceval + D, fp, work);
        } else if ((r = dst_internal_boundary_polys(c, draws.data(), ip, fp))) return r;
        k_syn_div(c, ip, D, fe_one());
        k_syn_div(c, fp, D, c->x_last);
        k_intt8_cosets(c, c->ceval + 2 * D, tp, work);
        k_syn_div_expanded(c, tp, c->cpoly, D, n, c->x_last);
        k_add(c, c->cpoly, ip, D);
        k_add(c, c->cpoly, fp, D);
    } else {
        // boundary quotients (4 x (n + 1) coefficients), the eight inverse coset transforms, then ONE pass: 8-point step across cosets,
        // division of the transition part, sum (k_combine_fused)
        fe* q4 = c->cwork; const size_t qs = n + 16;
        if ((r = dst_internal_boundary_quotients(c, draws.data(), q4, qs))) return r;
        k_intt_cosets_local(c, c->ceval + 2 * D, work, 8);
        k_combine_fused(c, work, q4, qs, c->cpoly);
    }
    HIP_TRY(c, hipEventRecord(c->ph_ev[2], c->stream));
    // constraint_poly.eval + Merkle tree over raw evaluation pairs (prover.rs:82-86)
    k_lde_fold8(c, c->cpoly, c->cevals);
    k_constraint_tree(c);
    HIP_TRY(c, hipMemcpyAsync(c->constraint_root, c->cnodes + 1, 32, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipEventRecord(c->ph_ev[3], c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    if (!steps) {
        c->air_flag_host = *reinterpret_cast<unsigned long long*>(c->h_stage + HS_AIR_FLAG);
        if ((r = k_constraint_check(c, bad_step)) != DST_OK) { c->err = "transition constraints were not satisfied"; return r; }
    }
    memcpy(constraint_root, c->constraint_root, 32);
    // phase times from the stream's own clock; what the host spent before the first and after the last event goes to the outer phases
    const double total = wall_ms() - t0, e1 = event_ms(c, 0, 1), e2 = event_ms(c, 1, 2), e3 = event_ms(c, 2, 3);
    const double rest = total > e1 + e2 + e3 ? total - (e1 + e2 + e3) : 0.0;
    c->phase_ms[2] = e1 + rest / 2; c->phase_ms[3] = e2; c->phase_ms[4] = e3 + rest / 2;
    c->constraints_done = true; c->composed = false;
    return DST_OK;
}

// ---- step 6 ---------------------------------------------------------------------------------------------------------------------
// the DeepValues land in page-locked memory; a caller that did not wait (dst_prove) picks them up at its next synchronisation
static void finish_deep_values(dst_ctx* c) {
    if (!c->deep_pending) return;
    const size_t W = c->W;
    c->deep_z1.assign(c->h_stage + HS_DEEP, c->h_stage + HS_DEEP + W * 16);
    c->deep_z2.assign(c->h_stage + HS_DEEP + HS_DEEP_HALF, c->h_stage + HS_DEEP + HS_DEEP_HALF + W * 16);
    c->deep_pending = false;
}
static int compose_impl(dst_ctx* c, const uint8_t* draws_bytes, uint8_t* trace_at_z1, uint8_t* trace_at_z2, bool wait);
int dst_compose(dst_ctx* c, const uint8_t* draws_bytes, uint8_t* trace_at_z1, uint8_t* trace_at_z2) {
    if (!c || !draws_bytes || !trace_at_z1 || !trace_at_z2) return DST_ERR_ARG;
    return compose_impl(c, draws_bytes, trace_at_z1, trace_at_z2, true);
}
static int compose_impl(dst_ctx* c, const uint8_t* draws_bytes, uint8_t* trace_at_z1, uint8_t* trace_at_z2, bool wait) {
    if (!c->constraints_done) { c->err = "dst_compose: constraints not evaluated"; return DST_ERR_STATE; }
    HIP_TRY(c, hipSetDevice(c->device));
    double t0 = wall_ms();
    HIP_TRY(c, hipEventRecord(c->ph_ev[0], c->stream));         // device-side extent of the composition (dst_prove does not wait for it)
    const size_t n = c->n, D = 8 * n, W = c->W;
    std::vector<fe> draws = copy_fe(draws_bytes, 516);
    const fe z = draws[0];
    const fe next_z = fe_mul(z, c->g_trace);
    const fe k1 = draws[513], k2 = draws[514], k3 = draws[515];
    fe* d_draws = c->scratch + c->scratch_elems - 1024;           // [516] draws, then [W] T(z), [W] T(z*g), [1] C(z)
    fe* d_tz1 = d_draws + 520; fe* d_tz2 = d_tz1 + 128; fe* d_cz = d_tz2 + 128;
    {   // queued from the page-locked staging area: `draws` goes out of scope while the copy may still be pending (wait == false)
        fe* h_draws = reinterpret_cast<fe*>(c->h_stage + HS_COMPOSE);
        memcpy(h_draws, draws.data(), 516 * 16);
        HIP_TRY(c, hipMemcpyAsync(d_draws, h_draws, 516 * 16, hipMemcpyHostToDevice, c->stream));
    }
    // trace_table.rs:206-261
    const bool steps = dst_internal_combine_by_steps(c);
    const fe zz[2] = {z, next_z};
    k_horner(c, c->polys, W, n, z, d_tz1);
    k_horner(c, c->polys, W, n, next_z, d_tz2);
    fe* t1 = c->cwork; fe* t2 = c->cwork + n; fe* cp = c->cwork + D;
    k_lincomb2(c, c->polys, W, n, d_draws + 1, 256, t1, t2);          // both combinations in one pass over the trace polynomials
    const size_t inc = 6 * n + 1;                                // get_incremental_trace_degree (utils/mod.rs:20)
    if (steps) {
        k_sub_dot_at0(c, t1, d_tz1, d_draws + 1, W);
        k_sub_dot_at0(c, t2, d_tz2, d_draws + 257, W);
        k_syn_div(c, t1, n, z);
        k_syn_div(c, t2, n, next_z);
        k_add(c, t1, t2, n);
        HIP_TRY(c, hipMemsetAsync(c->comp_poly, 0, D * 16, c->stream));
        k_axpy(c, c->comp_poly, t1, k1, n);
        k_axpy(c, c->comp_poly + inc, t1, k2, n);
        // constraint_poly.rs:39-52 merge_into
        k_horner(c, c->cpoly, 1, D, z, d_cz);
        HIP_TRY(c, hipMemcpyAsync(cp, c->cpoly, D * 16, hipMemcpyDeviceToDevice, c->stream));
        k_sub_at0(c, cp, d_cz);
        k_syn_div(c, cp, D, z);
        k_axpy(c, c->comp_poly, cp, k3, D);
    } else {
        // The constant term of a dividend enters no coefficient of its quotient by (x - b), so the subtractions of T(z), T(z g) and C(z)
        // (trace_table.rs:226-233, constraint_poly.rs:44) need not be made -- and C(z) need not be evaluated.  The division of the
        // constraint polynomial writes the composition polynomial directly: k3 * quotient + (k1 + k2 x^inc) * t1.
        fe* both[2] = {t1, t2};
        k_syn_div_batch(c, both, zz, 2, n);
        k_add(c, t1, t2, n);
        k_syn_div_compose(c, c->cpoly, c->comp_poly, D, z, t1, n, inc, k1, k2, k3);
    }
    // evaluate over the LDE domain (prover.rs:98-101)
    k_lde_fold8(c, c->comp_poly, c->comp);
    HIP_TRY(c, hipEventRecord(c->ph_ev[1], c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->h_stage + HS_DEEP, d_tz1, W * 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->h_stage + HS_DEEP + HS_DEEP_HALF, d_tz2, W * 16, hipMemcpyDeviceToHost, c->stream));
    c->deep_pending = true;
    if (wait) {
        CTX_SYNC(c, "the DEEP composition");
        HIP_TRY(c, hipGetLastError());
        finish_deep_values(c);
        memcpy(trace_at_z1, c->deep_z1.data(), W * 16);
        memcpy(trace_at_z2, c->deep_z2.data(), W * 16);
    }
    c->phase_ms[5] = wall_ms() - t0;        // without the wait: the host's share only; the first FRI layer's wait absorbs the rest (dst_prove adds it back)
    c->composed = true; c->fri_committed = 0; c->fri_folded = 0; c->fri_roots.clear(); c->fri_tail_pending = false;
    return DST_OK;
}

// ---- step 7 ---------------------------------------------------------------------------------------------------------------------
int dst_fri_commit_layer(dst_ctx* c, uint8_t layer_root[32], int* more) {
    if (!c || !layer_root || !more) return DST_ERR_ARG;
    if (!c->composed) { c->err = "dst_fri_commit_layer: composition not built"; return DST_ERR_STATE; }
    int d = c->fri_committed;
    if (d >= c->num_fri_layers || d != c->fri_folded) { c->err = "dst_fri_commit_layer: fold the previous layer first"; return DST_ERR_STATE; }
    HIP_TRY(c, hipSetDevice(c->device));
    double t0 = wall_ms();
    if (d == 0) k_fri_leaves_layer0(c); else k_fri_leaves(c, d);
    k_merkle_levels(c, c->fri_leaves[d], c->fri_nodes[d], c->fri_size[d] / 4);
    uint8_t root[32];
    HIP_TRY(c, hipMemcpyAsync(root, c->fri_nodes[d] + 1, 32, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    memcpy(layer_root, root, 32);
    c->fri_roots.push_back(std::vector<uint8_t>(root, root + 32));
    c->fri_committed = d + 1;
    *more = (d + 1 < c->num_fri_layers) ? 1 : 0;
    if (d == 0) c->phase_ms[6] = 0;
    c->phase_ms[6] += wall_ms() - t0;
    return DST_OK;
}
// The rest of the FRI commit phase in ONE launch once the next layer to commit is small (k_fri_tail: rows hashed, trees built, x drawn
// from every root and the folds done by one workgroup; DISTAFF_FRI_TAIL=0 keeps the per-layer launches, tests compare both).  Returns
// 1 when the tail ran (all remaining roots appended), 0 when the next layer is still too large, < 0 on error.
#define DST_FRI_TAIL_MAX_SIZE ((size_t)1 << 13)
int dst_internal_fri_tail(dst_ctx* c, std::vector<uint8_t>& roots) {
    const int d = c->fri_committed;
    if (d < 1 || d != c->fri_folded || d >= c->num_fri_layers || c->fri_size[d] > DST_FRI_TAIL_MAX_SIZE) return 0;
    if (const char* e = c->sw("DISTAFF_FRI_TAIL")) if (e[0] == '0') return 0;
    double t0 = wall_ms();
    const int count = c->num_fri_layers - d;
    std::vector<uint8_t> r((size_t)count * 32);
    int rc = k_fri_tail(c, d, r.data());
    if (rc) return rc;
    for (int i = 0; i < count; i++) c->fri_roots.push_back(std::vector<uint8_t>(r.begin() + 32 * i, r.begin() + 32 * (i + 1)));
    roots.insert(roots.end(), r.begin(), r.end());
    c->fri_committed = c->num_fri_layers; c->fri_folded = c->num_fri_layers - 1;
    c->phase_ms[6] += wall_ms() - t0;
    return 1;
}
int dst_fri_fold(dst_ctx* c, const uint8_t special_x[16]) {
    if (!c || !special_x) return DST_ERR_ARG;
    int d = c->fri_folded;
    if (d + 1 != c->fri_committed || d + 1 >= c->num_fri_layers) { c->err = "dst_fri_fold: nothing to fold"; return DST_ERR_STATE; }
    HIP_TRY(c, hipSetDevice(c->device));
    double t0 = wall_ms();
    k_fri_fold(c, d, fe_from_bytes(special_x));
    c->fri_folded = d + 1;
    c->phase_ms[6] += wall_ms() - t0;        // launch only; the next commit synchronises
    return DST_OK;
}

// ---- step 8 ---------------------------------------------------------------------------------------------------------------------
int dst_pow_grind(dst_ctx* c, const uint8_t seed[32], uint32_t grinding_factor, uint8_t out_seed[32], uint64_t* nonce) {
    if (!c || !seed || !out_seed || !nonce || grinding_factor > 64) return DST_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    int r = k_pow(c, seed, grinding_factor, nonce);
    if (r != DST_OK) return r;
    uint8_t buf[64];
    memset(buf, 0, 64);
    memcpy(buf, seed, 32);
    memcpy(buf + 32, nonce, 8);
    blake3_short(buf, 64, out_seed);
    return DST_OK;
}

// ---- step 9 ---------------------------------------------------------------------------------------------------------------------
// (openings are planned, gathered and serialised in shard.hip: dst_shard_open / dst_shard_assemble)
int dst_build_proof(dst_ctx* c, const uint64_t* positions_in, uint32_t num_positions, uint64_t pow_nonce, uint8_t* out, size_t cap, size_t* out_len) {
    if (!c || !positions_in || !out_len) return DST_ERR_ARG;
    if (!c->composed || c->fri_committed != c->num_fri_layers) { c->err = "dst_build_proof: FRI commit phase not finished"; return DST_ERR_STATE; }
    if (c->prm.world != 1) { c->err = "dst_build_proof: single-GPU contexts only"; return DST_ERR_ARG; }
    HIP_TRY(c, hipSetDevice(c->device));
    double t0 = wall_ms();
    finish_deep_values(c);                                      // every FRI layer since has synchronised the stream
    // one plan, one batched device gather, one fill (shard.hip; the single-GPU case is the plan with every item local)
    std::vector<uint8_t> proof;
    int rc = dst_internal_build_proof(c, positions_in, num_positions, pow_nonce, proof);
    if (rc == DST_OK) {
        *out_len = proof.size();
        if (out) {
            if (cap < proof.size()) { c->err = "proof buffer too small"; rc = DST_ERR_ARG; }
            else memcpy(out, proof.data(), proof.size());
        }
    }
    c->phase_ms[8] = wall_ms() - t0;
    return rc;
}

// fri::reduce (fri/prover.rs:11-53) for dst_prove.  chained = true: the layers above the single-launch tail are committed WITHOUT host
// round trips -- x = prng(root) is drawn on the device from the root where it lies (fri_draw_kernel, the same ChaCha20 / Uniform statement
// the tail kernel uses) and the fold reads it from device memory; all roots are read back once, with the tail's.  chained = false
// (DISTAFF_FRI_CHAIN=0, tests): one root read-back and one host draw per layer through the public phase calls.
static int fri_commit_all(dst_ctx* c, std::vector<uint8_t>& roots, bool chained) {
    int rc;
    if (!chained) {
        for (;;) {
            if (int rt = dst_internal_fri_tail(c, roots)) { if (rt < 0) return rt; break; }
            uint8_t root[32]; int more = 0;
            if ((rc = dst_fri_commit_layer(c, root, &more))) return rc;
            roots.insert(roots.end(), root, root + 32);
            if (!more) break;
            fe sx = prng(root);
            if ((rc = dst_fri_fold(c, (const uint8_t*)&sx))) return rc;
        }
        return DST_OK;
    }
    if (!c->composed || c->fri_committed != 0 || c->fri_folded != 0) { c->err = "dst_prove: FRI state"; return DST_ERR_STATE; }
    const int L = c->num_fri_layers;
    digest* d_roots = reinterpret_cast<digest*>(c->d_fri_chain);
    fe* d_alpha = reinterpret_cast<fe*>(c->d_fri_chain + DST_MAX_FRI_LAYERS * 32);
    const char* te = c->sw("DISTAFF_FRI_TAIL");
    const bool tail_on = !(te && te[0] == '0');
    int d = 0;
    for (; d < L; d++) {
        if (tail_on && d >= 1 && c->fri_size[d] <= DST_FRI_TAIL_MAX_SIZE) break;          // the rest in one launch
        if (d == 0) k_fri_leaves_layer0(c); else k_fri_leaves(c, d);
        k_merkle_levels(c, c->fri_leaves[d], c->fri_nodes[d], c->fri_size[d] / 4);
        k_fri_draw(c, d, d_alpha + d, d_roots + d);
        if (d + 1 < L) k_fri_fold_dev(c, d, d_alpha + d);
    }
    const int big = d;                                                // layers committed by the per-layer kernels
    uint8_t* h_roots = c->h_stage + HS_FRI_ROOTS;                      // page-locked: queued, picked up after the wait below
    if (big) HIP_TRY(c, hipMemcpyAsync(h_roots, d_roots, (size_t)big * 32, hipMemcpyDeviceToHost, c->stream));
    c->fri_committed = big; c->fri_folded = big < L ? big : L - 1;
    std::vector<uint8_t> tail_roots;
    if (big < L) {
        if (int rt = dst_internal_fri_tail(c, tail_roots)) { if (rt < 0) return rt; }     // synchronises the stream; appends its roots to c->fri_roots
        else { c->err = "dst_prove: the FRI tail did not run"; return DST_ERR_STATE; }
    } else {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipGetLastError());
    }
    // c->fri_roots in layer order: the tail pushed its own behind what was there (nothing yet): put the big layers' in front
    std::vector<std::vector<uint8_t>> all;
    for (int i = 0; i < big; i++) all.emplace_back(h_roots + 32 * i, h_roots + 32 * (i + 1));
    for (auto& r : c->fri_roots) all.push_back(r);
    c->fri_roots.swap(all);
    for (auto& r : c->fri_roots) roots.insert(roots.end(), r.begin(), r.end());
    c->fri_committed = L; c->fri_folded = L - 1;
    return DST_OK;
}

// ---- the whole prover -----------------------------------------------------------------------------------------------------------
int dst_prove(dst_ctx* c, const dst_public* pub, uint8_t* proof_out, size_t cap, size_t* proof_len) {
    if (!c || !pub || !proof_len) return DST_ERR_ARG;
    int rc;
    uint8_t trace_root[32], constraint_root[32];
    if ((rc = dst_commit_trace(c, trace_root))) return rc;
    std::vector<fe> coef(344);
    prng_vector(trace_root, 344, coef.data());                   // ConstraintCoefficients::new (coefficients.rs:66)
    int64_t bad = -1;
    if ((rc = dst_eval_constraints(c, pub, (const uint8_t*)coef.data(), constraint_root, &bad))) return rc;
    std::vector<fe> draws(516);
    prng_vector(constraint_root, 516, draws.data());             // z = draw 0; CompositionCoefficients (coefficients.rs:82)
    // the DEEP values are not needed before the openings: the host queues the composition and goes on to the first FRI layer
    std::vector<uint8_t> z1(c->W * 16), z2(c->W * 16);
    const double t_compose = wall_ms();
    if ((rc = compose_impl(c, (const uint8_t*)draws.data(), z1.data(), z2.data(), false))) return rc;
    std::vector<uint8_t> roots;
    {
        const char* ce = c->sw("DISTAFF_FRI_CHAIN");
        if ((rc = fri_commit_all(c, roots, !(ce && ce[0] == '0')))) return rc;
        // the commit phase's first wait covered the composition too: split at the device's own boundary (events of compose_impl)
        const double both = wall_ms() - t_compose;
        float dev = 0;
        if (hipEventElapsedTime(&dev, c->ph_ev[0], c->ph_ev[1]) == hipSuccess && dev > 0 && dev < both) { c->phase_ms[5] = dev; c->phase_ms[6] = both - dev; }
        else { c->phase_ms[6] = both > c->phase_ms[5] ? both - c->phase_ms[5] : 0.0; }
    }
    double t0 = wall_ms();
    uint8_t seed0[32], seed1[32];
    if (!blake3_short(roots.data(), roots.size(), seed0)) { c->err = "too many FRI roots"; return DST_ERR_ARG; }   // prover.rs:120-127
    uint64_t nonce = 0;
    if ((rc = dst_pow_grind(c, seed0, c->prm.grinding_factor, seed1, &nonce))) return rc;
    std::vector<uint64_t> positions;
    if (query_positions(seed1, c->N, (uint32_t)c->B, c->prm.num_queries, positions)) { c->err = "could not generate enough query positions"; return DST_ERR_ARG; }
    c->phase_ms[7] = wall_ms() - t0;
    return dst_build_proof(c, positions.data(), (uint32_t)positions.size(), nonce, proof_out, cap, proof_len);
}

// ---- host helpers ----------------------------------------------------------------------------------------------------------------
void dst_prng_vector(const uint8_t seed[32], uint32_t count, uint8_t* out) {
    std::vector<fe> v(count);
    prng_vector(seed, count, v.data());
    memcpy(out, v.data(), (size_t)count * 16);
}
int dst_query_positions(const uint8_t seed[32], uint64_t domain_size, uint32_t blowup, uint32_t num_queries, uint64_t* out) {
    // an error code, never an abort: a zero domain or extension factor would divide by zero, and the caller's buffer holds at most 128 positions
    if (!seed || !out || domain_size == 0 || blowup == 0 || num_queries == 0 || num_queries > 128) return DST_ERR_ARG;
    std::vector<uint64_t> p;
    if (query_positions(seed, domain_size, blowup, num_queries, p)) return DST_ERR_ARG;
    memcpy(out, p.data(), p.size() * 8);
    return (int)p.size();
}
void dst_blake3(const uint8_t* in, size_t len, uint8_t out[32]) { if (!blake3_short(in, len, out)) memset(out, 0, 32); }

int dst_fibonacci_trace(uint32_t log_n, uint8_t* cols, uint8_t program_hash[32], uint8_t result[16]) {
    if (!cols || !program_hash || !result || log_n < 7 || log_n > 26) return DST_ERR_ARG;
    u128 ph[2], res;
    int r = fibonacci_trace(log_n, (u128*)cols, ph, &res);
    if (r) return DST_ERR_ARG;
    memcpy(program_hash, ph, 32);
    memcpy(result, &res, 16);
    return DST_OK;
}

// ---- inspection --------------------------------------------------------------------------------------------------------------------
int dst_read_buffer(dst_ctx* c, uint32_t what, uint32_t arg, uint8_t* out, size_t cap, size_t* len) {
    if (!c || !len) return DST_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = c->n, Nl = c->Bc * n;
    const void* src = nullptr; size_t bytes = 0;
    const fe* coset_major = nullptr; size_t cosets = 0;
    switch (what) {
        case DST_BUF_POLYS: src = c->polys; bytes = c->W * n * 16; break;
        case DST_BUF_LDE: if (arg >= c->W) return DST_ERR_ARG; coset_major = c->lde + (size_t)arg * Nl; cosets = c->Bc; break;
        case DST_BUF_TRACE_LEAVES: src = c->trace_leaves; bytes = Nl * 32; break;
        case DST_BUF_TRACE_NODES: src = c->trace_nodes; bytes = Nl * 32; break;
        case DST_BUF_CEVAL_I: case DST_BUF_CEVAL_F: case DST_BUF_CEVAL_T: coset_major = c->ceval + (size_t)(what - DST_BUF_CEVAL_I) * 8 * n; cosets = 8; break;
        case DST_BUF_CPOLY: src = c->cpoly; bytes = 8 * n * 16; break;
        case DST_BUF_CEVALS: coset_major = c->cevals; cosets = c->Bc; break;
        case DST_BUF_CNODES: src = c->cnodes; bytes = Nl / 2 * 32; break;
        case DST_BUF_COMP_POLY: src = c->comp_poly; bytes = 8 * n * 16; break;
        case DST_BUF_COMP_EVALS: coset_major = c->comp; cosets = c->Bc; break;
        case DST_BUF_FRI_EVALS:
            if ((int)arg >= c->num_fri_layers) return DST_ERR_ARG;
            if (arg == 0) { coset_major = c->comp; cosets = c->Bc; } else { src = c->fri_e[arg]; bytes = c->fri_size[arg] * 16; }
            break;
        case DST_BUF_FRI_NODES: if ((int)arg >= c->num_fri_layers) return DST_ERR_ARG; src = c->fri_nodes[arg]; bytes = c->fri_size[arg] / 4 * 32; break;
        case DST_BUF_FRI_LEAVES: if ((int)arg >= c->num_fri_layers) return DST_ERR_ARG; src = c->fri_leaves[arg]; bytes = c->fri_size[arg] / 4 * 32; break;
        default: c->err = "unknown buffer id"; return DST_ERR_ARG;
    }
    if (coset_major) bytes = cosets * n * 16;
    *len = bytes;
    if (!out) return DST_OK;
    if (cap < bytes) { c->err = "output buffer too small"; return DST_ERR_ARG; }
    if (coset_major) {
        fe* tmp = nullptr;
        HIP_TRY(c, hipMalloc((void**)&tmp, bytes));
        k_coset_to_natural(c, coset_major, cosets, tmp);
        hipError_t e = hipMemcpyAsync(out, tmp, bytes, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        hipFree(tmp);
        HIP_TRY(c, e);
    } else {
        HIP_TRY(c, hipMemcpyAsync(out, src, bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return DST_OK;
}

int dst_field_op(dst_ctx* c, int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t count) {
    if (!c || !a || !b || !out) return DST_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    return k_field_op(c, op, a, b, out, count);
}
int dst_set_profiling(dst_ctx* c, int level) { if (!c || level < 0 || level > 2) return DST_ERR_ARG; c->profile = level; return DST_OK; }
int dst_kernel_stats(dst_ctx* c, char* json_out, size_t cap, int reset) {
    if (!c || !json_out) return DST_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (auto& e : c->kpending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, e.e0, e.e1) == hipSuccess) { auto& st = c->kstats[e.name]; st.launches++; st.ms += ms; st.bytes += e.bytes; st.mads += e.mads; }
        c->event_pool.push_back(e.e0); c->event_pool.push_back(e.e1);
    }
    c->kpending.clear();
    std::string js = "{";
    bool first = true;
    for (auto& kv : c->kstats) {
        char buf[256];
        snprintf(buf, sizeof(buf), "%s\"%s\": {\"launches\": %llu, \"ms\": %.6f, \"bytes\": %.0f, \"mads\": %.0f}", first ? "" : ", ", kv.first.c_str(),
                 (unsigned long long)kv.second.launches, kv.second.ms, kv.second.bytes, kv.second.mads);
        js += buf; first = false;
    }
    js += "}";
    if (reset) c->kstats.clear();
    if (js.size() + 1 > cap) { c->err = "stats buffer too small"; return DST_ERR_ARG; }
    memcpy(json_out, js.c_str(), js.size() + 1);
    return DST_OK;
}

// Calibration kernels (dependent multiplication chains, the multiply-add peak, the straight-line-code probe): laboratory instruments of
// bench.py and the tests, compiled only into the test / bench build (libdistaff_hip_hooks.so).  The product library keeps the entry
// points and answers DST_ERR_STATE.
#if DST_TEST_HOOKS
int dst_bench_mulmod(dst_ctx* c, uint64_t lanes, uint32_t iters, double* ms) {
    if (!c || !ms) return DST_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    return k_bench_mulmod(c, lanes, iters, ms);
}
int dst_bench_code(dst_ctx* c, uint32_t code_kib, double* ms) {
    if (!c || !ms) return DST_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    return k_bench_code(c, code_kib, ms);
}
int dst_bench_clock(dst_ctx* c, uint64_t lanes, uint32_t iters, double* mhz) {
    if (!c || !mhz) return DST_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    return k_bench_clock(c, lanes, iters, mhz);
}
int dst_bench_mad(dst_ctx* c, uint64_t lanes, uint32_t iters, double* ms) {
    if (!c || !ms) return DST_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    return k_bench_mad(c, lanes, iters, ms);
}
#else
static int no_hooks(dst_ctx* c) { if (c) c->err = "calibration kernels are part of the test / bench build (libdistaff_hip_hooks.so) only"; return c ? DST_ERR_STATE : DST_ERR_ARG; }
int dst_bench_mulmod(dst_ctx* c, uint64_t, uint32_t, double*) { return no_hooks(c); }
int dst_bench_code(dst_ctx* c, uint32_t, double*) { return no_hooks(c); }
int dst_bench_mad(dst_ctx* c, uint64_t, uint32_t, double*) { return no_hooks(c); }
int dst_bench_clock(dst_ctx* c, uint64_t, uint32_t, double*) { return no_hooks(c); }
#endif
// 1 when this is the test / bench build
int dst_test_hooks(void) { return DST_TEST_HOOKS; }

}  // extern "C"
