// Launcher of the AIR constraint kernel (air_kernel.h) and its fully generic instance.
#include "air_kernel.h"
#include "rescue_constants.h"

#if DST_TEST_HOOKS            // the per-operation formulation: an independent statement of the constraints for the tests (DISTAFF_AIR=generic)
void air_launch_generic(dst_ctx* c, const AirArgs& a, uint32_t Q) {
    // any shape the VM can produce (up to 16 context, 8 loop and 32 stack registers, known at run time), per-operation formulation,
    // cut into the same section launches as the specialised instances
    launch_air<16, 8, 0, 32, 2, 0, AF_FIRST>(c, a, Q);     // op bits
    if (dst_internal_boundary_by_evaluation(c)) launch_air<16, 8, 0, 32, 1, 0, 0>(c, a, Q);    // boundary (normally in coefficient form, api.hip)
    launch_air<16, 8, 0, 32, 132, 0, 0>(c, a, Q);          // sponge, loop image, context / loop stacks
    launch_air<16, 8, 0, 32, 8, 0, 0>(c, a, Q);            // stack: low-degree ops that move items
    launch_air<16, 8, 0, 32, 32, 0, 0>(c, a, Q);           // stack: low-degree arithmetic / selection ops
    launch_air<16, 8, 0, 32, 16, 0, 0>(c, a, Q);           // stack: PUSH, CMP, BEGIN / NOOP
    launch_air<16, 8, 0, 32, 64, 0, AF_LAST>(c, a, Q);     // stack: RESCR + combination
}
#endif


int k_constraint_check(dst_ctx* c, int64_t* bad_step) {
    const unsigned long long res = c->air_flag_host;
    if (res != ~0ull) { if (bad_step) *bad_step = (int64_t)res; return DST_ERR_AIR; }
    if (bad_step) *bad_step = -1;
    return DST_OK;
}
int k_eval_constraints(dst_ctx* c, const fe* coeffs_dev, const fe* tc_dev, int64_t* bad_step, bool defer_check) {
    if (!c->air_consts) {
        AirConsts h;
        memcpy(h.sponge_mds, SPONGE_MDS, sizeof(h.sponge_mds)); memcpy(h.sponge_inv_mds, SPONGE_INV_MDS, sizeof(h.sponge_inv_mds));
        memcpy(h.hasher_mds, HASHER_MDS, sizeof(h.hasher_mds)); memcpy(h.hasher_inv_mds, HASHER_INV_MDS, sizeof(h.hasher_inv_mds));
        HIP_TRY(c, hipMalloc(&c->air_consts, sizeof(AirConsts)));
        HIP_TRY(c, hipMemcpy(c->air_consts, &h, sizeof(AirConsts), hipMemcpyHostToDevice));
    }
    AirArgs a{};
    a.lde = c->lde; a.out = c->ceval; a.coef = coeffs_dev; a.tc = tc_dev; a.periodic = c->periodic; a.consts = (const AirConsts*)c->air_consts; a.partial = c->cwork;
    // partial values of the stack constraints between launches: buffers that are idle during the evaluation -- the staging buffer of the
    // transforms (>= 4 registers x Bc cosets = (B / 2) of these arrays for B >= 16) and the two polynomials written after this phase
    {
        const size_t per = (size_t)(c->Bc / (c->B / 8)) * c->n;
        if (8 * per > c->Bc * c->tmp_regs * c->n) { c->err = "constraint evaluation: staging buffer too small for the stack partial values"; return DST_ERR_STATE; }
        for (int i = 0; i < 8; i++) a.ev[i] = c->tmp + (size_t)i * per;
        a.ev[8] = c->cpoly; a.ev[9] = c->comp_poly;
    }
    a.tw_lo = c->tw_lo; a.tw_hi = c->tw_hi; a.lo_bits = c->tw_lo_bits;
    a.bad_step = (unsigned long long*)c->d_u64;
    a.n = c->n; a.col_stride = c->Bc * c->n;
    a.log_n = c->log_n; a.log_N = c->log_N; a.log_b = c->log_b; a.W = (uint32_t)c->W;
    a.ctx_depth = c->prm.ctx_depth; a.loop_depth = c->prm.loop_depth; a.stack_depth = (uint32_t)c->stack_depth;
    a.coset_step = (uint32_t)(c->B / 8);
    a.q0 = (uint32_t)(c->j0 / (c->B / 8));
    uint32_t Q = (uint32_t)(c->Bc / (c->B / 8));
    a.num_inputs = c->pub.num_inputs; a.num_outputs = c->pub.num_outputs;
    memcpy(a.inputs, c->pub.inputs, sizeof(a.inputs)); memcpy(a.outputs, c->pub.outputs, sizeof(a.outputs));
    memcpy(a.program_hash, c->program_hash, sizeof(a.program_hash));
    a.op_count = fe_from_u64(c->op_count);
    HIP_TRY(c, hipMemsetAsync(c->d_u64, 0xFF, 8, c->stream));          // ~0 = no failing step yet
    const uint32_t cd = c->prm.ctx_depth, lp = c->prm.loop_depth, sd = (uint32_t)c->stack_depth;
    a.cl = cd > 1 ? cd : 1; a.ll = lp > 1 ? lp : 1; a.sl = sd > 8 ? sd : 8;
    // DISTAFF_AIR=generic|deep|small forces a more general instance than the shape needs (tests run the same trace through all of them)
    const char* force = c->sw("DISTAFF_AIR");
    const bool want_generic = force && !strcmp(force, "generic"), want_small = force && !strcmp(force, "small"), want_deep = force && !strcmp(force, "deep");
    const bool general = want_generic || want_deep;
    if (a.cl <= 2 && a.ll <= 1 && sd == 4 && !general && !want_small) air_launch_sd4(c, a, Q);
    else if (a.cl <= 2 && a.ll <= 1 && sd <= 8 && !general) air_launch_small(c, a, Q);
#if DST_TEST_HOOKS
    else if (want_generic) air_launch_generic(c, a, Q);
#endif
    else air_launch_deep(c, a, Q);
    // the failing step, if any (evaluator.rs:152-158 panics there): read back into page-locked memory; with defer_check the host does not
    // wait here -- the caller looks at it (k_constraint_check) at its next synchronisation, after work that does not depend on it
    unsigned long long* flag = c->h_stage ? reinterpret_cast<unsigned long long*>(c->h_stage + HS_AIR_FLAG) : &c->air_flag_host;
    HIP_TRY(c, hipMemcpyAsync(flag, c->d_u64, 8, hipMemcpyDeviceToHost, c->stream));
    if (defer_check && c->h_stage) return DST_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->air_flag_host = *flag;
    return k_constraint_check(c, bad_step);
}
