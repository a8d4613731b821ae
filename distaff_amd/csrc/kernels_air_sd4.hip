// AIR kernel instances for traces with <= 2 context registers, no loop register and a user stack of depth 4 (the
// Fibonacci shape).  The evaluation is cut into five launches (see air_kernel.h): every one stays inside the register file without
// scratch and inside the 64 KiB instruction cache of a CU pair.
#include "air_kernel.h"
void air_launch_sd4(dst_ctx* c, const AirArgs& a, uint32_t Q) {
    // the boundary combinations are normally written in coefficient form (api.hip dst_internal_boundary_polys), not evaluated
#if DST_TEST_HOOKS
    if (dst_internal_boundary_by_evaluation(c)) launch_air<2, 1, 4, 8, 1, 0, 0>(c, a, Q);           // boundary constraints
#endif
    launch_air<2, 1, 4, 8, 130, 0, AF_FIRST>(c, a, Q);                                             // op bits, loop image, context / loop stacks (starts the partial sums)
    launch_air<2, 1, 4, 8, 4, 0, 0>(c, a, Q);                                                      // sponge
    launch_air<2, 1, 4, 8, 0, AG_HIGH, AF_EV_OUT>(c, a, Q);                                        // stack: PUSH, CMP, RESCR, BEGIN / NOOP
    launch_air<2, 1, 4, 8, 0, AG_LOW0, AF_EV_IN | AF_EV_OUT>(c, a, Q);                             // stack: low-degree operations 0x00 .. 0x0F, both auxiliary constraints
    launch_air<2, 1, 4, 8, 0, AG_LOW1, AF_EV_IN | AF_LAST>(c, a, Q);                               // stack: low-degree operations 0x10 .. 0x1F; emits the stack constraints; combination
}
