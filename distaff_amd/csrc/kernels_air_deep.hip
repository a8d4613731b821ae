// AIR kernel instances for any trace shape the VM can produce (up to 16 context, 8 loop and 32 stack registers, known at run time)
// in the nested-sum formulation: the first eight stack slots as in the depth <= 8 instances (from twelve register-resident items of
// the current row), deeper slots from memory as flag sums times shifted differences (air_kernel.h, AG_DEEP).  The per-operation
// formulation of kernels_air.hip stays as an independent statement of the same constraints (DISTAFF_AIR=generic).
#include "air_kernel.h"
void air_launch_deep(dst_ctx* c, const AirArgs& a, uint32_t Q) {
#if DST_TEST_HOOKS
    if (dst_internal_boundary_by_evaluation(c)) launch_air<16, 8, 0, 12, 1, 0, 0>(c, a, Q);          // boundary constraints
#endif
    launch_air<16, 8, 0, 12, 2, 0, AF_FIRST>(c, a, Q);                                              // op bits (starts the partial sums)
    launch_air<16, 8, 0, 12, 4, 0, 0>(c, a, Q);                                                     // sponge
    launch_air<16, 8, 0, 12, 128, 0, 0>(c, a, Q);                                                   // loop image, context / loop stacks
    launch_air<16, 8, 0, 12, 0, AG_DEEP, 0>(c, a, Q);                                               // stack slots 8..: flag sums x shifted differences
    launch_air<16, 8, 0, 12, 0, AG_RESCR, AF_EV_OUT>(c, a, Q);                                      // stack slots 0..7: RESCR
    launch_air<16, 8, 0, 12, 0, AG_PUSH | AG_CMP | AG_KEEP, AF_EV_IN | AF_EV_OUT>(c, a, Q);         // PUSH, CMP, BEGIN / NOOP
    launch_air<16, 8, 0, 12, 0, AG_LOW(0, 0) | AG_LOW(0, 2), AF_EV_IN | AF_EV_OUT>(c, a, Q);        // operations 0x00 .. 0x03, 0x08 .. 0x0B
    launch_air<16, 8, 0, 12, 0, AG_LOW(0, 1) | AG_LOW(0, 3), AF_EV_IN | AF_EV_OUT>(c, a, Q);        // operations 0x04 .. 0x07, 0x0C .. 0x0F
    launch_air<16, 8, 0, 12, 0, AG_LOW(1, 2) | AG_LOW(1, 3), AF_EV_IN | AF_EV_OUT>(c, a, Q);        // operations 0x18 .. 0x1F
    launch_air<16, 8, 0, 12, 0, AG_LOW(1, 0) | AG_LOW(1, 1), AF_EV_IN | AF_LAST>(c, a, Q);          // operations 0x10 .. 0x17; emits slots 0..7; combination
}
