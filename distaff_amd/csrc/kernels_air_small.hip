// AIR kernel instances for traces with <= 2 context registers, <= 1 loop register and a user stack of depth <= 8 (depth known
// at run time); cut into launches by constraint section and, for the stack, by operation group (air_kernel.h): every launch without
// scratch and inside the 64 KiB instruction cache of a CU pair.
#include "air_kernel.h"
void air_launch_small(dst_ctx* c, const AirArgs& a, uint32_t Q) {
#if DST_TEST_HOOKS
    if (dst_internal_boundary_by_evaluation(c)) launch_air<2, 1, 0, 8, 1, 0, 0>(c, a, Q);           // boundary constraints
#endif
    launch_air<2, 1, 0, 8, 130, 0, AF_FIRST>(c, a, Q);                                             // op bits, loop image, context / loop stacks (starts the partial sums)
    launch_air<2, 1, 0, 8, 4, 0, 0>(c, a, Q);                                                      // sponge
    launch_air<2, 1, 0, 8, 0, AG_RESCR, AF_EV_OUT>(c, a, Q);                                       // stack: RESCR
    launch_air<2, 1, 0, 8, 0, AG_PUSH | AG_CMP | AG_KEEP, AF_EV_IN | AF_EV_OUT>(c, a, Q);          // stack: PUSH, CMP, BEGIN / NOOP
    launch_air<2, 1, 0, 8, 0, AG_LOW(0, 0) | AG_LOW(0, 2), AF_EV_IN | AF_EV_OUT>(c, a, Q);         // stack: operations 0x00 .. 0x03, 0x08 .. 0x0B
    launch_air<2, 1, 0, 8, 0, AG_LOW(0, 1) | AG_LOW(0, 3), AF_EV_IN | AF_EV_OUT>(c, a, Q);         // stack: operations 0x04 .. 0x07, 0x0C .. 0x0F
    launch_air<2, 1, 0, 8, 0, AG_LOW(1, 2) | AG_LOW(1, 3), AF_EV_IN | AF_EV_OUT>(c, a, Q);         // stack: operations 0x18 .. 0x1F
    launch_air<2, 1, 0, 8, 0, AG_LOW(1, 0) | AG_LOW(1, 1), AF_EV_IN | AF_LAST>(c, a, Q);           // stack: operations 0x10 .. 0x17; emits the stack constraints; combination
}
