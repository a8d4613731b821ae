// AIR kernel instances for traces with <= 2 context registers, no loop register and a user stack of depth 4 (the
// Fibonacci shape): the evaluation is split into four launches (see air_kernel.h) to stay inside the register file.
#include "air_kernel.h"
void air_launch_sd4(dst_ctx* c, const AirArgs& a, uint32_t Q) {
    launch_air<2, 1, 4, 8, 3, true, false>(c, a, Q);      // boundary + op bits
    launch_air<2, 1, 4, 8, 4, false, false>(c, a, Q);     // sponge, loop image, context / loop stacks
    launch_air<2, 1, 4, 8, 8, false, false>(c, a, Q);     // stack: low-degree ops
    launch_air<2, 1, 4, 8, 16, false, true>(c, a, Q);     // stack: PUSH, CMP, RESCR, BEGIN/NOOP + combination
}
