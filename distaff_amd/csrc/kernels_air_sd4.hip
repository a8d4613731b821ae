// AIR kernel instances for traces with <= 2 context registers, no loop register and a user stack of depth 4 (the
// Fibonacci shape): the evaluation is split into three launches (see air_kernel.h) to stay inside the register file.
#include "air_kernel.h"
void air_launch_sd4(dst_ctx* c, const AirArgs& a, uint32_t Q) {
    // the boundary combinations are normally written in coefficient form (api.hip dst_internal_boundary_polys), not evaluated
    if (dst_internal_boundary_by_evaluation()) launch_air<2, 1, 4, 8, 3, true, false>(c, a, Q);      // boundary constraints + op bits
    else launch_air<2, 1, 4, 8, 2, true, false>(c, a, Q);                                         // op bits (starts the partial sums)
    launch_air<2, 1, 4, 8, 4, false, false>(c, a, Q);     // sponge, loop image, context / loop stacks
    launch_air<2, 1, 4, 8, 88, false, true>(c, a, Q);     // stack: low-degree ops as nested sums (st_low_degree), PUSH, CMP, BEGIN / NOOP, RESCR + combination
}
