// AIR kernel instances for traces with <= 2 context registers, <= 1 loop register and a user stack of depth <= 8 (depth known
// at run time); split into section launches like the depth-4 instances (air_kernel.h).
#include "air_kernel.h"
void air_launch_small(dst_ctx* c, const AirArgs& a, uint32_t Q) {
    if (dst_internal_boundary_by_evaluation()) launch_air<2, 1, 0, 8, 3, true, false>(c, a, Q);      // boundary constraints + op bits
    else launch_air<2, 1, 0, 8, 2, true, false>(c, a, Q);                                         // op bits (starts the partial sums)
    launch_air<2, 1, 0, 8, 4, false, false>(c, a, Q);     // sponge, loop image, context / loop stacks
    launch_air<2, 1, 0, 8, 8, false, false>(c, a, Q);     // stack: low-degree ops as nested sums over all 8 slots (st_low_degree)
    launch_air<2, 1, 0, 8, 80, false, true>(c, a, Q);     // stack: PUSH, CMP, BEGIN / NOOP, RESCR + combination
}
