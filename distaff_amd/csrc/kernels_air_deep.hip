// AIR kernel instances for any trace shape the VM can produce (up to 16 context, 8 loop and 32 stack registers, known at run time)
// in the nested-sum formulation: the first eight stack slots as in the depth <= 8 instances (from twelve register-resident items of
// the current row), deeper slots from memory as flag sums times shifted differences (air_kernel.h, DEEP).  The per-operation
// formulation of kernels_air.hip stays as an independent statement of the same constraints (DISTAFF_AIR=generic).
#include "air_kernel.h"
void air_launch_deep(dst_ctx* c, const AirArgs& a, uint32_t Q) {
    if (dst_internal_boundary_by_evaluation()) launch_air<16, 8, 0, 12, 3, true, false>(c, a, Q);     // boundary constraints + op bits
    else launch_air<16, 8, 0, 12, 2, true, false>(c, a, Q);                                        // op bits (starts the partial sums)
    launch_air<16, 8, 0, 12, 4, false, false>(c, a, Q);    // sponge, loop image, context / loop stacks
    launch_air<16, 8, 0, 12, 8, false, false>(c, a, Q);    // stack slots 0..7: low-degree ops as nested sums (st_low_degree)
    launch_air<16, 8, 0, 12, 80, false, true>(c, a, Q);    // stack slots 0..7: PUSH, CMP, BEGIN / NOOP, RESCR; slots 8..: flag sums; combination
}
