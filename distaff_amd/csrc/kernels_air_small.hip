// AIR kernel instances for traces with <= 2 context registers, <= 1 loop register and a user stack of depth <= 8 (depth known
// at run time); split into section launches like the depth-4 instances (air_kernel.h).
#include "air_kernel.h"
void air_launch_small(dst_ctx* c, const AirArgs& a, uint32_t Q) {
    launch_air<2, 1, 0, 8, 7, true, false>(c, a, Q);      // boundary + op bits + sponge / context / loop
    launch_air<2, 1, 0, 8, 120, false, true>(c, a, Q);     // stack + combination
}
