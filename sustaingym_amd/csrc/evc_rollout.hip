// evc_rollout.hip — translation unit of the fused rollout kernels (evc_rollout.h).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#define EVC_TEMPLATES_ONLY
#include "evc_rollout.h"

namespace evc {

bool launch_rollout_kernel(const Params& P, const RolloutIO& io, int grid, hipStream_t stream, hipEvent_t start, hipEvent_t stop,
                           int waves) {
    const int words = (P.G + 1) / 2;
    const int kind = io.policy == EVC_ACTION_GREEDY ? 0 : (io.policy == EVC_ACTION_RANDOM ? 1 : 2);
    auto launch = [&](auto kernel) {
        if (start && stop) hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, stream, start, stop, 0, P, io);
        else hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, stream, P, io);
    };
#define EVC_ROLL_P(PROJ, W, WV)                                                                     \
        if (kind == 0) launch(rollout_kernel<PROJ, W, 0, WV>);                                      \
        else if (kind == 1) launch(rollout_kernel<PROJ, W, 1, WV>);                                 \
        else launch(rollout_kernel<PROJ, W, 2, WV>);
#define EVC_ROLL(W)                                                                                 \
    case W:                                                                                         \
        if (P.project && waves == 2) { EVC_ROLL_P(true, W, 2) }                                     \
        else if (P.project) { EVC_ROLL_P(true, W, 3) }                                              \
        else { EVC_ROLL_P(false, W, 3) }                                                            \
        return true;
    switch (words) {
        EVC_ROLL(1) EVC_ROLL(2) EVC_ROLL(3) EVC_ROLL(4) EVC_ROLL(5) EVC_ROLL(6) EVC_ROLL(7) EVC_ROLL(8)
        default: return false;
    }
#undef EVC_ROLL_P
#undef EVC_ROLL
}

}  // namespace evc
