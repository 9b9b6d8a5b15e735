// evc_rollout_launch.h — host-side entry of the fused rollout kernels (evc_rollout.h), which are compiled in their own
// translation unit (evc_rollout.hip) so that the library builds in parallel.
#pragma once

#include <hip/hip_runtime.h>

#include "evc_device.h"

namespace evc {

struct RolloutIO {
    int policy;                  // EVC_ACTION_GREEDY / EVC_ACTION_RANDOM
    int bins;                    // RANDOM: >= 2 draws DiscreteActionWrapper levels
    int steps;                   // T
    unsigned env_id_base;        // global id of environment 0 (random stream)
    unsigned long long seed;     // random stream key
    evc_step_out out;
};

// Launches rollout_kernel<P.project, (P.G + 1) / 2, io.policy == EVC_ACTION_RANDOM> on `stream`; with start / stop events
// the launch carries them (hipExtLaunchKernel: the dispatch's own begin / end timestamps).  false: unsupported class count.
bool launch_rollout_kernel(const Params& P, const RolloutIO& io, int grid, hipStream_t stream, hipEvent_t start, hipEvent_t stop);

}  // namespace evc
