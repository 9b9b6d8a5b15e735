// evc_rollout_launch.h — host-side entry of the fused rollout kernels (evc_rollout.h), which are compiled in their own
// translation unit (evc_rollout.hip) so that the library builds in parallel.
#pragma once

#include <hip/hip_runtime.h>

#include "evc_device.h"

namespace evc {

struct RolloutIO {
    int policy;                  // EVC_ACTION_GREEDY / EVC_ACTION_RANDOM, or EVC_ACTION_F32 / EVC_ACTION_DISCRETE = replay of `actions`
    int bins;                    // RANDOM: >= 2 draws DiscreteActionWrapper levels; DISCRETE: the wrapper's bins
    const void* actions;         // replay: [ring_len][N][n] float32 (F32) / int64 levels (DISCRETE); period i reads slice i mod ring_len
    int ring_len;
    int steps;                   // T
    unsigned env_id_base;        // global id of environment 0 (random stream)
    unsigned long long seed;     // random stream key
    evc_step_out out;
};

// Launches rollout_kernel<P.project, (P.G + 1) / 2, policy kind (0 greedy, 1 random, 2 replay)> on `stream`; with start / stop events
// the launch carries them (hipExtLaunchKernel: the dispatch's own begin / end timestamps).  waves = 2 | 3: the register
// budget of the projecting kernels (wavefronts per SIMD; rollout_kernel's WAVES).  false: unsupported class count.
bool launch_rollout_kernel(const Params& P, const RolloutIO& io, int grid, hipStream_t stream, hipEvent_t start, hipEvent_t stop,
                           int waves = 3);

}  // namespace evc
