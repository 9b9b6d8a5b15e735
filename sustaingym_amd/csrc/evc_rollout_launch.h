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
    // Round 6: the grid is persistent (as many workgroups as stay resident) and a wavefront that has finished its quad's T
    // periods takes the next quad from this counter (device memory; the engine sets it to 4 x grid before the launch: the first
    // quads are assigned by position).  Wavefront times differ by 3x on congested days; with one quad per wavefront and four
    // wavefronts per workgroup a launch took 13 mean wavefront times where 8 rounds were resident (tools/rollout_stats.py).
    unsigned* quad_counter;
    // ... in the order of this list (nquads entries; NULL: by index): the engine sorts the quads by the sessions their
    // environments still have to serve (rollout_order_kernel), busiest first, so that the launch does not end on a long quad
    // that happened to be dispatched last.
    const unsigned* quad_order;
};

// One launch of one workgroup: order[0 .. nquads) = the quads sorted by descending load estimate (counting sort; key = sessions
// not yet arrived + EVs plugged in, summed over the quad's four environments, from the per-environment scalars).
void launch_rollout_order(const Params& P, unsigned* order, hipStream_t stream);

// Launches rollout_kernel<P.project, (P.G + 1) / 2, policy kind (0 greedy, 1 random, 2 replay)> on `stream`; with start / stop events
// the launch carries them (hipExtLaunchKernel: the dispatch's own begin / end timestamps).  waves = 2 | 3: the register
// budget of the projecting kernels (wavefronts per SIMD; rollout_kernel's WAVES).  false: unsupported class count.
// site_alive: the copies with the site's shape compiled in and every environment inside its episode (rollout_kernel's NC / ALIVE;
// the caller has checked that Params describes that shape, that the batch is whole quads, autoreset on, clocks in range).
bool launch_rollout_kernel(const Params& P, const RolloutIO& io, int grid, hipStream_t stream, hipEvent_t start, hipEvent_t stop,
                           int waves = 3, bool site_alive = false);

}  // namespace evc
