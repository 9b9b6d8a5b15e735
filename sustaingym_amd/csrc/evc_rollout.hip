// evc_rollout.hip — translation unit of the fused rollout kernels (evc_rollout.h).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#define EVC_TEMPLATES_ONLY
#include "evc_rollout.h"

namespace evc {

constexpr int kOrderBins = 1024;
__global__ __launch_bounds__(1024) void rollout_order_kernel(Params P, unsigned* __restrict__ order) {
    __shared__ unsigned hist[kOrderBins];
    const unsigned tid = threadIdx.x, N = (unsigned)P.N, nquads = (N + 3u) >> 2;
    hist[tid] = 0u;
    __syncthreads();
    auto key_of = [&](unsigned quad) {
        unsigned key = 0u;
        for (unsigned r = 0; r < 4u; r++) {
            const unsigned env = quad * 4u + r;
            if (env < N) {
                const int4 s0 = P.scal[2 * env], s1 = P.scal[2 * env + 1];
                const int left = s0.x >= EVC_EPISODE_STEPS ? 0 : (s1.x - s0.y);          // sessions behind the cursor (none once the episode is over)
                key += (unsigned)(left > 0 ? left : 0) + (((unsigned)s1.z >> kCountShift) & 0x7fu);
            }
        }
        return key < (unsigned)kOrderBins ? key : (unsigned)kOrderBins - 1u;
    };
    for (unsigned quad = tid; quad < nquads; quad += 1024u) atomicAdd(&hist[key_of(quad)], 1u);
    __syncthreads();
    if (tid == 0u) {                                   // exclusive prefix, busiest bin first (1 024 bins: a serial pass is a microsecond)
        unsigned at = 0u;
        for (int b = kOrderBins - 1; b >= 0; b--) { const unsigned c = hist[b]; hist[b] = at; at += c; }
    }
    __syncthreads();
    for (unsigned quad = tid; quad < nquads; quad += 1024u) order[atomicAdd(&hist[key_of(quad)], 1u)] = quad;
}

void launch_rollout_order(const Params& P, unsigned* order, hipStream_t stream) {
    hipLaunchKernelGGL(rollout_order_kernel, dim3(1), dim3(1024), 0, stream, P, order);
}

bool launch_rollout_kernel(const Params& P, const RolloutIO& io, int grid, hipStream_t stream, hipEvent_t start, hipEvent_t stop,
                           int waves, bool site_alive) {
    const int words = (P.G + 1) / 2;
    const int kind = io.policy == EVC_ACTION_GREEDY ? 0 : (io.policy == EVC_ACTION_RANDOM ? 1 : 2);
    auto launch = [&](auto kernel) {
        if (start && stop) hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, stream, start, stop, 0, P, io);
        else hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, stream, P, io);
    };
#define EVC_ROLL_K(PROJ, W, KD, WV)                                                                 \
        launch((site_alive && SiteStations<W>::value != 0 && P.n == SiteStations<W>::value)         \
                   ? rollout_kernel<PROJ, W, KD, WV, SiteStations<W>::value, SiteStations<W>::value != 0> \
                   : rollout_kernel<PROJ, W, KD, WV, 0, false>)
#define EVC_ROLL_P(PROJ, W, WV)                                                                     \
        if (kind == 0) EVC_ROLL_K(PROJ, W, 0, WV);                                                  \
        else if (kind == 1) EVC_ROLL_K(PROJ, W, 1, WV);                                             \
        else EVC_ROLL_K(PROJ, W, 2, WV);
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
#undef EVC_ROLL_K
#undef EVC_ROLL
}

}  // namespace evc
