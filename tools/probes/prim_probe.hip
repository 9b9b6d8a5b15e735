// tools/probes/prim_probe.hip — cycle cost of the slow path's primitives for ONE lone wavefront (what a queued
// environment's solve is made of): float64 DPP wave sums (1 / 5 interleaved), float32 fused-DPP wave sums, IEEE float64
// divide / sqrt chains, v_rcp_f64 + Newton.  hipcc --offload-arch=gfx950 -O3 -I sustaingym_amd/csrc -I include
#include <hip/hip_runtime.h>
#include <cstdio>
#include "evc_quad.h"
using namespace evc;

__device__ __forceinline__ float wave_sum_f32_dpp(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, false));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__global__ void probe(double* out, long long* cyc, int reps) {
    const int lane = threadIdx.x;
    double x = 1.0 + lane * 0.01, acc = 0.0;
    long long t0, t1;
    // 1: one f64 wave sum, dependent chain
    t0 = clock64();
    for (int i = 0; i < reps; i++) { x = wave_sum_f64(x * 1e-2 + lane); }
    t1 = clock64(); cyc[0] = t1 - t0; acc += x;
    // 2: five independent f64 wave sums per round (as wave_cone<1>)
    double a = x, b = x + 1, c = x + 2, d = x + 3, e = x + 4;
    t0 = clock64();
    for (int i = 0; i < reps; i++) {
        const double s0 = wave_sum_f64(a * 1e-2 + lane), s1 = wave_sum_f64(b * 1e-2 + lane), s2 = wave_sum_f64(c * 1e-2 + lane),
                     s3 = wave_sum_f64(d * 1e-2 + lane), s4 = wave_sum_f64(e * 1e-2 + lane);
        a = s1; b = s2; c = s3; d = s4; e = s0;
    }
    t1 = clock64(); cyc[1] = t1 - t0; acc += a + b + c + d + e;
    // 3: f32 fused-dpp wave sum x5
    float fa = (float)x, fb = fa + 1, fc = fa + 2, fd = fa + 3, fe = fa + 4;
    t0 = clock64();
    for (int i = 0; i < reps; i++) {
        const float s0 = wave_sum_f32_dpp(fa * 1e-2f + lane), s1 = wave_sum_f32_dpp(fb * 1e-2f + lane), s2 = wave_sum_f32_dpp(fc * 1e-2f + lane),
                    s3 = wave_sum_f32_dpp(fd * 1e-2f + lane), s4 = wave_sum_f32_dpp(fe * 1e-2f + lane);
        fa = s1; fb = s2; fc = s3; fd = s4; fe = s0;
    }
    t1 = clock64(); cyc[2] = t1 - t0; acc += fa + fb + fc + fd + fe;
    // 4: dependent f64 divides
    double q = 1.2345 + lane;
    t0 = clock64();
    for (int i = 0; i < reps; i++) q = 1.0 + 3.7 / q;
    t1 = clock64(); cyc[3] = t1 - t0; acc += q;
    // 5: dependent f64 sqrt
    t0 = clock64();
    for (int i = 0; i < reps; i++) q = sqrt(q + 2.0);
    t1 = clock64(); cyc[4] = t1 - t0; acc += q;
    // 6: dependent f64 fma
    t0 = clock64();
    for (int i = 0; i < reps; i++) q = __builtin_fma(q, 0.999, 0.5);
    t1 = clock64(); cyc[5] = t1 - t0; acc += q;
    // 7: rcp_f64 + 2 Newton steps
    t0 = clock64();
    for (int i = 0; i < reps; i++) { double r = __builtin_amdgcn_rcp(q); r = r * (2.0 - q * r); r = r * (2.0 - q * r); q = 1.0 + 3.7 * r; }
    t1 = clock64(); cyc[6] = t1 - t0; acc += q;
    // 8: row (16-lane) f64 allreduce x5 (4 steps)
    t0 = clock64();
    for (int i = 0; i < reps; i++) {
        const double s0 = row_allreduce_f64(a * 1e-2 + lane), s1 = row_allreduce_f64(b * 1e-2 + lane), s2 = row_allreduce_f64(c * 1e-2 + lane),
                     s3 = row_allreduce_f64(d * 1e-2 + lane), s4 = row_allreduce_f64(e * 1e-2 + lane);
        a = s1; b = s2; c = s3; d = s4; e = s0;
    }
    t1 = clock64(); cyc[7] = t1 - t0; acc += a + b + c + d + e;
    // 9: dependent f32 rcp / rsq
    float fq = 1.5f + lane;
    t0 = clock64();
    for (int i = 0; i < reps; i++) fq = 1.0f + 3.7f * __builtin_amdgcn_rcpf(fq);
    t1 = clock64(); cyc[8] = t1 - t0; acc += fq;
    // 10: 9 independent f64 wave sums (JPL class sums)
    double v[9];
    for (int j = 0; j < 9; j++) v[j] = x + j;
    t0 = clock64();
    for (int i = 0; i < reps; i++) {
        double s[9];
#pragma unroll
        for (int j = 0; j < 9; j++) s[j] = wave_sum_f64(v[j] * 1e-2 + lane);
#pragma unroll
        for (int j = 0; j < 9; j++) v[j] = s[(j + 1) % 9];
    }
    t1 = clock64(); cyc[9] = t1 - t0;
    for (int j = 0; j < 9; j++) acc += v[j];
    out[lane] = acc;
}

int main() {
    double* out; long long* cyc;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 16 * 8);
    const int reps = 200;
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, cyc, reps);
    long long h[16];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"f64 wave sum, dependent", "5 f64 wave sums, independent", "5 f32 fused-dpp wave sums", "f64 divide, dependent",
                           "f64 sqrt, dependent", "f64 fma, dependent", "rcp_f64 + 2 Newton, dependent", "5 row(16) f64 allreduces",
                           "f32 rcp, dependent", "9 f64 wave sums, independent"};
    for (int i = 0; i < 10; i++) printf("%-34s %8.1f clock64 ticks per round\n", names[i], (double)h[i] / reps);
    return 0;
}
