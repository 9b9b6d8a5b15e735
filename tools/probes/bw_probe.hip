// Standalone probe (not product): what does the MI355X memory system deliver for the byte mix of
// one step (60.6 MB read, 90.7 MB written per 65536 envs) under different access shapes?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// V0: linear float4 copy: reads R bytes, writes W bytes (independent regions), grid-stride
__global__ __launch_bounds__(256) void linear_rw(const float4* __restrict__ src, float4* __restrict__ dst, size_t nr, size_t nw) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float4 acc = make_float4(0, 0, 0, 0);
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t j = i; j < nr; j += stride) { float4 v = src[j]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    for (size_t j = i; j < nw; j += stride) dst[j] = acc;
}
// V1: interleaved: each thread alternates 2 loads / 3 stores per iteration (same mix, interleaved in time)
__global__ __launch_bounds__(256) void interleaved_rw(const float4* __restrict__ src, float4* __restrict__ dst, size_t nr, size_t nw) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t jw = i;
    for (size_t j = i; j < nr; j += 2 * stride) {
        float4 a = src[j];
        float4 b = (j + stride < nr) ? src[j + stride] : a;
        a.x += b.x;
        for (int k = 0; k < 3; k++) { if (jw < nw) dst[jw] = a; jw += stride; }
    }
}
// V2: env-row pattern: one 16-lane row per env; reads rows of RB bytes, writes rows of WB bytes, dword granularity
template <int RD, int WD>   // dwords per env read / written
__global__ __launch_bounds__(256, 4) void row_rw(const unsigned* __restrict__ src, unsigned* __restrict__ dst, unsigned N) {
    const unsigned lane = threadIdx.x & 63u, q = lane & 15u, row = lane >> 4;
    const unsigned wave = (blockIdx.x * 4u + (threadIdx.x >> 6)), nw = gridDim.x * 4u;
    for (unsigned quad = wave; quad < (N + 3) / 4; quad += nw) {
        const unsigned env = quad * 4 + row;
        if (env >= N) continue;
        unsigned acc = 0;
#pragma unroll
        for (int j = 0; j < (RD + 15) / 16; j++) { unsigned idx = j * 16 + q; if (idx < RD) acc += src[(size_t)env * RD + idx]; }
#pragma unroll
        for (int j = 0; j < (WD + 15) / 16; j++) { unsigned idx = j * 16 + q; if (idx < WD) dst[(size_t)env * WD + idx] = acc + j; }
    }
}
// V3: same bytes, wave-contiguous tiles: each wave reads 4*RD dwords contiguous, writes 4*WD dwords contiguous with dwordx4
template <int RD, int WD>
__global__ __launch_bounds__(256, 4) void tile_rw(const uint4* __restrict__ src, uint4* __restrict__ dst, unsigned N) {
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = (blockIdx.x * 4u + (threadIdx.x >> 6)), nw = gridDim.x * 4u;
    for (unsigned quad = wave; quad < (N + 3) / 4; quad += nw) {
        uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < (RD + 63) / 64; j++) { unsigned idx = j * 64 + lane; if (idx < RD) { uint4 v = src[(size_t)quad * RD + idx]; acc.x += v.x; acc.y ^= v.y; acc.z += v.z; acc.w ^= v.w; } }
#pragma unroll
        for (int j = 0; j < (WD + 63) / 64; j++) { unsigned idx = j * 64 + lane; if (idx < WD) dst[(size_t)quad * WD + idx] = acc; }
    }
}

// V4: the step kernel's exact arrays and access shapes (no arithmetic): per env row (16 lanes):
// reads rem f64[54], depest u32[54], action f32[54], scal 32 B, acc 24 B; writes rem, depest, scal, acc,
// obs f32[146], reward f64, terminated u8, breakdown f64[3].
struct Arrays { double* rem2; unsigned* de2; double* rem; unsigned* de; const float* act; uint4* scal; double* acc; float* obs; double* rew; unsigned char* term; double* bd; };
template <bool XCD, bool INPLACE, bool SMALL, bool OBS>
__global__ __launch_bounds__(256, 4) void step_shape(Arrays A, unsigned N) {
    const unsigned lane = threadIdx.x & 63u, q = lane & 15u, row = lane >> 4;
    const unsigned n = 54, F = 146;
    unsigned first, stride, hi;
    const unsigned nquads = (N + 3) / 4, wv = threadIdx.x >> 6;
    if (XCD) { const unsigned xcd = blockIdx.x & 7, bx = blockIdx.x >> 3, nbx = gridDim.x >> 3; const unsigned lo = nquads * xcd / 8; hi = nquads * (xcd + 1) / 8; first = lo + bx * 4 + wv; stride = nbx * 4; }
    else { hi = nquads; first = blockIdx.x * 4 + wv; stride = gridDim.x * 4; }
    for (unsigned quad = first; quad < hi; quad += stride) {
        const unsigned env = quad * 4 + row;
        if (env >= N) continue;
        uint4 s0 = make_uint4(0,0,0,0), s1 = s0; if (SMALL) { s0 = A.scal[env * 2]; s1 = A.scal[env * 2 + 1]; }
        double rem[4]; unsigned de[4]; float a[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { const unsigned s = j * 16 + q; const bool v = s < n; rem[j] = v ? A.rem[(size_t)env * n + s] : 0.0; de[j] = v ? A.de[(size_t)env * n + s] : 0u; a[j] = v ? A.act[(size_t)env * n + s] : 0.f; }
        double acc = (SMALL && q < 3) ? A.acc[(size_t)env * 3 + q] : 0.0;
        double* remo = INPLACE ? A.rem : A.rem2; unsigned* deo = INPLACE ? A.de : A.de2;
#pragma unroll
        for (int j = 0; j < 4; j++) { const unsigned s = j * 16 + q; if (s < n) { remo[(size_t)env * n + s] = rem[j] + a[j]; deo[(size_t)env * n + s] = de[j] + 1; } }
        if (OBS)
#pragma unroll
        for (int j = 0; j < 10; j++) { const unsigned idx = j * 16 + q; if (idx < F) A.obs[(size_t)env * F + idx] = (float)rem[j & 3] + a[j & 3]; }
        if (SMALL) {
        s0.x += 1; A.scal[env * 2] = s0; A.scal[env * 2 + 1] = s1;
        if (q < 3) { A.acc[(size_t)env * 3 + q] = acc + 1.0; A.bd[(size_t)env * 3 + q] = acc; }
        if (q == 0) { A.rew[env] = acc; A.term[env] = (unsigned char)(s0.x == 288); }
        }
    }
}

// V5: step shape + dependent arithmetic between the loads and the stores (WORK f64 fmas per slot) and an
// optional register prefetch of the next quad's rows: how much does load/compute overlap buy?
template <int WORK, bool PREFETCH, int WAVES>
__global__ __launch_bounds__(256, WAVES) void step_work(Arrays A, unsigned N) {
    const unsigned lane = threadIdx.x & 63u, q = lane & 15u, row = lane >> 4;
    const unsigned n = 54, F = 146;
    const unsigned nquads = (N + 3) / 4, wv = threadIdx.x >> 6;
    const unsigned xcd = blockIdx.x & 7, bx = blockIdx.x >> 3, nbx = gridDim.x >> 3;
    const unsigned lo = nquads * xcd / 8, hi = nquads * (xcd + 1) / 8, first = lo + bx * 4 + wv, stride = nbx * 4;
    struct Ld { uint4 s0, s1; double rem[4]; unsigned de[4]; float a[4]; double acc; };
    auto issue = [&](unsigned quad) {
        Ld L;
        const unsigned env = quad * 4 + row;
        const bool ok = quad < hi && env < N;
        L.s0 = ok ? A.scal[env * 2] : make_uint4(0, 0, 0, 0); L.s1 = ok ? A.scal[env * 2 + 1] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; j++) { const unsigned s = j * 16 + q; const bool v = ok && s < n; L.rem[j] = v ? A.rem[(size_t)env * n + s] : 0.0; L.de[j] = v ? A.de[(size_t)env * n + s] : 0u; L.a[j] = v ? A.act[(size_t)env * n + s] : 0.f; }
        L.acc = (ok && q < 3) ? A.acc[(size_t)env * 3 + q] : 0.0;
        return L;
    };
    Ld nxt;
    if (PREFETCH) nxt = issue(first);
    for (unsigned quad = first; quad < hi; quad += stride) {
        const unsigned env = quad * 4 + row;
        Ld cur;
        if (PREFETCH) { cur = nxt; nxt = issue(quad + stride); } else cur = issue(quad);
        if (env >= N) continue;
        // dependent MOER-like second round trip
        const float m0 = A.act[(size_t)(cur.s0.w % 64u) * 37 + (cur.s0.x % 288u) + q];
        double x[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { x[j] = cur.rem[j] + cur.a[j];
#pragma unroll
            for (int w = 0; w < WORK; w++) x[j] = fma(x[j], 1.0000001, 1e-9); }
#pragma unroll
        for (int j = 0; j < 4; j++) { const unsigned s = j * 16 + q; if (s < n) { A.rem[(size_t)env * n + s] = x[j]; A.de[(size_t)env * n + s] = cur.de[j] + 1; } }
#pragma unroll
        for (int j = 0; j < 10; j++) { const unsigned idx = j * 16 + q; if (idx < F) A.obs[(size_t)env * F + idx] = (float)x[j & 3] + m0; }
        cur.s0.x += 1; A.scal[env * 2] = cur.s0; A.scal[env * 2 + 1] = cur.s1;
        if (q < 3) { A.acc[(size_t)env * 3 + q] = cur.acc + 1.0; A.bd[(size_t)env * 3 + q] = cur.acc; }
        if (q == 0) { A.rew[env] = cur.acc; A.term[env] = (unsigned char)(cur.s0.x == 288); }
    }
}

int main() {
    const unsigned N = 65536;
    const int RD = 231, WD = 346;           // dwords per env: 924 B read, 1384 B written (= 2308 B)
    const size_t rbytes = (size_t)N * RD * 4, wbytes = (size_t)N * WD * 4;
    void *src, *dst;
    CK(hipMalloc(&src, rbytes + 4096)); CK(hipMalloc(&dst, wbytes + 4096));
    CK(hipMemset(src, 1, rbytes)); CK(hipMemset(dst, 0, wbytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch) {
        for (int i = 0; i < 5; i++) launch();
        float best = 1e9, sum = 0;
        for (int rep = 0; rep < 20; rep++) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; sum += ms;
        }
        printf("%-34s best %.2f us (%.2f TB/s)  avg %.2f us\n", name, best * 1e3, (rbytes + wbytes) / (best * 1e-3) / 1e12, sum / 20 * 1e3);
        return 0;
    };
    for (int grid : {1024, 2048, 4096, 8192}) {
        char nm[64];
        snprintf(nm, 64, "linear float4 r then w, grid %d", grid);
        timeit(nm, [&] { hipLaunchKernelGGL(linear_rw, dim3(grid), dim3(256), 0, 0, (const float4*)src, (float4*)dst, rbytes / 16, wbytes / 16); });
        snprintf(nm, 64, "interleaved float4, grid %d", grid);
        timeit(nm, [&] { hipLaunchKernelGGL(interleaved_rw, dim3(grid), dim3(256), 0, 0, (const float4*)src, (float4*)dst, rbytes / 16, wbytes / 16); });
    }
    for (int grid : {1024, 2048, 4096}) {
        char nm[64];
        snprintf(nm, 64, "env rows dword, grid %d", grid);
        timeit(nm, [&] { hipLaunchKernelGGL((row_rw<RD, WD>), dim3(grid), dim3(256), 0, 0, (const unsigned*)src, (unsigned*)dst, N); });
        snprintf(nm, 64, "wave tiles dwordx4, grid %d", grid);
        timeit(nm, [&] { hipLaunchKernelGGL((tile_rw<RD, WD>), dim3(grid), dim3(256), 0, 0, (const uint4*)src, (uint4*)dst, N); });
    }
    {
        Arrays A;
        CK(hipMalloc(&A.rem, (size_t)N * 54 * 8)); CK(hipMalloc(&A.de, (size_t)N * 54 * 4)); CK(hipMalloc((void**)&A.act, (size_t)N * 54 * 4));
        CK(hipMalloc(&A.scal, (size_t)N * 32)); CK(hipMalloc(&A.acc, (size_t)N * 24)); CK(hipMalloc(&A.obs, (size_t)N * 146 * 4));
        CK(hipMalloc(&A.rew, (size_t)N * 8)); CK(hipMalloc(&A.term, N)); CK(hipMalloc(&A.bd, (size_t)N * 24));
        CK(hipMemset(A.rem, 0, (size_t)N * 54 * 8)); CK(hipMemset(A.de, 0, (size_t)N * 54 * 4)); CK(hipMemset((void*)A.act, 0, (size_t)N * 54 * 4));
        CK(hipMemset(A.scal, 0, (size_t)N * 32)); CK(hipMemset(A.acc, 0, (size_t)N * 24));
        CK(hipMalloc(&A.rem2, (size_t)N * 54 * 8)); CK(hipMalloc(&A.de2, (size_t)N * 54 * 4));
        const int grid = 1024;
        timeit("step shape inplace small obs", [&] { hipLaunchKernelGGL((step_shape<true, true, true, true>), dim3(grid), dim3(256), 0, 0, A, N); });
        timeit("step shape pingpong small obs", [&] { hipLaunchKernelGGL((step_shape<true, false, true, true>), dim3(grid), dim3(256), 0, 0, A, N); });
        timeit("step shape inplace nosmall obs", [&] { hipLaunchKernelGGL((step_shape<true, true, false, true>), dim3(grid), dim3(256), 0, 0, A, N); });
        timeit("step shape pingpong nosmall obs", [&] { hipLaunchKernelGGL((step_shape<true, false, false, true>), dim3(grid), dim3(256), 0, 0, A, N); });
        timeit("step shape inplace small noobs", [&] { hipLaunchKernelGGL((step_shape<true, true, true, false>), dim3(grid), dim3(256), 0, 0, A, N); });
        timeit("step shape inplace nosmall noobs", [&] { hipLaunchKernelGGL((step_shape<true, true, false, false>), dim3(grid), dim3(256), 0, 0, A, N); });
        timeit("step shape pingpong nosmall noobs", [&] { hipLaunchKernelGGL((step_shape<true, false, false, false>), dim3(grid), dim3(256), 0, 0, A, N); });
    }
    {
        Arrays A;
        CK(hipMalloc(&A.rem, (size_t)N * 54 * 8)); CK(hipMalloc(&A.de, (size_t)N * 54 * 4)); CK(hipMalloc((void**)&A.act, (size_t)N * 54 * 4));
        CK(hipMalloc(&A.scal, (size_t)N * 32)); CK(hipMalloc(&A.acc, (size_t)N * 24)); CK(hipMalloc(&A.obs, (size_t)N * 146 * 4));
        CK(hipMalloc(&A.rew, (size_t)N * 8)); CK(hipMalloc(&A.term, N)); CK(hipMalloc(&A.bd, (size_t)N * 24));
        CK(hipMemset(A.rem, 0, (size_t)N * 54 * 8)); CK(hipMemset(A.de, 0, (size_t)N * 54 * 4)); CK(hipMemset((void*)A.act, 0, (size_t)N * 54 * 4));
        CK(hipMemset(A.scal, 0, (size_t)N * 32)); CK(hipMemset(A.acc, 0, (size_t)N * 24));
        const int grid = 1024;
        timeit("work 0   no prefetch 4w", [&] { hipLaunchKernelGGL((step_work<0, false, 4>), dim3(grid), dim3(256), 0, 0, A, N); });
        timeit("work 0   prefetch    4w", [&] { hipLaunchKernelGGL((step_work<0, true, 4>), dim3(grid), dim3(256), 0, 0, A, N); });
        timeit("work 50  no prefetch 4w", [&] { hipLaunchKernelGGL((step_work<50, false, 4>), dim3(grid), dim3(256), 0, 0, A, N); });
        timeit("work 50  prefetch    4w", [&] { hipLaunchKernelGGL((step_work<50, true, 4>), dim3(grid), dim3(256), 0, 0, A, N); });
        timeit("work 100 no prefetch 4w", [&] { hipLaunchKernelGGL((step_work<100, false, 4>), dim3(grid), dim3(256), 0, 0, A, N); });
        timeit("work 100 prefetch    4w", [&] { hipLaunchKernelGGL((step_work<100, true, 4>), dim3(grid), dim3(256), 0, 0, A, N); });
        timeit("work 200 no prefetch 4w", [&] { hipLaunchKernelGGL((step_work<200, false, 4>), dim3(grid), dim3(256), 0, 0, A, N); });
        timeit("work 200 prefetch    4w", [&] { hipLaunchKernelGGL((step_work<200, true, 4>), dim3(grid), dim3(256), 0, 0, A, N); });
        timeit("work 200 no prefetch 8w grid2048", [&] { hipLaunchKernelGGL((step_work<200, false, 8>), dim3(2048), dim3(256), 0, 0, A, N); });
        timeit("work 200 prefetch 8w grid2048", [&] { hipLaunchKernelGGL((step_work<200, true, 8>), dim3(2048), dim3(256), 0, 0, A, N); });
        timeit("work 200 no prefetch 3w grid 768", [&] { hipLaunchKernelGGL((step_work<200, false, 3>), dim3(768), dim3(256), 0, 0, A, N); });
        timeit("work 200 prefetch 3w grid 768", [&] { hipLaunchKernelGGL((step_work<200, true, 3>), dim3(768), dim3(256), 0, 0, A, N); });
    }
    // copy of equal halves for reference (the guide's 6.29 TB/s float4 copy)
    timeit("float4 copy 75.6 MB -> 75.6 MB", [&] { hipLaunchKernelGGL(interleaved_rw, dim3(4096), dim3(256), 0, 0, (const float4*)src, (float4*)dst, rbytes / 16, rbytes / 16 / 2 * 3 > wbytes / 16 ? wbytes / 16 : rbytes / 16); });
    return 0;
}
