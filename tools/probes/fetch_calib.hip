// Standalone probe (not product): what does the FETCH_SIZE counter report for the READ access shapes of the compact
// streaming kernel?  MI355X_MICROARCH.md documents "x2" for 16 B/lane streaming reads on gfx950 and calls narrower
// shapes uncalibrated.  Each kernel reads a KNOWN number of bytes exactly once from a 1 GiB buffer (nothing cached);
// run under `rocprofv3 --pmc FETCH_SIZE` (tools/scratch/fetch_calib.sh) and divide.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void read_x4_linear(const uint4* __restrict__ src, unsigned* sink, size_t n16) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { uint4 v = src[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345u) *sink = acc;
}
// rows of `row_dwords` dwords per environment, 16 lanes per environment, lane q reads dwords q, q+16, ... (action rows: 54)
__global__ __launch_bounds__(256) void read_rows_dword(const unsigned* __restrict__ src, unsigned* sink, unsigned N, unsigned row_dwords) {
    const unsigned lane = threadIdx.x & 63u, q = lane & 15u, row = lane >> 4;
    unsigned acc = 0;
    for (unsigned quad = blockIdx.x * 4u + (threadIdx.x >> 6); quad < N / 4; quad += gridDim.x * 4u) {
        const unsigned env = quad * 4 + row;
        for (unsigned j = q; j < row_dwords; j += 16) acc += src[(size_t)env * row_dwords + j];
    }
    if (acc == 0x12345u) *sink = acc;
}
// first 16 entries of a row of `row_qwords` 8-byte values per environment (remaining demand of the compact layout: stride 54)
__global__ __launch_bounds__(256) void read_rows_x2_head(const uint2* __restrict__ src, unsigned* sink, unsigned N, unsigned row_qwords) {
    const unsigned lane = threadIdx.x & 63u, q = lane & 15u, row = lane >> 4;
    unsigned acc = 0;
    for (unsigned quad = blockIdx.x * 4u + (threadIdx.x >> 6); quad < N / 4; quad += gridDim.x * 4u) {
        const unsigned env = quad * 4 + row;
        uint2 v = src[(size_t)env * row_qwords + q];
        acc += v.x ^ v.y;
    }
    if (acc == 0x12345u) *sink = acc;
}
// first 16 dwords of a row (entry words of the compact layout: stride 54 dwords)
__global__ __launch_bounds__(256) void read_rows_dword_head(const unsigned* __restrict__ src, unsigned* sink, unsigned N, unsigned row_dwords) {
    const unsigned lane = threadIdx.x & 63u, q = lane & 15u, row = lane >> 4;
    unsigned acc = 0;
    for (unsigned quad = blockIdx.x * 4u + (threadIdx.x >> 6); quad < N / 4; quad += gridDim.x * 4u) {
        const unsigned env = quad * 4 + row;
        acc += src[(size_t)env * row_dwords + q];
    }
    if (acc == 0x12345u) *sink = acc;
}
// 32 bytes per environment, every lane of the row loads the same two 16-byte words (environment scalars)
__global__ __launch_bounds__(256) void read_scalars_x4(const uint4* __restrict__ src, unsigned* sink, unsigned N) {
    const unsigned lane = threadIdx.x & 63u, row = lane >> 4;
    unsigned acc = 0;
    for (unsigned quad = blockIdx.x * 4u + (threadIdx.x >> 6); quad < N / 4; quad += gridDim.x * 4u) {
        const unsigned env = quad * 4 + row;
        uint4 a = src[(size_t)env * 2], b = src[(size_t)env * 2 + 1];
        acc += a.x ^ b.w;
    }
    if (acc == 0x12345u) *sink = acc;
}

int main() {
    const size_t bytes = 1ull << 30;
    void* buf; unsigned* sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 1, bytes));
    CK(hipDeviceSynchronize());
    const unsigned N = 1u << 20;          // "environments"
    // expected bytes read, printed so that the shell script can divide
    printf("expected read_x4_linear %zu\n", bytes);
    printf("expected read_rows_dword %zu\n", (size_t)N * 54 * 4);
    printf("expected read_rows_x2_head %zu\n", (size_t)N * 16 * 8);
    printf("expected read_rows_dword_head %zu\n", (size_t)N * 16 * 4);
    printf("expected read_scalars_x4 %zu\n", (size_t)N * 32);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(read_x4_linear, dim3(4096), dim3(256), 0, 0, (const uint4*)buf, sink, bytes / 16);
        CK(hipMemset(sink, 0, 4));        // separates the dispatches; the 1 GiB pass above evicted everything smaller
        hipLaunchKernelGGL(read_rows_dword, dim3(4096), dim3(256), 0, 0, (const unsigned*)buf, sink, N, 54u);
        hipLaunchKernelGGL(read_x4_linear, dim3(4096), dim3(256), 0, 0, (const uint4*)buf + (bytes / 32), sink, bytes / 32);
        hipLaunchKernelGGL(read_rows_x2_head, dim3(4096), dim3(256), 0, 0, (const uint2*)buf, sink, N, 54u);
        hipLaunchKernelGGL(read_x4_linear, dim3(4096), dim3(256), 0, 0, (const uint4*)buf + (bytes / 32), sink, bytes / 32);
        hipLaunchKernelGGL(read_rows_dword_head, dim3(4096), dim3(256), 0, 0, (const unsigned*)buf, sink, N, 54u);
        hipLaunchKernelGGL(read_x4_linear, dim3(4096), dim3(256), 0, 0, (const uint4*)buf + (bytes / 32), sink, bytes / 32);
        hipLaunchKernelGGL(read_scalars_x4, dim3(4096), dim3(256), 0, 0, (const uint4*)buf, sink, N);
        hipLaunchKernelGGL(read_x4_linear, dim3(4096), dim3(256), 0, 0, (const uint4*)buf + (bytes / 32), sink, bytes / 32);
    }
    CK(hipDeviceSynchronize());
    return 0;
}
