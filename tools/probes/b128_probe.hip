// Round 6: what the 16-byte station side of the streaming kernel (evc_cquad.h) relies on, checked on gfx950:
//  (a) buffer_load_dwordx4 / buffer_store_dwordx4 through a RAW buffer at offsets that are only 4- or 8-byte aligned
//      (action rows are 216 B, observation rows 584 B apart: 8-byte aligned, not 16);
//  (b) the range check of a multi-dword raw access is per COMPONENT: a load that straddles num_records returns its
//      in-range dwords and 0 for the rest, a store that straddles it writes only the in-range dwords.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
constexpr int kWords = 1024;
__global__ void probe(unsigned* src, unsigned* dst, unsigned* out) {
    rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(src, 0, kWords * 4, 0x00020000);
    rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, kWords * 4, 0x00020000);
    const unsigned lane = threadIdx.x;
    // (a) loads at byte offsets 4 + 20*lane (4-byte aligned, never 16) and 8 + 24*lane (8-byte aligned)
    v4u a = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(4u + 20u * lane), 0, 0);
    v4u b = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(8u + 24u * lane), 0, 0);
    out[lane * 16 + 0] = a.x; out[lane * 16 + 1] = a.y; out[lane * 16 + 2] = a.z; out[lane * 16 + 3] = a.w;
    out[lane * 16 + 4] = b.x; out[lane * 16 + 5] = b.y; out[lane * 16 + 6] = b.z; out[lane * 16 + 7] = b.w;
    // (b) a load that straddles the end: the last 2 dwords in range, 2 beyond
    v4u c = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((kWords - 2) * 4u), 0, 0);
    out[lane * 16 + 8] = c.x; out[lane * 16 + 9] = c.y; out[lane * 16 + 10] = c.z; out[lane * 16 + 11] = c.w;
    // stores: lane l writes 4 dwords at byte 8 + 16*l (8-byte aligned only) of dst[0 .. 258)
    v4u s = {0xA0000000u + lane * 4u, 0xA0000001u + lane * 4u, 0xA0000002u + lane * 4u, 0xA0000003u + lane * 4u};
    __builtin_amdgcn_raw_buffer_store_b128(s, rd, (int)(8u + 16u * lane), 0, 0);
    // a store that straddles the end of the buffer (lane 0 only): dwords kWords-2, kWords-1 in range
    v4u t = {0xB0u, 0xB1u, 0xB2u, 0xB3u};
    __builtin_amdgcn_raw_buffer_store_b128(t, rd, lane == 0u ? (int)((kWords - 2) * 4u) : (int)0xffffffffu, 0, 0);
    v2u u = {0xC0u, 0xC1u};
    __builtin_amdgcn_raw_buffer_store_b64(u, rd, lane == 0u ? (int)(600u * 4u + 4u) : (int)0xffffffffu, 0, 0);   // 4-byte aligned b64
}
int main() {
    unsigned *src, *dst, *out;
    hipMalloc(&src, kWords * 4 + 64); hipMalloc(&dst, kWords * 4 + 64); hipMalloc(&out, 64 * 16 * 4);
    unsigned h[kWords + 16];
    for (int i = 0; i < kWords + 16; i++) h[i] = 0x51000000u + i;
    hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice);
    hipMemset(dst, 0, kWords * 4 + 64); hipMemset(out, 0xff, 64 * 16 * 4);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, dst, out); hipDeviceSynchronize();
    unsigned o[64 * 16], d[kWords + 16];
    hipMemcpy(o, out, sizeof o, hipMemcpyDeviceToHost); hipMemcpy(d, dst, sizeof d, hipMemcpyDeviceToHost);
    int bad_a = 0, bad_b = 0, bad_s = 0;
    for (int l = 0; l < 64; l++)
        for (int j = 0; j < 4; j++) {
            const unsigned ia = 1 + 5 * l + j, ib = 2 + 6 * l + j;
            bad_a += o[l * 16 + j] != (ia < kWords ? 0x51000000u + ia : 0u);
            bad_b += o[l * 16 + 4 + j] != (ib < kWords ? 0x51000000u + ib : 0u);
        }
    printf("x4 loads at 4-byte-aligned offsets: %d wrong dwords; at 8-byte-aligned offsets: %d wrong (0 expected)\n", bad_a, bad_b);
    printf("x4 load straddling num_records: %#x %#x %#x %#x (expect %#x %#x 0 0: per-component range check)\n", o[8], o[9], o[10], o[11],
           0x51000000u + kWords - 2, 0x51000000u + kWords - 1);
    for (int i = 0; i < 256; i++) bad_s += d[2 + i] != 0xA0000000u + i;
    printf("x4 stores at 8-byte-aligned offsets: %d wrong dwords of 256 (0 expected); neighbours %#x %#x (0 0 expected)\n", bad_s, d[1], d[258]);
    printf("x4 store straddling num_records: in-range %#x %#x (expect 0xb0 0xb1), beyond the descriptor %#x %#x (expect 0 0)\n",
           d[kWords - 2], d[kWords - 1], d[kWords], d[kWords + 1]);
    printf("b64 store at a 4-byte-aligned offset: %#x %#x (expect 0xc0 0xc1)\n", d[601], d[602]);
    return 0;
}
