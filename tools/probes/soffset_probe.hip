// Does the gfx950 raw-buffer range check include the scalar offset?  Store/load with voffset = 0xffffffff
// (out of range by itself) and soffset = 64 on a 4 KiB buffer: if the check used voffset + soffset the
// access would wrap to byte 63 and land inside the buffer.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__global__ void probe(unsigned* buf, unsigned* out) {
    rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 4096, 0x00020000);
    const unsigned lane = threadIdx.x;
    // in-range reference: voffset = lane*4, soffset = 256 -> element 64 + lane
    __builtin_amdgcn_raw_buffer_store_b32(0xAAAA0000u + lane, r, (int)(lane * 4u), 256, 0);
    // out-of-range voffset with a scalar offset
    __builtin_amdgcn_raw_buffer_store_b32(0xBBBB0000u + lane, r, (int)0xffffffffu, 64, 0);
    __builtin_amdgcn_raw_buffer_store_b32(0xCCCC0000u + lane, r, (int)(0xfffffffcu - lane * 4u), 1024, 0);
    // voffset in range but voffset + soffset beyond num_records
    __builtin_amdgcn_raw_buffer_store_b32(0xDDDD0000u + lane, r, (int)(lane * 4u), 4096, 0);
    out[lane] = __builtin_amdgcn_raw_buffer_load_b32(r, (int)0xffffffffu, 64, 0);
}
int main() {
    unsigned *buf, *out; hipMalloc(&buf, 16384); hipMalloc(&out, 256); hipMemset(buf, 0, 16384); hipMemset(out, 0xff, 256);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, buf, out); hipDeviceSynchronize();
    unsigned h[4096], o[64]; hipMemcpy(h, buf, 16384, hipMemcpyDeviceToHost); hipMemcpy(o, out, 256, hipMemcpyDeviceToHost);
    int a = 0, b = 0, c = 0, d = 0;
    for (int i = 0; i < 4096; i++) { unsigned t = h[i] >> 16; a += t == 0xAAAA; b += t == 0xBBBB; c += t == 0xCCCC; d += t == 0xDDDD; }
    printf("in-range+soffset stores landed: %d (expect 64)\n", a);
    printf("voffset OOB (+soffset 64) stores landed: %d; (+soffset 1024, descending) landed: %d  -> 0 means the check is on voffset alone\n", b, c);
    printf("voffset in range, soffset pushes past num_records: landed %d (inside the 16 KiB allocation)\n", d);
    printf("OOB load returned %#x (0 expected)\n", o[0]);
    return 0;
}
