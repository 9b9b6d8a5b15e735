// evc_hostcopy.h — every transfer between device memory and host memory the library does not own.
//
// hipMemcpy to / from PAGEABLE host memory (numpy arrays, std::vector, the stack) lets the runtime choose how: small
// copies are staged, larger ones pin the caller's pages on the fly, let the GPU access them directly and unpin them
// again.  That path produced an intermittent "Memory access fault by GPU ... Reason: Unknown" (fault address inside the
// host heap, always at the first synchronisation of evc_download_episodes: several back-to-back copies into adjacent
// numpy arrays that share pages; 2-3 of ~60 GPU test-suite runs, never reproducible in isolation).
// The library therefore never hands pageable memory to the runtime: transfers are chunked through its own page-locked
// bounce buffers (two, so the CPU copy of one chunk overlaps the DMA of the other); memory that IS page-locked already
// (evc_host_register, hipHostMalloc) is copied directly.
#pragma once

#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>

namespace evc {

class HostCopier {
public:
    static constexpr size_t kChunk = 4u << 20;

    // device -> host, ordered after everything already enqueued on `stream`; returns when dst is complete
    hipError_t d2h(void* dst, const void* src_dev, size_t bytes, hipStream_t stream) {
        if (bytes == 0) return hipSuccess;
        if (is_pinned(dst)) {
            hipError_t rc = hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, stream);
            return rc != hipSuccess ? rc : hipStreamSynchronize(stream);
        }
        std::lock_guard<std::mutex> guard(mu_);
        if (hipError_t rc = ensure()) return rc;
        char* d = static_cast<char*>(dst);
        const char* s = static_cast<const char*>(src_dev);
        size_t issued = 0, copied = 0;
        int k = 0;
        while (copied < bytes) {
            if (issued < bytes && issued - copied < 2 * kChunk) {           // keep up to two DMAs in flight
                const size_t len = bytes - issued < kChunk ? bytes - issued : kChunk;
                const int b = (int)((issued / kChunk) & 1);
                if (hipError_t rc = hipMemcpyAsync(buf_[b], s + issued, len, hipMemcpyDeviceToHost, stream)) return rc;
                if (hipError_t rc = hipEventRecord(ev_[b], stream)) return rc;
                issued += len;
                continue;
            }
            const size_t len = bytes - copied < kChunk ? bytes - copied : kChunk;
            const int b = k & 1;
            if (hipError_t rc = hipEventSynchronize(ev_[b])) return rc;
            std::memcpy(d + copied, buf_[b], len);
            copied += len;
            k++;
        }
        return hipSuccess;
    }

    // host -> device, ordered after everything already enqueued on `stream`; returns when src may be reused
    hipError_t h2d(void* dst_dev, const void* src, size_t bytes, hipStream_t stream) {
        if (bytes == 0) return hipSuccess;
        if (is_pinned(src)) {
            hipError_t rc = hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyHostToDevice, stream);
            return rc != hipSuccess ? rc : hipStreamSynchronize(stream);
        }
        std::lock_guard<std::mutex> guard(mu_);
        if (hipError_t rc = ensure()) return rc;
        char* d = static_cast<char*>(dst_dev);
        const char* s = static_cast<const char*>(src);
        bool used[2] = {false, false};
        int k = 0;
        for (size_t off = 0; off < bytes; off += kChunk, k++) {
            const size_t len = bytes - off < kChunk ? bytes - off : kChunk;
            const int b = k & 1;
            if (used[b])
                if (hipError_t rc = hipEventSynchronize(ev_[b])) return rc;      // its previous DMA has read the buffer
            std::memcpy(buf_[b], s + off, len);
            if (hipError_t rc = hipMemcpyAsync(d + off, buf_[b], len, hipMemcpyHostToDevice, stream)) return rc;
            if (hipError_t rc = hipEventRecord(ev_[b], stream)) return rc;
            used[b] = true;
        }
        return hipStreamSynchronize(stream);
    }

    // Page-locked host memory only: the copy is ENQUEUED on `stream` and not waited for (the caller synchronises once, behind
    // several of them); returns false — nothing enqueued — for pageable memory, which goes through d2h / h2d above.
    static bool d2h_async(void* dst, const void* src_dev, size_t bytes, hipStream_t stream, hipError_t& rc) {
        rc = hipSuccess;
        if (bytes == 0) return true;
        if (!is_pinned(dst)) return false;
        rc = hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, stream);
        return true;
    }
    static bool h2d_async(void* dst_dev, const void* src, size_t bytes, hipStream_t stream, hipError_t& rc) {
        rc = hipSuccess;
        if (bytes == 0) return true;
        if (!is_pinned(src)) return false;
        rc = hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyHostToDevice, stream);
        return true;
    }

    // The device-side address of page-locked host memory (hipHostRegister / hipHostMalloc), or nullptr: kernels can read and
    // write it directly over PCIe (evc_step_host's direct mode for small batches).
    static void* device_view(const void* host) {
        if (!host || !is_pinned(host)) return nullptr;
        void* dev = nullptr;
        if (hipHostGetDevicePointer(&dev, const_cast<void*>(host), 0) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        return dev;
    }

    // One copier per DEVICE (the calling engine has made its device current): HIP events belong to the device that was
    // current when they were created and cannot be recorded on another device's stream, so a single process-wide set
    // would break the second engine of a process that drives two GPUs.  The buffers live until the process ends.
    static constexpr int kMaxDevices = 64;
    static HostCopier& instance() {
        static HostCopier per_device[kMaxDevices];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
        return per_device[dev % kMaxDevices];
    }

private:
    static bool is_pinned(const void* p) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
            (void)hipGetLastError();                     // unregistered host memory: not an error for us
            return false;
        }
        return attr.type == hipMemoryTypeHost;
    }
    hipError_t ensure() {
        if (buf_[0]) return hipSuccess;
        for (int b = 0; b < 2; b++) {
            if (hipError_t rc = hipHostMalloc(&buf_[b], kChunk, hipHostMallocDefault)) return rc;
            if (hipError_t rc = hipEventCreateWithFlags(&ev_[b], hipEventDisableTiming)) return rc;
        }
        return hipSuccess;
    }
    std::mutex mu_;
    void* buf_[2] = {nullptr, nullptr};
    hipEvent_t ev_[2] = {nullptr, nullptr};
};

inline hipError_t copy_d2h(void* dst, const void* src_dev, size_t bytes, hipStream_t stream) {
    return HostCopier::instance().d2h(dst, src_dev, bytes, stream);
}
inline hipError_t copy_h2d(void* dst_dev, const void* src, size_t bytes, hipStream_t stream) {
    return HostCopier::instance().h2d(dst_dev, src, bytes, stream);
}

}  // namespace evc
