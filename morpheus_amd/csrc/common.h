// Shared device/host helpers for libmorpheus_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/morpheus_hip.h"

#define MH_WAVE 64

#define MH_CHECK_LAUNCH()                                  \
    do {                                                   \
        if (hipGetLastError() != hipSuccess) return MH_ERR_LAUNCH; \
    } while (0)

static inline hipStream_t mh_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int mh_lane() { return threadIdx.x & 63; }

// Per-device host state.  The library keeps NO state that depends on which device a call runs on in plain statics: what
// is cached (a device's CU count, "this kernel's dynamic-LDS limit was raised on this device") is indexed by the current
// device, so one process may drive several GPUs (the product runs one process per GPU; this is belt and braces).  The
// guarded actions are idempotent, so the benign race of two threads doing one twice needs no lock.
#define MH_MAX_DEVICES 64
static inline int mh_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return 0;
    return dev < MH_MAX_DEVICES ? dev : MH_MAX_DEVICES - 1;   // devices beyond the table share its last entry's slot (re-done harmlessly)
}
struct MhOncePerDevice {
    unsigned char done[MH_MAX_DEVICES];
    bool need(int dev) const { return dev >= MH_MAX_DEVICES - 1 || !done[dev]; }
    void mark(int dev) { done[dev] = 1; }
};

// compute units of the current device, queried once per device (256 on MI355X; partitioned modes expose fewer)
static inline int mh_cu_count() {
    static int cus[MH_MAX_DEVICES];
    const int dev = mh_device();
    if (cus[dev] <= 0 || dev == MH_MAX_DEVICES - 1) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n;
    }
    return cus[dev];
}

// Zero-fill as an ordinary KERNEL.  hipMemsetAsync becomes a memset NODE when the stream is being captured into a HIP graph,
// and on ROCm 7.2 such nodes were observed not to hold their place in the replayed order (the d/dx buffer of the binned
// hash-grid backward was accumulated into before it was cleared, from the second replay on -- the first one found fresh,
// zeroed pool memory; tests/test_gpu_render.py::test_graphed_real_view_step_replays_the_eager_step).  A kernel node does.
static __global__ __launch_bounds__(256) void mh_zero_kernel(uint32_t *__restrict__ p, int64_t n_words) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0u;
}
static inline bool mh_zero_async(void *p, size_t bytes, hipStream_t stream) {      // bytes: a multiple of 4, p 4-byte aligned
    if (bytes == 0) return true;
    const int64_t n = (int64_t)(bytes / 4);
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(mh_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<uint32_t *>(p), n);
    return hipGetLastError() == hipSuccess;
}
