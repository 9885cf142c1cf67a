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
