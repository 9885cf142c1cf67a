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

// compute units of the current device, queried once per device (256 on MI355X; partitioned modes expose fewer)
static inline int mh_cu_count() {
    static int cached_dev = -1, cached_cus = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev != cached_dev) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached_cus = n;
        cached_dev = dev;
    }
    return cached_cus;
}

