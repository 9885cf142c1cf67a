// does LDS-DMA (global_load_lds_dwordx4) reach LDS addresses above 64 KB?  512 threads copy 96 KB global -> LDS by DMA,
// then LDS -> global by ds_read; the host compares.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
extern __shared__ f32x4 lds[];
__global__ __launch_bounds__(512) void k(const f32x4 *src, f32x4 *dst) {
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int kk = 0; kk < 12; kk++)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + kk * 512 + threadIdx.x),
                                         (__attribute__((address_space(3))) void *)(lds + kk * 512 + wave * 64), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 6144; i += 512) dst[i] = lds[i];
}
int main() {
    std::vector<float> h(6144 * 4), o(6144 * 4);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)i;
    f32x4 *s, *d;
    hipMalloc(&s, h.size() * 4); hipMalloc(&d, h.size() * 4);
    hipMemcpy(s, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemset(d, 0, h.size() * 4);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipLaunchKernelGGL(k, dim3(1), dim3(512), 98304, 0, s, d);
    hipMemcpy(o.data(), d, o.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0, first = 0;
    for (size_t i = 0; i < h.size(); i++) if (o[i] != h[i]) { if (!bad) first = i; bad++; }
    printf("LDS-DMA 96 KB: %zu of %zu floats wrong (first at float %zu = byte %zu; got %g)\n", bad, h.size(), first, first * 4, bad ? o[first] : 0.0);
    return 0;
}
