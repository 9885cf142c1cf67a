// micro-benchmark: issue cost of the layer epilogue's 64 row stores per lane (one dword per lane per row, rows 128 B apart)
// as (a) global_store_dword with 64-bit VGPR addresses, (b) raw buffer stores (128-bit SGPR descriptor + 32-bit VGPR offset),
// measured with s_memtime around the 64 stores (+ s_waitcnt vmcnt(0) reported separately), one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float *out, long long *t, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pt = lane & 31, h = lane >> 5;
    float *tile = out + ((size_t)blockIdx.x * 4 + wave) * (128 * 32);
    float v[64];
    for (int i = 0; i < 64; i++) v[i] = (float)(i + lane);
    long long issue = 0, total = 0;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(tile, 0, 128 * 32 * 4, 0x00027000);
    for (int it = 0; it < iters; it++) {
        const long long t0 = __builtin_amdgcn_s_memtime();
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 64; i++) tile[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + pt] = v[i];
        } else if (MODE == 3) {
            // 16 x dwordx4 stores of the same 16 KB (what a transposed layout would allow)
            typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int i = 0; i < 16; i++) {
                f4 q = {v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
                reinterpret_cast<f4 *>(tile)[i * 64 + lane] = q;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 64; i++)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v[i]), rs, (pt + 128 * h) * 4, 0 + ((i & 3) + 8 * (i >> 2)) * 128, 0);
        }
        const long long t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long t2 = __builtin_amdgcn_s_memtime();
        issue += t1 - t0;
        total += t2 - t0;
#pragma unroll
        for (int i = 0; i < 64; i++) v[i] += 1.0f;
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = issue / iters; t[1] = total / iters; }
}
template <int MODE>
void run(const char *name) {
    const int blocks = 256;
    float *out; long long *t; hipMalloc(&out, (size_t)blocks * 4 * 128 * 32 * 4); hipMalloc(&t, 16);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, t, 200); hipDeviceSynchronize();
    long long h[2]; hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    printf("%-40s issue of 64 stores: %6lld cycles (%5.1f each)   until complete: %6lld\n", name, h[0], h[0] / 64.0, h[1]);
}
int main() {
    run<0>("global_store_dword (64-bit addresses)");
    run<1>("raw buffer_store_dword (32-bit offsets)");
    run<3>("16 x global_store_dwordx4 (same bytes)");
    return 0;
}
