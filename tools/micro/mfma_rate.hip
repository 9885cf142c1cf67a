// micro-benchmark: sustained v_mfma_f32_32x32x2_f32 rate on gfx950 under the conditions of the MLP / wgrad kernels
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC, int VARY>
__global__ __launch_bounds__(256) void k(float *out, const float *in, int iters) {
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; n++) for (int r = 0; r < 16; r++) acc[n][r] = 0.f;
    float a[16], b[16];
    for (int i = 0; i < 16; i++) { a[i] = in[threadIdx.x + i * 256]; b[i] = in[threadIdx.x + 4096 + i * 256]; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++)
#pragma unroll
            for (int n = 0; n < NACC; n++)
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(VARY ? a[i] : a[0], VARY ? b[(i + n) & 15] : b[0], acc[n], 0, 0, 0);
    }
    float s = 0;
    for (int n = 0; n < NACC; n++) for (int r = 0; r < 16; r++) s += acc[n][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC, int VARY>
void run(const char *name, int blocks) {
    float *out, *in; hipMalloc(&out, blocks * 256 * 4); hipMalloc(&in, 8192 * 4); hipMemset(in, 0, 8192 * 4);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, VARY>), dim3(blocks), dim3(256), 0, 0, out, in, iters); hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, VARY>), dim3(blocks), dim3(256), 0, 0, out, in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * 16 * NACC * 2.0 * 32 * 32 * 2;
    printf("%-44s blocks=%5d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks, ms, flops / ms / 1e9);
}
int main() {
    run<4, 0>("4 acc, constant operands, 1 wave/SIMD", 256);
    run<4, 0>("4 acc, constant operands, 2 waves/SIMD", 512);
    run<4, 1>("4 acc, varying operands, 1 wave/SIMD", 256);
    run<4, 1>("4 acc, varying operands, 2 waves/SIMD", 512);
    run<1, 1>("1 acc (dependent chain), 1 wave/SIMD", 256);
    run<2, 1>("2 acc, 2 waves/SIMD", 512);
    return 0;
}
