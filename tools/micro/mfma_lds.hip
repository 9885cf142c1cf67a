// micro-benchmark: the layer evaluators' inner loop in isolation -- per k-quad 4 ds_read_b128 (A fragments of 4 output
// tiles, conflict-free) feeding 16 fp32 MFMAs whose B operands are registers -- at 1 and 2 waves per SIMD, with and
// without the per-layer barrier pair.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__shared__ f32x4 lds_w[4096];
template <int BARRIER, int OCC>
__global__ __launch_bounds__(256, OCC) void k(float *out, const float *in, int layers) {
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 256) lds_w[i] = f32x4{1.f, 0.5f, 0.25f, 0.125f};
    __syncthreads();
    f32x16 acc[4];
    for (int n = 0; n < 4; n++) for (int r = 0; r < 16; r++) acc[n][r] = 0.f;
    float bin[64];
    for (int i = 0; i < 64; i++) bin[i] = in[threadIdx.x + (i & 15) * 256];
    for (int l = 0; l < layers; l++) {
        if (BARRIER) __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; q++) {
            f32x4 a[4];
#pragma unroll
            for (int t = 0; t < 4; t++) a[t] = lds_w[(t * 16 + q) * 64 + lane];
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][j], bin[4 * q + j], acc[t], 0, 0, 0);
        }
        if (BARRIER) __syncthreads();
#pragma unroll
        for (int i = 0; i < 64; i++) bin[i] = fmaxf(acc[i >> 4][i & 15] * 1e-3f, 0.f);   // next layer's B operands (ReLU)
    }
    float s = 0;
    for (int i = 0; i < 64; i++) s += bin[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int BARRIER, int OCC>
void run(const char *name, int blocks) {
    const int layers = 400;
    float *out, *in; hipMalloc(&out, blocks * 256 * 4); hipMalloc(&in, 8192 * 4); hipMemset(in, 0, 8192 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<BARRIER, OCC>), dim3(blocks), dim3(256), 0, 0, out, in, layers); hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<BARRIER, OCC>), dim3(blocks), dim3(256), 0, 0, out, in, layers);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * layers * 256 * 2.0 * 32 * 32 * 2;
    printf("%-52s %8.3f ms  %7.1f TFLOP/s\n", name, ms, flops / ms / 1e9);
}
int main() {
    run<0, 1>("LDS-fed A, no barriers, 1 wave/SIMD", 256);
    run<0, 2>("LDS-fed A, no barriers, 2 waves/SIMD", 512);
    run<1, 1>("LDS-fed A, barrier pair per layer, 1 wave/SIMD", 256);
    run<1, 2>("LDS-fed A, barrier pair per layer, 2 waves/SIMD", 512);
    return 0;
}
