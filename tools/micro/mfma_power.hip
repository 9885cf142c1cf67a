// micro-benchmark: sustained v_mfma_f32_32x32x2_f32 rate and EFFECTIVE SHADER CLOCK with zero vs random operands.
// The spec peak (157.3 TFLOP/s) assumes 2.4 GHz; the chip clocks to its power budget (MI355X_MICROARCH.md "DVFS
// give-back"), and fp32 MFMAs on random data draw more power than on zeros -- the other micro-benchmarks in this directory
// all ran on hipMemset(0) inputs.  s_memtime ticks at the shader clock, s_memrealtime at a constant 100 MHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float *out, const float *in, int iters, unsigned long long *clk) {
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; n++) for (int r = 0; r < 16; r++) acc[n][r] = 0.f;
    float a[16], b[16];
    for (int i = 0; i < 16; i++) { a[i] = in[threadIdx.x + i * 256]; b[i] = in[threadIdx.x + 4096 + i * 256]; }
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++)
#pragma unroll
            for (int n = 0; n < NACC; n++)
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[(i + n) & 15], acc[n], 0, 0, 0);
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int n = 0; n < NACC; n++) for (int r = 0; r < 16; r++) s += acc[n][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}
void run(const char *name, int blocks, float scale) {
    float *out, *in; unsigned long long *clk;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&in, 8192 * 4); hipMalloc(&clk, 16);
    std::vector<float> h(8192);
    srand(7);
    for (auto &v : h) v = scale * ((rand() / (float)RAND_MAX) * 2.f - 1.f);    // small enough that 1.3e6 products stay finite
    hipMemcpy(in, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<4>), dim3(blocks), dim3(256), 0, 0, out, in, 2000, clk); hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<4>), dim3(blocks), dim3(256), 0, 0, out, in, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
    double flops = (double)blocks * 4 * iters * 16 * 4 * 2.0 * 32 * 32 * 2;
    printf("%-52s %8.2f ms  %7.1f TFLOP/s  shader clock %6.0f MHz  (peak at that clock %6.1f)\n", name, ms, flops / ms / 1e9,
           (double)c[0] / (double)c[1] * 100.0, 157.3 * ((double)c[0] / (double)c[1] * 100.0) / 2400.0);
}
int main() {
    run("zero operands, 1 wave/SIMD", 256, 0.f);
    run("random operands (|x|<1e-3), 1 wave/SIMD", 256, 1e-3f);
    run("zero operands, 2 waves/SIMD", 512, 0.f);
    run("random operands (|x|<1e-3), 2 waves/SIMD", 512, 1e-3f);
    run("random operands (|x|<1e-3), 1 wave/SIMD (repeat)", 256, 1e-3f);
    return 0;
}
