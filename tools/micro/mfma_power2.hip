// micro-benchmark: what sets the shader clock (and with it the attainable fp32-MFMA rate) in the layer evaluators?
// The evaluators' structure with RANDOM data, stripped to MFMA + operand delivery:
//   NP = 1 : a wave owns 32 points; one ds_read_b128 (A fragments of one output tile, 4 k-steps) feeds 4 MFMAs   [product]
//   NP = 2 : a wave owns 64 points; the same read feeds 8 MFMAs (half the LDS bytes per FLOP), one wave per SIMD
//   DMA    : re-stage the layer's 64 KB of fragments from global memory by LDS-DMA every layer (barrier pair), as the
//            forward kernel does, or keep one layer resident (no DMA, no barriers)
// Prints TFLOP/s and the shader clock (s_memtime ticks per s_memrealtime tick x 100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__shared__ f32x4 lds_w[4096];
template <int NP, int DMA, int OCC>
__global__ __launch_bounds__(256, OCC) void k(float *out, const float *in, const float *wts, int layers, unsigned long long *clk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4 *src = reinterpret_cast<const f32x4 *>(wts);
    for (int i = threadIdx.x; i < 4096; i += 256) lds_w[i] = src[i];
    __syncthreads();
    f32x16 acc[NP][4];
    float bin[NP][64];
    for (int p = 0; p < NP; p++)
        for (int i = 0; i < 64; i++) bin[p][i] = in[(threadIdx.x + (i & 15) * 256 + p * 1024) & 8191];
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int l = 0; l < layers; l++) {
        if (DMA) {
            __syncthreads();
            const f32x4 *s2 = src + (size_t)(l & 7) * 4096;
#pragma unroll
            for (int kk = 0; kk < 16; kk++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(s2 + kk * 256 + threadIdx.x),
                                                 (__attribute__((address_space(3))) void *)(lds_w + kk * 256 + wave * 64), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        } else {
            asm volatile("" ::: "memory");      // keep the LDS reads inside the layer loop
        }
#pragma unroll
        for (int p = 0; p < NP; p++)
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[p][t][r] = 0.f;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            f32x4 a[4];
#pragma unroll
            for (int t = 0; t < 4; t++) a[t] = lds_w[(t * 16 + q) * 64 + lane];
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int t = 0; t < 4; t++)
#pragma unroll
                    for (int p = 0; p < NP; p++)
                        acc[p][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][j], bin[p][4 * q + j], acc[p][t], 0, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < NP; p++)
#pragma unroll
            for (int i = 0; i < 64; i++) bin[p][i] = fmaxf(acc[p][i >> 4][i & 15], -1.0f) * 0.37f + 0.011f;   // stays O(1), never zero
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int p = 0; p < NP; p++)
        for (int i = 0; i < 64; i++) s += bin[p][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}
template <int NP, int DMA, int OCC>
void run(const char *name, int blocks) {
    const int layers = 1200 / NP;
    float *out, *in, *wts; unsigned long long *clk;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&in, 8192 * 4); hipMalloc(&wts, 8 * 16384 * 4); hipMalloc(&clk, 16);
    std::vector<float> h(8192), w(8 * 16384);
    srand(11);
    for (auto &v : h) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    for (auto &v : w) v = 0.15f * ((rand() / (float)RAND_MAX) * 2.f - 1.f);      // ~1/sqrt(128): activations stay O(1)
    hipMemcpy(in, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    hipMemcpy(wts, w.data(), 8 * 16384 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NP, DMA, OCC>), dim3(blocks), dim3(256), 0, 0, out, in, wts, 100, clk); hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NP, DMA, OCC>), dim3(blocks), dim3(256), 0, 0, out, in, wts, layers, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
    double flops = (double)blocks * 4 * layers * 256 * NP * 2.0 * 32 * 32 * 2;
    double mhz = (double)c[0] / (double)c[1] * 100.0;
    printf("%-64s %8.2f ms  %7.1f TFLOP/s  clock %5.0f MHz  MFMA-pipe use %5.1f %%\n", name, ms, flops / ms / 1e9, mhz,
           100.0 * (flops / ms / 1e9) / (157.3 * mhz / 2400.0));
}
int main() {
    run<1, 0, 2>("32 pts/wave, 2 waves/SIMD, resident LDS weights", 512);
    run<1, 1, 2>("32 pts/wave, 2 waves/SIMD, LDS-DMA re-stage per layer [product]", 512);
    run<2, 0, 1>("64 pts/wave, 1 wave/SIMD, resident LDS weights", 256);
    run<2, 1, 1>("64 pts/wave, 1 wave/SIMD, LDS-DMA re-stage per layer", 256);
    run<1, 0, 1>("32 pts/wave, 1 wave/SIMD, resident LDS weights", 256);
    run<1, 1, 1>("32 pts/wave, 1 wave/SIMD, LDS-DMA re-stage per layer", 256);
    return 0;
}
