// micro-benchmark: does VALU / global-store / LDS-read work issued between fp32 MFMAs of ONE wave per SIMD hide in the
// MFMA shadow?  (design question for a two-tile interleaved MLP kernel)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__shared__ f32x4 lds[4096];
template <int NVALU, int NST, int NLDS>
__global__ __launch_bounds__(256, 1) void k(float *out, const float *in, float *sink, int iters) {
    f32x16 acc[4];
    for (int n = 0; n < 4; n++) for (int r = 0; r < 16; r++) acc[n][r] = 0.f;
    float a[16], b[16], v[16];
    for (int i = 0; i < 16; i++) { a[i] = in[threadIdx.x + i * 256]; b[i] = in[threadIdx.x + 4096 + i * 256]; v[i] = a[i] + 1.f; }
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = f32x4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    float *sp = sink + (size_t)blockIdx.x * 256 * 64 + threadIdx.x;
    f32x4 la = lds[threadIdx.x];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
#pragma unroll
            for (int n = 0; n < 4; n++) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i] + la[n], b[(i + n) & 15], acc[n], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NVALU; j++) v[(i + j) & 15] = fmaxf(v[(i + j) & 15] * 1.0001f + 0.5f, 0.f);
            if (NST && (i % (16 / NST) == 0)) sp[(it & 3) * 16384 + i * 256] = v[i];
            if (NLDS && (i % (16 / NLDS) == 0)) la = lds[(threadIdx.x + i * 64 + it) & 4095];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int n = 0; n < 4; n++) for (int r = 0; r < 16; r++) s += acc[n][r];
    for (int i = 0; i < 16; i++) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + la[0];
}
template <int NVALU, int NST, int NLDS>
void run(const char *name) {
    const int blocks = 256, iters = 2000;
    float *out, *in, *sink; hipMalloc(&out, blocks * 256 * 4); hipMalloc(&in, 8192 * 4); hipMemset(in, 0, 8192 * 4);
    hipMalloc(&sink, (size_t)blocks * 256 * 64 * 4 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NVALU, NST, NLDS>), dim3(blocks), dim3(256), 0, 0, out, in, sink, iters); hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NVALU, NST, NLDS>), dim3(blocks), dim3(256), 0, 0, out, in, sink, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * 16 * 4 * 2.0 * 32 * 32 * 2;
    printf("%-52s %8.3f ms  %7.1f TFLOP/s\n", name, ms, flops / ms / 1e9);
}
int main() {
    run<0, 0, 0>("MFMA only (4 per group)");
    run<4, 0, 0>("+ 4 VALU (8 instr) per 4 MFMA");
    run<8, 0, 0>("+ 8 VALU (16 instr) per 4 MFMA");
    run<4, 4, 0>("+ 4 VALU + 1 global store per 16 MFMA");
    run<4, 16, 0>("+ 4 VALU + 1 global store per 4 MFMA");
    run<4, 16, 4>("+ 4 VALU + store/4 MFMA + ds_read_b128 per 16 MFMA");
    run<16, 16, 16>("+ 16 VALU + store + ds_read_b128 per 4 MFMA");
    return 0;
}
