// micro-benchmark: two waves per SIMD alternating a burst of 256 fp32 MFMAs (changing operands) with a pause (the layer
// evaluators' pattern).  Does s_setprio reduce the interleaving penalty of two waves issuing MFMA on one SIMD?
//   mode 0: no priority; 1: prio 3 during the burst, 0 in the pause; 2: prio ramps 0..3 through the burst (the wave
//   further along wins); 3: static, by workgroup parity.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE, int PAUSE>
__global__ __launch_bounds__(256, 2) void k(float *out, const float *in, int iters) {
    f32x16 acc[4];
    for (int n = 0; n < 4; n++) for (int r = 0; r < 16; r++) acc[n][r] = 0.f;
    float a[16], b[16];
    for (int i = 0; i < 16; i++) { a[i] = in[threadIdx.x + i * 256]; b[i] = in[threadIdx.x + 4096 + i * 256]; }
    // de-phase the workgroups
    for (int i = 0; i < (int)(blockIdx.x * 37 % 64); i++) __builtin_amdgcn_s_sleep(8);
    if (MODE == 3) { if (blockIdx.x & 1) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0); }
    for (int it = 0; it < iters; it++) {
        if (MODE == 1) __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int qd = 0; qd < 4; qd++) {
            if (MODE == 2) { if (qd == 0) __builtin_amdgcn_s_setprio(0); if (qd == 1) __builtin_amdgcn_s_setprio(1); if (qd == 2) __builtin_amdgcn_s_setprio(2); if (qd == 3) __builtin_amdgcn_s_setprio(3); }
#pragma unroll
            for (int i = 0; i < 16; i++)
#pragma unroll
                for (int n = 0; n < 4; n++)
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(i + qd) & 15], b[(i + n) & 15], acc[n], 0, 0, 0);
        }
        if (MODE == 1 || MODE == 2) __builtin_amdgcn_s_setprio(0);
#pragma unroll 1
        for (int s = 0; s < PAUSE; s++) __builtin_amdgcn_s_sleep(16);   // ~16*64 cycles each
    }
    float s = 0;
    for (int n = 0; n < 4; n++) for (int r = 0; r < 16; r++) s += acc[n][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE, int PAUSE>
void run(const char *name) {
    const int blocks = 512, iters = 400;
    float *out, *in; hipMalloc(&out, blocks * 256 * 4); hipMalloc(&in, 8192 * 4); hipMemset(in, 0, 8192 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, PAUSE>), dim3(blocks), dim3(256), 0, 0, out, in, iters); hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, PAUSE>), dim3(blocks), dim3(256), 0, 0, out, in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * 256 * 2.0 * 32 * 32 * 2;
    printf("%-58s %8.3f ms  %7.1f TFLOP/s\n", name, ms, flops / ms / 1e9);
}
int main() {
    run<0, 0>("no pause, no prio");
    run<1, 0>("no pause, prio 3 in burst");
    run<2, 0>("no pause, ramp");
    run<3, 0>("no pause, static by parity");
    run<0, 4>("pause ~4k cycles (burst 16k), no prio");
    run<1, 4>("pause ~4k, prio 3 in burst");
    run<2, 4>("pause ~4k, ramp");
    run<3, 4>("pause ~4k, static by parity");
    run<0, 10>("pause ~10k cycles, no prio");
    run<1, 10>("pause ~10k, prio 3 in burst");
    run<2, 10>("pause ~10k, ramp");
    run<3, 10>("pause ~10k, static by parity");
    return 0;
}
