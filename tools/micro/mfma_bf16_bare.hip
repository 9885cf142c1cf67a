// The bare stream of v_mfma_f32_32x32x16_bf16 on live operand slices (hi / mid / lo of N(0, 0.3) values), A and B from registers,
// nothing else: the reference point of tools/gpu/power_clock.py -- socket power and shader clock of the matrix pipe alone, for
// `seconds` of back-to-back launches.   ./mfma_bf16_bare [seconds=2] [waves_per_simd=1] [zero=0]
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_bf16_bare mfma_bf16_bare.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
union Frag {
    f32x4 f;
    bf16x8 h;
};
__global__ __launch_bounds__(512) void k(float *out, const f32x4 *__restrict__ frags, int iters, unsigned long long *clk) {
    Frag a[3], b[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        a[j].f = frags[(j * 2048 + threadIdx.x) % 6144];
        b[j].f = frags[(j * 2048 + 1024 + threadIdx.x) % 6144];
    }
    f32x16 acc[4];
    for (int n = 0; n < 4; n++)
        for (int r = 0; r < 16; r++) acc[n][r] = 0.f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 6; q++)
#pragma unroll
            for (int n = 0; n < 4; n++) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(q + n) % 3].h, b[q % 3].h, acc[n], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int n = 0; n < 4; n++)
        for (int r = 0; r < 16; r++) s += acc[n][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        clk[0] = c1 - c0;
        clk[1] = r1 - r0;
    }
}
static uint16_t bf16_rn(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    u += 0x7fff + ((u >> 16) & 1);
    return (uint16_t)(u >> 16);
}
static float bf16_f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float x;
    memcpy(&x, &u, 4);
    return x;
}
int main(int argc, char **argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 2.0;
    const int occ = argc > 2 ? atoi(argv[2]) : 1, zero = argc > 3 ? atoi(argv[3]) : 0;
    std::vector<uint16_t> live(6144 * 8, 0);
    srand(7);
    if (!zero)
        for (int g = 0; g < 2048 * 8; g++) {
            float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = rand() / (float)RAND_MAX;
            float x = 0.3f * sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
            uint16_t h = bf16_rn(x);
            float r = x - bf16_f(h);
            uint16_t m = bf16_rn(r);
            live[g] = h;
            live[2048 * 8 + g] = m;
            live[4096 * 8 + g] = bf16_rn(r - bf16_f(m));
        }
    f32x4 *d;
    float *out;
    unsigned long long *clk;
    hipMalloc(&d, 6144 * 16);
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&clk, 16);
    hipMemcpy(d, live.data(), 6144 * 16, hipMemcpyHostToDevice);
    const int iters = 20000, threads = 256 * occ;
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, d, 1000, clk);
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    int launches = 0;
    double el = 0;
    while (el < seconds) {
        hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, d, iters, clk);
        hipDeviceSynchronize();
        launches++;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    unsigned long long c[2];
    hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
    const double flops = (double)launches * 256 * (threads / 64) * iters * 24 * 2.0 * 32 * 32 * 16;
    printf("bare v_mfma_f32_32x32x16_bf16, %s operands, %d wave(s)/SIMD: %d launches in %.2f s, %.1f TFLOP/s of slice products, shader clock %.0f MHz "
           "(%.1f cycles per MFMA and SIMD)\n", zero ? "zero" : "live", occ, launches, el, flops / el / 1e12, (double)c[0] / (double)c[1] * 100.0,
           (double)c[0] / ((double)iters * 24 * occ));
    return 0;
}
