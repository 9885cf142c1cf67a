// micro-benchmark (round 4): what rate does the bf16 matrix pipe SUSTAIN on MI355X, chip-wide, and what sets it?
// The b3 warp kernels issue 6 x v_mfma_f32_32x32x16_bf16 per fp32 MAC and reach 0.33-0.46 of the pipe's nominal 2.5 PFLOP/s;
// the same binary runs 22 % faster on all-zero operands (profiles/r04_ab_pipelined_warp_fwd.txt).  This program strips the
// kernels down to the MFMA stream and adds their companions back one at a time:
//   operands   zero | live (hi / mid / lo slices of N(0,1) values: what split2 produces)
//   A from     registers | LDS (three ds_read_b128 per six MFMAs, the kernels' ratio)
//   fillers    none | ds_read_b64_tr_b16 (two per MFMA: the LDS transpose read a fused backward-data + weight-gradient tile
//              would use) | 4 or 8 VALU (v_fma_f32) per MFMA (the chain kernels average ~4 non-MFMA
//              instructions per MFMA over a layer, ~7 in the quarter that carries an epilogue)
//   occupancy  1 | 2 waves per SIMD
// and prints TFLOP/s of slice products, the effective shader clock (s_memtime / s_memrealtime) and the pipe use at THAT clock.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_bf16_rate mfma_bf16_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
union Frag {
    f32x4 f;
    bf16x8 h;
    uint32_t u[4];
};
extern __shared__ f32x4 lds[];

// LDSA: A fragments from LDS (6144 float4 = one 128 x 128 layer's three planes);  FILL: 0 none, 1 tr_b16 reads, 2 / 3: 8 / 4 VALU per MFMA
template <int LDSA, int FILL, int OCC>
__global__ __launch_bounds__(256 * OCC, OCC) void k(float *out, const f32x4 *__restrict__ frags, int iters, unsigned long long *clk) {
    const int lane = threadIdx.x & 63;          // OCC waves per SIMD = one workgroup of 4 * OCC waves sharing one 96 KB LDS copy
    for (int i = threadIdx.x; i < 6144; i += 256 * OCC) lds[i] = frags[i];
    __syncthreads();
    Frag a[3], b[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        a[j].f = frags[(j * 2048 + threadIdx.x) % 6144];
        b[j].f = frags[(j * 2048 + 1024 + threadIdx.x) % 6144];
    }
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
    float v0 = (float)lane, v1 = 1.0f, v2 = 0.5f, v3 = 0.25f;
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned loff = lane;
    for (int it = 0; it < iters; it++) {
        asm volatile("" : "+v"(loff));           // the reads belong to THIS iteration (a layer re-reads its slices per tile)
#pragma unroll
        for (int s = 0; s < 8; s++) {
#pragma unroll
            for (int tp = 0; tp < 4; tp += 2) {
                Frag ah[2], am[2], al[2];
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    if (LDSA) {
                        const f32x4 *w = lds + loff + ((tp + t) * 8 + s) * 64;
                        ah[t].f = w[0];
                        am[t].f = w[2048];
                        al[t].f = w[4096];
                    } else {
                        ah[t] = a[0];
                        am[t] = a[1];
                        al[t] = a[2];
                    }
                }
#define MM(A, B)                                                                                                      \
    _Pragma("unroll") for (int t = 0; t < 2; t++) {                                                                   \
        acc[tp + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[t].h, B.h, acc[tp + t], 0, 0, 0);                      \
        if (FILL == 1) {                                                                                              \
            unsigned long long d0, d1;                                                                                \
            const unsigned addr = (unsigned)((lane & 15) * 8 + (lane >> 4) * 128 + s * 512);                          \
            asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:2048" : "=v"(d0), "=v"(d1) : "v"(addr)); \
            asm volatile("" ::"v"(d0), "v"(d1));                                                                      \
        } else if (FILL >= 2) {                                                                                       \
            _Pragma("unroll") for (int q = 0; q < (FILL == 2 ? 2 : 1); q++) {                                                           \
                v0 = __builtin_fmaf(v0, 0.999f, v1);                                                                  \
                v1 = __builtin_fmaf(v1, 0.998f, v2);                                                                  \
                v2 = __builtin_fmaf(v2, 0.997f, v3);                                                                  \
                v3 = __builtin_fmaf(v3, 0.996f, v0);                                                                  \
            }                                                                                                         \
        }                                                                                                             \
    }
                MM(al, b[0]);
                MM(am, b[1]);
                MM(ah, b[2]);
                MM(am, b[0]);
                MM(ah, b[1]);
                MM(ah, b[0]);
#undef MM
            }
        }
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sum = v0 + v1 + v2 + v3;
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) sum += acc[t][r];
    out[blockIdx.x * 256 * OCC + threadIdx.x] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        clk[0] = c1 - c0;
        clk[1] = r1 - r0;
    }
}

static uint16_t bf16_rn(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf16_f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

template <int LDSA, int FILL, int OCC>
static void run(const char *name, const f32x4 *frags, int blocks) {
    const int iters = 400;
    float *out;
    unsigned long long *clk;
    hipMalloc(&out, (size_t)blocks * 256 * OCC * 4);
    hipMalloc(&clk, 16);
    hipFuncSetAttribute((const void *)k<LDSA, FILL, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, 6144 * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<LDSA, FILL, OCC>), dim3(blocks), dim3(256 * OCC), 6144 * 16, 0, out, frags, 40, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<LDSA, FILL, OCC>), dim3(blocks), dim3(256 * OCC), 6144 * 16, 0, out, frags, iters, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2];
    hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
    const double flops = (double)blocks * 4 * OCC * iters * 8 * 2 * 12 * 2.0 * 32 * 32 * 16;      // waves x iters x steps x pairs x 12 MFMAs
    const double mhz = (double)c[0] / (double)c[1] * 100.0;
    const double tf = flops / ms / 1e9;
    printf("%-74s %7.2f ms  %7.0f TFLOP/s = %4.2f of 2500   clock %5.0f MHz   pipe use at that clock %5.1f %%\n", name, ms, tf, tf / 2500.0,
           mhz, 100.0 * tf / (2500.0 * mhz / 2400.0));
    hipFree(out);
    hipFree(clk);
}

int main() {
    std::vector<uint16_t> zero(6144 * 8, 0), live(6144 * 8);
    srand(7);
    for (int g = 0; g < 2048 * 8; g++) {          // planes: hi | mid | lo of the same N(0,1)-ish values
        float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = rand() / (float)RAND_MAX;
        float x = 0.3f * sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
        uint16_t h = bf16_rn(x);
        float r = x - bf16_f(h);
        uint16_t m = bf16_rn(r);
        uint16_t l = bf16_rn(r - bf16_f(m));
        live[g] = h;
        live[2048 * 8 + g] = m;
        live[4096 * 8 + g] = l;
    }
    f32x4 *dz, *dl;
    hipMalloc(&dz, 6144 * 16);
    hipMalloc(&dl, 6144 * 16);
    hipMemcpy(dz, zero.data(), 6144 * 16, hipMemcpyHostToDevice);
    hipMemcpy(dl, live.data(), 6144 * 16, hipMemcpyHostToDevice);
    const int cu = 256;
    run<0, 0, 1>("zero operands, A from registers, 1 wave/SIMD", dz, cu);
    run<0, 0, 1>("live operands, A from registers, 1 wave/SIMD", dl, cu);
    run<0, 0, 2>("live operands, A from registers, 2 waves/SIMD", dl, cu);
    run<1, 0, 1>("zero operands, A from LDS (3 ds_read_b128 / 6 MFMAs), 1 wave/SIMD", dz, cu);
    run<1, 0, 1>("live operands, A from LDS, 1 wave/SIMD", dl, cu);
    run<1, 0, 2>("live operands, A from LDS, 2 waves/SIMD  [the kernels' configuration]", dl, cu);
    run<1, 1, 2>("live operands, A from LDS, 2 waves/SIMD + 2 ds_read_b64_tr_b16 per MFMA", dl, cu);
    run<0, 1, 1>("live operands, A from registers, 1 wave/SIMD + 2 ds_read_b64_tr_b16 per MFMA", dl, cu);
    run<1, 3, 2>("live operands, A from LDS, 2 waves/SIMD + 4 v_fma_f32 per MFMA  [the kernels' density]", dl, cu);
    run<0, 3, 1>("live operands, A from registers, 1 wave/SIMD + 4 v_fma_f32 per MFMA", dl, cu);
    run<1, 2, 2>("live operands, A from LDS, 2 waves/SIMD + 8 v_fma_f32 per MFMA", dl, cu);
    run<0, 2, 1>("live operands, A from registers, 1 wave/SIMD + 8 v_fma_f32 per MFMA", dl, cu);
    return 0;
}
