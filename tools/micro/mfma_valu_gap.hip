// micro-benchmark (round 5): how many single-issue instructions ride for free in the shadow of a v_mfma_f32_32x32x16_bf16?
//
// MI355X_MICROARCH.md measures <= 5 hidden per 32-cycle MFMA gap for one wave per SIMD (32.4 cycles per MFMA hand-placed, 35.8
// compiler-scheduled).  Round 4's mfma_bf16_rate.hip reported "~4.8 cycles per VALU instruction, none of it hidden" -- from a stream
// whose ISA was never looked at: hipcc had turned its "4 v_fma_f32 per MFMA" into 7-9 VALU per gap, v_pk_fma_f32 / v_pk_mov_b32
// among them, on a loop-carried dependent chain (profiles/r05_micro_mfma_valu_gap.txt quotes the excerpt).  This program settles
// it with streams whose placement is not left to the compiler:
//   asm  fma<K>    K independent v_fma_f32 behind every MFMA, one asm block per 16 MFMAs (gen_mfma_valu_gap.py)
//   asm  split<K>  the kernels' real slicing arithmetic (mlp_dev.h split2), two value pairs interleaved, K instructions per gap
//   asm  mix<K>    split<K> + one ds_read_b128 per two MFMAs + one global_store_dword per three (the warp kernels' companions)
//   C++  split     MFMA builtins and split2() calls, in hipcc's own order or ordered by __builtin_amdgcn_sched_group_barrier(MFMA 1,
//                  VALU K) -- what a product kernel can use without going to asm
// each at 1 and 2 waves per SIMD (one 4- or 8-wave workgroup per CU, as the warp kernels), live operand slices.
// Prints shader cycles per MFMA per SIMD (s_memtime of wave 0; at 2 waves/SIMD the wave's cycles per MFMA / 2), wall time and the
// effective clock.
// Build: python gen_mfma_valu_gap.py > mfma_valu_gap_gen.h && hipcc --offload-arch=gfx950 -O3 -o mfma_valu_gap mfma_valu_gap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include "mfma_valu_gap_gen.h"
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
union Frag {
    f32x4 f;
    bf16x8 h;
};
extern __shared__ f32x4 lds[];
#define NMFMA 16

#define PROLOGUE                                                                                                        \
    const int lane = threadIdx.x & 63;                                                                                  \
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = frags[i];                                             \
    __syncthreads();                                                                                                    \
    Frag a, b;                                                                                                          \
    a.f = frags[threadIdx.x % 2048];                                                                                    \
    b.f = frags[(threadIdx.x + 1024) % 2048];                                                                           \
    f32x16 acc0, acc1, acc2, acc3;                                                                                      \
    for (int r = 0; r < 16; r++) acc0[r] = acc1[r] = acc2[r] = acc3[r] = 0.f;                                           \
    float s[18];                                                                                                        \
    for (int i = 0; i < 18; i++) s[i] = frags[(threadIdx.x * 7 + i * 131) % 2048][i & 3];                              \
    float cst = 0.999f + 1e-6f * lane;                                                                                  \
    unsigned laddr = lane * 16;                                                                                         \
    float *gaddr = sink + ((size_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63)) * 1024 + lane;                        \
    f32x4 d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0};                                                                         \
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();

#define EPILOGUE                                                                                                        \
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();                        \
    float sum = d0[0] + d1[1];                                                                                          \
    for (int i = 0; i < 18; i++) sum += s[i];                                                                           \
    for (int r = 0; r < 16; r++) sum += acc0[r] + acc1[r] + acc2[r] + acc3[r];                                          \
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;                                                                   \
    if (threadIdx.x == 0 && blockIdx.x == 0) {                                                                          \
        clk[0] = c1 - c0;                                                                                               \
        clk[1] = r1 - r0;                                                                                               \
    }

// (the 18 scratch registers and the two ds_read targets are in/out operands: 4 + 2 + 18 + 3 + 2 = 29 of the 30 an asm block may name)
#define DEF_ASM_KERNEL(NAME, STREAM)                                                                                    \
    template <int OCC>                                                                                                  \
    __global__ __launch_bounds__(256 * OCC, 1) void NAME(float *out, const f32x4 *__restrict__ frags, float *sink, int iters, \
                                                         unsigned long long *clk) {                                     \
        PROLOGUE                                                                                                        \
        for (int it = 0; it < iters; it++)                                                                              \
            asm volatile(STREAM                                                                                         \
                         : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3)                                               \
                         : "v"(a.h), "v"(b.h), "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(s[4]), "v"(s[5]), "v"(s[6]), "v"(s[7]), \
                           "v"(s[8]), "v"(s[9]), "v"(s[10]), "v"(s[11]), "v"(s[12]), "v"(s[13]), "v"(s[14]), "v"(s[15]), "v"(s[16]),   \
                           "v"(s[17]), "v"(cst), "v"(laddr), "v"(gaddr), "v"(d0), "v"(d1)                                \
                         : "memory");                                                                                   \
        EPILOGUE                                                                                                        \
    }
// (scratch registers are WRITTEN by the block though declared as inputs -- an in/out declaration of all of them passes the
//  30-operand limit; nothing else touches them inside the loop, the loop body is this one block, and what they hold afterwards is
//  only summed into the sink)

DEF_ASM_KERNEL(k_fma0, STREAM_FMA_0)
DEF_ASM_KERNEL(k_fma1, STREAM_FMA_1)
DEF_ASM_KERNEL(k_fma2, STREAM_FMA_2)
DEF_ASM_KERNEL(k_fma3, STREAM_FMA_3)
DEF_ASM_KERNEL(k_fma4, STREAM_FMA_4)
DEF_ASM_KERNEL(k_fma5, STREAM_FMA_5)
DEF_ASM_KERNEL(k_fma6, STREAM_FMA_6)
DEF_ASM_KERNEL(k_fma7, STREAM_FMA_7)
DEF_ASM_KERNEL(k_fma8, STREAM_FMA_8)
DEF_ASM_KERNEL(k_split2, STREAM_SPLIT_2)
DEF_ASM_KERNEL(k_split3, STREAM_SPLIT_3)
DEF_ASM_KERNEL(k_split4, STREAM_SPLIT_4)
DEF_ASM_KERNEL(k_split5, STREAM_SPLIT_5)
DEF_ASM_KERNEL(k_split6, STREAM_SPLIT_6)
DEF_ASM_KERNEL(k_mix2, STREAM_MIX_2)
DEF_ASM_KERNEL(k_mix3, STREAM_MIX_3)
DEF_ASM_KERNEL(k_mix4, STREAM_MIX_4)
DEF_ASM_KERNEL(k_mix5, STREAM_MIX_5)

// the same work written in C++ -- MFMA builtins and mlp_dev.h's split2 on P value pairs (11 VALU instructions each) per 16 MFMAs --
// in hipcc's own order, or ORDERED by scheduling-group barriers: (1 MFMA, then K VALU) sixteen times
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2(float x0, float x1, uint32_t &hi, uint32_t &mid, uint32_t &lo) {
    union {
        bf16x2_t b;
        uint32_t u;
    } h, m, l;
    h.b = __builtin_convertvector((f32x2_t){x0, x1}, bf16x2_t);
    const float r0 = x0 - __uint_as_float(h.u << 16), r1 = x1 - __uint_as_float(h.u & 0xffff0000u);
    m.b = __builtin_convertvector((f32x2_t){r0, r1}, bf16x2_t);
    const float s0 = r0 - __uint_as_float(m.u << 16), s1 = r1 - __uint_as_float(m.u & 0xffff0000u);
    l.b = __builtin_convertvector((f32x2_t){s0, s1}, bf16x2_t);
    hi = h.u;
    mid = m.u;
    lo = l.u;
}

template <int K, int OCC, bool SGB>
__global__ __launch_bounds__(256 * OCC, 1) void k_sgb(float *out, const f32x4 *__restrict__ frags, float *sink, int iters,
                                                       unsigned long long *clk) {
    constexpr int P = (NMFMA * K + 5) / 11;       // value pairs whose slicing makes ~K instructions per MFMA
    static_assert(P <= 9, "18 scratch values");
    PROLOGUE
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int g = 0; g < NMFMA; g += 4) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, acc3, 0, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < P; p++) {
            asm volatile("" : "+v"(s[2 * p]), "+v"(s[2 * p + 1]));      // this iteration's values (no hoisting)
            uint32_t hi, mid, lo;
            split2(s[2 * p], s[2 * p + 1], hi, mid, lo);
            asm volatile("" ::"v"(hi), "v"(mid), "v"(lo));
        }
        if (SGB) {
#pragma unroll
            for (int g = 0; g < NMFMA; g++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, K, 0);
            }
        }
    }
    EPILOGUE
}

static uint16_t bf16_rn(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

template <typename KF>
static void run(const char *name, KF kern, int occ, const f32x4 *frags, float *sink) {
    const int blocks = 256, iters = 3000;
    float *out;
    unsigned long long *clk;
    hipMalloc(&out, (size_t)blocks * 256 * occ * 4);
    hipMalloc(&clk, 16);
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256 * occ), 96 * 1024, 0, out, frags, sink, 200, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256 * occ), 96 * 1024, 0, out, frags, sink, iters, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2];
    hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
    const double n_mfma_wave = (double)iters * NMFMA;
    const double flops = (double)blocks * 4 * occ * n_mfma_wave * 2.0 * 32 * 32 * 16;
    const double mhz = (double)c[0] / (double)c[1] * 100.0;
    printf("%-60s %d w/SIMD %6.1f cyc/MFMA/SIMD %7.3f ms %6.0f TFLOP/s  clock %5.0f MHz\n", name, occ, (double)c[0] / n_mfma_wave / occ, ms,
           flops / ms / 1e9, mhz);
    fflush(stdout);
    hipFree(out);
    hipFree(clk);
}

int main() {
    std::vector<uint16_t> live(2048 * 8);
    srand(7);
    for (size_t g = 0; g < live.size(); g++) {
        float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = rand() / (float)RAND_MAX;
        live[g] = bf16_rn(0.3f * sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2));
    }
    f32x4 *dl;
    float *sink;
    hipMalloc(&dl, 2048 * 16);
    hipMalloc(&sink, (size_t)256 * 512 * 1024 * 4 + 65536);
    hipMemcpy(dl, live.data(), 2048 * 16, hipMemcpyHostToDevice);
#define RUN2(label, K)             \
    run(label, K<1>, 1, dl, sink); \
    run(label, K<2>, 2, dl, sink);
    RUN2("asm  bare MFMA stream", k_fma0)
    RUN2("asm  + 1 independent v_fma_f32 per gap", k_fma1)
    RUN2("asm  + 2 independent v_fma_f32 per gap", k_fma2)
    RUN2("asm  + 3 independent v_fma_f32 per gap", k_fma3)
    RUN2("asm  + 4 independent v_fma_f32 per gap", k_fma4)
    RUN2("asm  + 5 independent v_fma_f32 per gap", k_fma5)
    RUN2("asm  + 6 independent v_fma_f32 per gap", k_fma6)
    RUN2("asm  + 7 independent v_fma_f32 per gap", k_fma7)
    RUN2("asm  + 8 independent v_fma_f32 per gap", k_fma8)
    RUN2("asm  + 2 split2 instructions per gap", k_split2)
    RUN2("asm  + 3 split2 instructions per gap", k_split3)
    RUN2("asm  + 4 split2 instructions per gap", k_split4)
    RUN2("asm  + 5 split2 instructions per gap", k_split5)
    RUN2("asm  + 6 split2 instructions per gap", k_split6)
    RUN2("asm  + 2 split2 + ds_read_b128/2 + store/3 per gap", k_mix2)
    RUN2("asm  + 3 split2 + ds_read_b128/2 + store/3 per gap", k_mix3)
    RUN2("asm  + 4 split2 + ds_read_b128/2 + store/3 per gap", k_mix4)
    RUN2("asm  + 5 split2 + ds_read_b128/2 + store/3 per gap", k_mix5)
#define RUNS(label, K, SGB)                    \
    run(label, k_sgb<K, 1, SGB>, 1, dl, sink); \
    run(label, k_sgb<K, 2, SGB>, 2, dl, sink);
    RUNS("C++  split2 x 4 (44 VALU) / 16 MFMAs, hipcc's own order", 3, false)
    RUNS("C++  split2 x 4, sched_group_barrier(MFMA 1, VALU 3)", 3, true)
    RUNS("C++  split2 x 6 (66 VALU) / 16 MFMAs, hipcc's own order", 4, false)
    RUNS("C++  split2 x 6, sched_group_barrier(MFMA 1, VALU 4)", 4, true)
    RUNS("C++  split2 x 7 (77 VALU) / 16 MFMAs, hipcc's own order", 5, false)
    RUNS("C++  split2 x 7, sched_group_barrier(MFMA 1, VALU 5)", 5, true)
    RUNS("C++  split2 x 9 (99 VALU) / 16 MFMAs, sched_group_barrier(MFMA 1, VALU 6)", 6, true)
    return 0;
}
