// micro-benchmark + layout check: one 128 -> 128 layer of the register-resident MLP chain with EXACT fp32 products from the
// bf16 matrix pipe: every fp32 operand is cut into three bf16 slices (8 + 8 + 8 significand bits, truncation: hi + mid + lo
// == x exactly) and the six significant cross products go through v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
//   (1) correctness: D = W . X for one 32-point tile against a float64 host product (checks the A / B / D lane layouts the
//       kernels rely on, and that the result is fp32-grade);
//   (2) rate: the layer loop with LDS-fed weight slices, the slicing VALU work, ReLU and the 64 parking stores, two waves
//       per SIMD, against the fp32-MFMA skeleton of mfma_pingpong.hip (P: ~20 k cycles per layer and wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
union Frag { f32x4 f; bf16x8 h; uint32_t u[4]; };
extern __shared__ f32x4 lds_dyn[];

__host__ __device__ inline int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// x0, x1 -> packed bf16 pairs of the three slices
__device__ __forceinline__ void split2(float x0, float x1, uint32_t &hi, uint32_t &mid, uint32_t &lo) {
    const uint32_t b0 = __float_as_uint(x0), b1 = __float_as_uint(x1);
    hi = __builtin_amdgcn_perm(b1, b0, 0x07060302);
    const float r0 = x0 - __uint_as_float(b0 & 0xffff0000u), r1 = x1 - __uint_as_float(b1 & 0xffff0000u);
    const uint32_t c0 = __float_as_uint(r0), c1 = __float_as_uint(r1);
    mid = __builtin_amdgcn_perm(c1, c0, 0x07060302);
    const float s0 = r0 - __uint_as_float(c0 & 0xffff0000u), s1 = r1 - __uint_as_float(c1 & 0xffff0000u);
    lo = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302);
}

// B operands of one layer from 64 fp32 activations (accumulator order): k-step s takes registers 8s..8s+7
__device__ __forceinline__ void make_b(const float (&v)[64], Frag (&bh)[8], Frag (&bm)[8], Frag (&bl)[8]) {
#pragma unroll
    for (int s = 0; s < 8; s++)
#pragma unroll
        for (int e = 0; e < 4; e++) split2(v[8 * s + 2 * e], v[8 * s + 2 * e + 1], bh[s].u[e], bm[s].u[e], bl[s].u[e]);
}

// acc[mt] += W . b  with the weight slices in LDS: [plane][mt][s][lane] float4
__device__ __forceinline__ void layer_b3(const f32x4 *w, const Frag (&bh)[8], const Frag (&bm)[8], const Frag (&bl)[8],
                                         f32x16 (&acc)[4], int lane) {
#pragma unroll
    for (int s = 0; s < 8; s++) {
#pragma unroll
        for (int mp = 0; mp < 4; mp += 2) {
            Frag ah[2], am[2], al[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                ah[t].f = w[((0 * 4 + mp + t) * 8 + s) * 64 + lane];
                am[t].f = w[((1 * 4 + mp + t) * 8 + s) * 64 + lane];
                al[t].f = w[((2 * 4 + mp + t) * 8 + s) * 64 + lane];
            }
#pragma unroll
            for (int t = 0; t < 2; t++) acc[mp + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[t].h, bh[s].h, acc[mp + t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 2; t++) acc[mp + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[t].h, bm[s].h, acc[mp + t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 2; t++) acc[mp + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t].h, bl[s].h, acc[mp + t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 2; t++) acc[mp + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[t].h, bh[s].h, acc[mp + t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 2; t++) acc[mp + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t].h, bm[s].h, acc[mp + t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 2; t++) acc[mp + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t].h, bh[s].h, acc[mp + t], 0, 0, 0);
        }
    }
}

// correctness: one wave, one layer, X given feature-major [128][32]; writes D feature-major [128][32]
__global__ void k_check(const f32x4 *wfrag, const float *X, float *D) {
    const int lane = threadIdx.x & 63, pt = lane & 31, h = lane >> 5;
    for (int i = threadIdx.x; i < 3 * 4 * 8 * 64; i += 64) lds_dyn[i] = wfrag[i];
    __syncthreads();
    float v[64];
    for (int t = 0; t < 4; t++)
        for (int r = 0; r < 16; r++) v[16 * t + r] = X[(32 * t + acc_row(r, h)) * 32 + pt];
    Frag bh[8], bm[8], bl[8];
    make_b(v, bh, bm, bl);
    f32x16 acc[4];
    for (int t = 0; t < 4; t++) for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
    layer_b3(lds_dyn, bh, bm, bl, acc, lane);
    for (int t = 0; t < 4; t++)
        for (int r = 0; r < 16; r++) D[(32 * t + acc_row(r, h)) * 32 + pt] = acc[t][r];
}

// rate: 2 x 4-wave workgroups per CU is impossible with 96 KB per layer -> ONE 8-wave workgroup per CU, one layer buffer
template <int SPLIT, int PARK, int DMA>
__global__ __launch_bounds__(512, 2) void k_rate(float *out, const float *in, const f32x4 *wfrag, float *park, int layers) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float v[64];
    for (int i = 0; i < 64; i++) v[i] = in[(threadIdx.x + i * 512) & 8191];
    float *mypark = park + ((size_t)blockIdx.x * 8 + wave) * 4096;
    f32x16 acc[4];
    Frag bh[8], bm[8], bl[8];
    for (int l = 0; l < layers; l++) {
        __syncthreads();
        const f32x4 *src = wfrag + (size_t)(l & 3) * 6144;
        if (DMA || l == 0) {
#pragma unroll
            for (int kk = 0; kk < 12; kk++)       // 96 KB = 6144 float4 = 12 DMA pieces per thread
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + kk * 512 + threadIdx.x),
                                                 (__attribute__((address_space(3))) void *)(lds_dyn + kk * 512 + wave * 64), 16, 0, 0);
        }
        if (SPLIT || l == 0) make_b(v, bh, bm, bl);                 // the slicing runs under the DMA's latency
        for (int t = 0; t < 4; t++) for (int r = 0; r < 16; r++) acc[t][r] = 0.01f;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        layer_b3(lds_dyn, bh, bm, bl, acc, lane);
#pragma unroll
        for (int i = 0; i < 64; i++) {
            float y;
            asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(acc[i >> 4][i & 15] * 0.37f + 0.011f));
            v[i] = y;
            if (PARK) mypark[(size_t)(l & 15) * (size_t)gridDim.x * 8 * 4096 + i * 64 + lane] = y;
        }
    }
    float s = 0;
    for (int i = 0; i < 64; i++) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

static void split_host(float x, uint16_t &h, uint16_t &m, uint16_t &l) {
    uint32_t b; memcpy(&b, &x, 4); uint32_t hb = b & 0xffff0000u; float hf; memcpy(&hf, &hb, 4);
    float r1 = x - hf; uint32_t c; memcpy(&c, &r1, 4); uint32_t mb = c & 0xffff0000u; float mf; memcpy(&mf, &mb, 4);
    float r2 = r1 - mf; uint32_t d; memcpy(&d, &r2, 4);
    h = hb >> 16; m = mb >> 16; l = d >> 16;
}

int main() {
    srand(5);
    std::vector<float> W(128 * 128), X(128 * 32);
    for (auto &v : W) v = 0.3f * ((rand() / (float)RAND_MAX) * 2.f - 1.f);
    for (auto &v : X) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    // weight slices: [plane][mt][s][lane][8 bf16]; lane (i, g): element e = W[32 mt + i][row(s, g, e)], row = 32 (s>>1) + acc_row(8 (s&1) + e, g)
    std::vector<uint16_t> frag(3 * 4 * 8 * 64 * 8 * 4);     // x4: four different "layers" for the rate loop (same content)
    for (int rep = 0; rep < 4; rep++)
    for (int mt = 0; mt < 4; mt++) for (int s = 0; s < 8; s++) for (int lane = 0; lane < 64; lane++) for (int e = 0; e < 8; e++) {
        const int i = lane & 31, g = lane >> 5;
        const int col = 32 * (s >> 1) + acc_row(8 * (s & 1) + e, g);
        uint16_t h, m, l;
        split_host(W[(32 * mt + i) * 128 + col], h, m, l);
        const size_t base = (size_t)rep * 3 * 4 * 8 * 64 * 8;
        frag[base + ((((size_t)0 * 4 + mt) * 8 + s) * 64 + lane) * 8 + e] = h;
        frag[base + ((((size_t)1 * 4 + mt) * 8 + s) * 64 + lane) * 8 + e] = m;
        frag[base + ((((size_t)2 * 4 + mt) * 8 + s) * 64 + lane) * 8 + e] = l;
    }
    f32x4 *dfrag; float *dX, *dD;
    hipMalloc(&dfrag, frag.size() * 2); hipMalloc(&dX, X.size() * 4); hipMalloc(&dD, 128 * 32 * 4);
    hipMemcpy(dfrag, frag.data(), frag.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)k_check, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 98304, 0, dfrag, dX, dD);
    std::vector<float> D(128 * 32);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0, worst32 = 0, scale = 0;
    for (int o = 0; o < 128; o++) for (int p = 0; p < 32; p++) {
        double ref = 0; float f32 = 0.f;
        for (int k = 0; k < 128; k++) { ref += (double)W[o * 128 + k] * (double)X[k * 32 + p]; f32 = fmaf(W[o * 128 + k], X[k * 32 + p], f32); }
        worst = fmax(worst, fabs((double)D[o * 32 + p] - ref)); worst32 = fmax(worst32, fabs((double)f32 - ref)); scale = fmax(scale, fabs(ref));
    }
    printf("layer check: max |bf16x3 - f64| = %.3e   max |fp32 fmaf chain - f64| = %.3e   (max |result| %.2f)\n", worst, worst32, scale);

    const int blocks = 256, layers = 400;
    float *out, *in, *park;
    hipMalloc(&out, blocks * 512 * 4); hipMalloc(&in, 8192 * 4);
    const size_t park_floats = (size_t)16 * 2048 * 4096;
    hipMalloc(&park, park_floats * 4);
    std::vector<float> hin(8192);
    for (auto &v : hin) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    hipMemcpy(in, hin.data(), 8192 * 4, hipMemcpyHostToDevice);
    auto run = [&](const char *name, void (*kern)(float *, const float *, const f32x4 *, float *, int)) {
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 98304, 0, out, in, dfrag, park, 40); hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 98304, 0, out, in, dfrag, park, layers);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double waves = (double)blocks * 8;
        const double alg = waves * layers * 2.0 * 128 * 128 * 32;        // algorithmic fp32 FLOPs
        printf("%-52s %7.2f ms  %6.1f algorithmic TFLOP/s (%.2f x fp32-MFMA peak; bf16 pipe %4.1f %%)  %6.0f cycles / layer / wave pair\n",
               name, ms, alg / ms / 1e9, alg / ms / 1e9 / 157.3, 100.0 * 6 * alg / ms / 1e9 / 2500.0, ms * 1e-3 * 2.4e9 / layers);
    };
    run("slicing + ReLU + parking + DMA (full layer)", k_rate<1, 1, 1>);
    run("no parking stores", k_rate<1, 0, 1>);
    run("no slicing (B slices reused)", k_rate<0, 1, 1>);
    run("no slicing, no parking", k_rate<0, 0, 1>);
    run("no slicing, no parking, no re-staging (MFMA + LDS)", k_rate<0, 0, 0>);
    run("full layer again", k_rate<1, 1, 1>);
    return 0;
}
