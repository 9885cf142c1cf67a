// micro-benchmark: two ways of keeping the fp32 MFMA pipe busy in the layer evaluators, WITH the activation parking
// (64 row stores of 256 B per wave and layer -- the expensive part of the epilogue) and the per-layer weight re-staging:
//   P  "product": two independent 4-wave workgroups per CU, each [wait DMA | barrier | 256-MFMA burst | barrier | issue
//       next DMA | ReLU + 64 stores]; the two waves of a SIMD overlap only by chance (measured: both are outside their
//       burst ~20 % of the time)
//   Q  "ping-pong": ONE 8-wave workgroup per CU, weights double-buffered in 2 x 64 KB of LDS, the two waves of a SIMD in
//       strict alternation: while group A bursts layer l, group B runs its epilogue of layer l-1 and vice versa (one
//       workgroup barrier per phase), so a burst is always in flight
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
extern __shared__ f32x4 lds_dyn[];

__device__ __forceinline__ void dma_layer(const f32x4 *src, f32x4 *dst, int tid, int nthreads) {
    // 64 KB = 4096 float4: each thread issues 4096 / nthreads LDS-DMA pieces (wave-uniform LDS base + lane * 16)
    const int wave = tid >> 6;
    for (int k = 0; k < 4096 / nthreads; k++)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + k * nthreads + tid),
                                         (__attribute__((address_space(3))) void *)(dst + k * nthreads + wave * 64), 16, 0, 0);
}

__device__ __forceinline__ void burst(const f32x4 *w, const float (&bin)[64], f32x16 (&acc)[4], int lane) {
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.01f;
#pragma unroll
    for (int q = 0; q < 16; q++) {
        f32x4 a[4];
#pragma unroll
        for (int t = 0; t < 4; t++) a[t] = w[(t * 16 + q) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][j], bin[4 * q + j], acc[t], 0, 0, 0);
    }
}

// same bytes as 16 dwordx4 stores per lane (what a [row/4][point][4] parked layout would allow)
__device__ __forceinline__ void epilogue4(const f32x16 (&acc)[4], float (&bin)[64], float *park, int lane) {
#pragma unroll
    for (int i = 0; i < 64; i += 4) {
        f32x4 y;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            float t;
            asm("v_max_f32 %0, 0, %1" : "=v"(t) : "v"(acc[(i + c) >> 4][(i + c) & 15] * 0.37f + 0.011f));
            bin[i + c] = t;
            y[c] = t;
        }
        *reinterpret_cast<f32x4 *>(park + (i / 4) * 256 + lane * 4) = y;     // 16 stores of 1 KB per wave
    }
}

// ReLU only: no parking (inference form)
__device__ __forceinline__ void epilogue0(const f32x16 (&acc)[4], float (&bin)[64], float *park, int lane) {
#pragma unroll
    for (int i = 0; i < 64; i++) {
        float y;
        asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(acc[i >> 4][i & 15] * 0.37f + 0.011f));
        bin[i] = y;
    }
}

__device__ __forceinline__ void epilogue(const f32x16 (&acc)[4], float (&bin)[64], float *park, int lane) {
#pragma unroll
    for (int i = 0; i < 64; i++) {
        float y;
        asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(acc[i >> 4][i & 15] * 0.37f + 0.011f));
        bin[i] = y;
        park[i * 64 + lane] = y;       // 64 row stores of 256 B
    }
}

// P: product-like; EPI selects the epilogue: 0 = 64 dword stores, 1 = 16 dwordx4 stores, 2 = no stores
template <int EPI>
__global__ __launch_bounds__(256, 2) void kP(float *out, const float *in, const float *wts, float *park, int layers) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 *lw = lds_dyn;
    const f32x4 *src = reinterpret_cast<const f32x4 *>(wts);
    float bin[64];
    for (int i = 0; i < 64; i++) bin[i] = in[(threadIdx.x + i * 256) & 8191];
    f32x16 acc[4];
    float *mypark = park + ((size_t)blockIdx.x * 4 + wave) * 4096;
    __syncthreads();
    dma_layer(src, lw, threadIdx.x, 256);
    for (int l = 0; l < layers; l++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        burst(lw, bin, acc, lane);
        __syncthreads();
        dma_layer(src + (size_t)((l + 1) & 7) * 4096, lw, threadIdx.x, 256);
        float *pk = mypark + (size_t)(l & 15) * (size_t)gridDim.x * 4 * 4096;
        if (EPI == 0) epilogue(acc, bin, pk, lane);
        else if (EPI == 1) epilogue4(acc, bin, pk, lane);
        else epilogue0(acc, bin, pk, lane);
    }
    float s = 0;
    for (int i = 0; i < 64; i++) s += bin[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// Q: ping-pong, 8 waves, double-buffered weights
__global__ __launch_bounds__(512, 2) void kQ(float *out, const float *in, const float *wts, float *park, int layers) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, grp = wave >> 2, gtid = threadIdx.x & 255;
    const f32x4 *src = reinterpret_cast<const f32x4 *>(wts);
    float bin[64];
    for (int i = 0; i < 64; i++) bin[i] = in[(threadIdx.x + i * 256) & 8191];
    f32x16 acc[4];
    float *mypark = park + ((size_t)blockIdx.x * 8 + wave) * 4096;
    // phase p: group A (grp 0) bursts layer p/2 when p is even and runs its epilogue when p is odd; group B is shifted by one
    // phase.  The weights of layer l live in buffer l & 1; they are fetched by the group that is in its epilogue one phase
    // before the first burst that needs them.
    if (grp == 0) dma_layer(src, lds_dyn, gtid, 256);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int phases = 2 * layers + 1;
    for (int p = 0; p < phases; p++) {
        const int q = p - grp;                 // this group's own phase counter
        if (q >= 0 && q < 2 * layers) {
            const int l = q >> 1;
            if ((q & 1) == 0) {
                burst(lds_dyn + (l & 1) * 4096, bin, acc, lane);
            } else {
                // group A fetches layer l+1 for everybody (buffer (l+1)&1 was last read by group B one phase ago)
                if (grp == 0 && l + 1 < layers) dma_layer(src + (size_t)((l + 1) & 7) * 4096, lds_dyn + ((l + 1) & 1) * 4096, gtid, 256);
                epilogue(acc, bin, mypark + (size_t)(l & 15) * (size_t)gridDim.x * 8 * 4096, lane);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __syncthreads();
    }
    float s = 0;
    for (int i = 0; i < 64; i++) s += bin[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <typename K>
void run(const char *name, K kern, int blocks, int threads, size_t lds) {
    const int layers = 600;
    float *out, *in, *wts, *park;
    const size_t park_floats = (size_t)16 * 2048 * 4096;   // 16 layer slots x 2048 waves x 16 KB = 512 MB
    hipMalloc(&out, 4096 * 512 * 4); hipMalloc(&in, 8192 * 4); hipMalloc(&wts, 8 * 16384 * 4); hipMalloc(&park, park_floats * 4);
    std::vector<float> h(8192), w(8 * 16384);
    srand(11);
    for (auto &v : h) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    for (auto &v : w) v = 0.15f * ((rand() / (float)RAND_MAX) * 2.f - 1.f);
    hipMemcpy(in, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    hipMemcpy(wts, w.data(), 8 * 16384 * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, out, in, wts, park, 50); hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, out, in, wts, park, layers);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)blocks * threads / 64;
    const double flops = waves * layers * 256 * 2.0 * 32 * 32 * 2;
    printf("%-58s %8.2f ms  %7.1f TFLOP/s  (%.1f %% of 157.3; parked %.2f TB/s)\n", name, ms, flops / ms / 1e9,
           100.0 * flops / ms / 1e9 / 157.3, waves * layers * 16384.0 / ms / 1e9);
    hipFree(out); hipFree(in); hipFree(wts); hipFree(park);
}
int main() {
    run("P: 2 x 4-wave workgroups per CU (product structure)", kP<0>, 512, 256, 65536);
    run("Q: 1 x 8-wave workgroup per CU, ping-pong phases", kQ, 256, 512, 131072);
    run("P again", kP<0>, 512, 256, 65536);
    run("Q again", kQ, 256, 512, 131072);
    run("P with 16 dwordx4 parking stores instead of 64 dword", kP<1>, 512, 256, 65536);
    run("P without parking stores", kP<2>, 512, 256, 65536);
    run("P with 16 dwordx4 parking stores (again)", kP<1>, 512, 256, 65536);
    return 0;
}
