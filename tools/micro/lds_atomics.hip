// micro-benchmark: LDS atomic throughput on gfx950 (f32 add vs u32 add vs u64 add), conflict-free and 4-way same-address
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE, int CONFLICT>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    __shared__ unsigned long long buf[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) buf[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x;
    int idx = CONFLICT ? (lane / 4) : lane;  // CONFLICT: 4 lanes share an address
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int a = (idx + u * 257 + it * 31) & 4095;
            if (MODE == 0) atomicAdd(reinterpret_cast<float *>(buf) + a, 1.0f);
            if (MODE == 1) atomicAdd(reinterpret_cast<unsigned *>(buf) + a, 1u);
            if (MODE == 2) atomicAdd(buf + a, 1ull);
            if (MODE == 4) atomicAdd(reinterpret_cast<double *>(buf) + a, 1.0);
            if (MODE == 5) __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float *)(reinterpret_cast<float *>(buf) + a), 1.0f, 0, 0, false);
            if (MODE == 3) reinterpret_cast<float *>(buf)[a] += 1.0f;  // plain RMW (racy) as a reference rate
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (float)buf[1];
}
template <int MODE, int CONFLICT>
void run(const char *name) {
    float *out; hipMalloc(&out, 4096 * 4);
    const int blocks = 1024, iters = 200;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, CONFLICT>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, CONFLICT>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)blocks * 256 * iters * 16;
    printf("%-28s %8.3f ms  %7.1f G lane-atomics/s  (%.2f cycles per wave-instr per CU @2.2GHz, 256 CUs)\n", name, ms, ops / ms / 1e6,
           ms * 1e-3 * 2.2e9 / (ops / 64 / 256));
}
int main() {
    run<0, 0>("ds_add_f32 conflict-free"); run<0, 1>("ds_add_f32 4-way same addr");
    run<1, 0>("ds_add_u32 conflict-free"); run<1, 1>("ds_add_u32 4-way same addr");
    run<2, 0>("ds_add_u64 conflict-free"); run<2, 1>("ds_add_u64 4-way same addr");
    run<4, 0>("ds_add_f64 conflict-free"); run<4, 1>("ds_add_f64 4-way same addr");
    run<5, 0>("ds_faddf builtin conflict-free");
    run<3, 0>("plain f32 RMW");
    return 0;
}
