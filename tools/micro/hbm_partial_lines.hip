// Does touching only the FRONT of every 128-byte line save HBM bandwidth on this box?  (Half of a parked activation / gradient
// row is ReLU zeros: a row whose non-zeros were compacted to its front could be read and written short -- DESIGN.md section 8.1.)
// A buffer far larger than L2 + MALL; every lane moves one 16-byte piece; only the first K pieces of each 128-byte line are
// touched, K = 2, 4, 6, 8 (32 ... 128 bytes).  Reported: useful GB/s (bytes touched / time) and line GB/s (lines x 128 B / time).
// If time follows the bytes touched, short rows pay; if it follows the lines, the memory side moves whole lines.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/hbm_partial_lines tools/micro/hbm_partial_lines.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int K, bool WRITE>
__global__ __launch_bounds__(256) void touch(f4 *__restrict__ buf, size_t n_lines, float *sink) {
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const size_t total = n_lines * K, stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const size_t line = i / K, piece = i % K;
        f4 *p = buf + line * 8 + piece;
        if (WRITE) *p = f4{1.f, 2.f, 3.f, (float)piece};
        else acc += *p;
    }
    if (!WRITE && acc[0] + acc[1] + acc[2] + acc[3] == 1.2345f) sink[0] = acc[0];
}

template <class F>
static float time_ms(F f, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f();
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(a));
        f();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
    }
    return best;
}

template <int K, bool WRITE>
static void run(f4 *buf, size_t n_lines, float *sink) {
    const float ms = time_ms([&] { hipLaunchKernelGGL((touch<K, WRITE>), dim3(256 * 16), dim3(256), 0, 0, buf, n_lines, sink); }, 5);
    printf("%s first %3d B of every 128-B line: %7.3f ms   useful %6.0f GB/s   lines x 128 B %6.0f GB/s\n", WRITE ? "write" : "read ", K * 16, ms,
           n_lines * K * 16.0 / ms / 1e6, n_lines * 128.0 / ms / 1e6);
}

int main() {
    const size_t bytes = (size_t)16 << 30, n_lines = bytes / 128;      // 16 GB: 60 x the 256 MB MALL
    f4 *buf;
    float *sink;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 0, bytes));
    run<8, false>(buf, n_lines, sink);
    run<6, false>(buf, n_lines, sink);
    run<4, false>(buf, n_lines, sink);
    run<2, false>(buf, n_lines, sink);
    run<8, true>(buf, n_lines, sink);
    run<6, true>(buf, n_lines, sink);
    run<4, true>(buf, n_lines, sink);
    run<2, true>(buf, n_lines, sink);
    return 0;
}
