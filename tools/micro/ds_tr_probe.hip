// ds_read_b64_tr_b16 semantics probe (gfx950).  LDS holds b16 element e at byte 2e with value e.  Pattern 0: lane L reads at byte
// 8 L (the natural image); pattern 1: lane L reads chunk perm(L) = (L * 7 + 3) % 64 -- to see which lane's ADDRESS feeds which result.
//   hipcc --offload-arch=gfx950 -O2 -o ds_tr_probe tools/micro/ds_tr_probe.hip && ./ds_tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(uint16_t *out, int pattern) {
    __shared__ uint16_t lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int L = threadIdx.x;
    const int chunk = pattern == 0 ? L : (L * 7 + 3) % 64;
    uint32_t addr = (uint32_t)(uintptr_t)lds + 8 * chunk;
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[4 * L + 0] = v[0] & 0xffff;
    out[4 * L + 1] = v[0] >> 16;
    out[4 * L + 2] = v[1] & 0xffff;
    out[4 * L + 3] = v[1] >> 16;
}
int main() {
    uint16_t *d, h[256];
    hipMalloc(&d, sizeof(h));
    for (int pattern = 0; pattern < 2; pattern++) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, pattern);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("pattern %d (lane: 4 results as element indices; element e lives in chunk e/4 at position e%%4)\n", pattern);
        int ok = 1;
        for (int L = 0; L < 64; L++) {
            printf("  lane %2d:", L);
            for (int j = 0; j < 4; j++) {
                printf(" %4d", h[4 * L + j]);
                // hypothesis: result j of lane L (16-lane group G = L / 16, i = L % 16) = position (i & 3) of the chunk addressed by
                // lane 16 G + 4 j + (i >> 2)
                const int G = L / 16, i = L % 16, src = 16 * G + 4 * j + (i >> 2);
                const int chunk = pattern == 0 ? src : (src * 7 + 3) % 64;
                if (h[4 * L + j] != 4 * chunk + (i & 3)) ok = 0;
            }
            printf("\n");
        }
        printf("hypothesis result[L][j] = chunk[addr of lane 16 (L / 16) + 4 j + ((L %% 16) >> 2)][L & 3]: %s\n", ok ? "HOLDS" : "FAILS");
    }
    return 0;
}
