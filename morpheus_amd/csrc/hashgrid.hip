// Multiresolution hash-grid encoder, forward + backward, for gfx950.
//
// Semantics: reference external/encoders/gridencoder/src/gridencoder.cu
//   kernel_grid :83-249, kernel_grid_backward :253-349, kernel_input_backward :353-378,
//   get_grid_index :61-79, fast_hash :45-58; host logic grid.py:28-96, :157.
// Design (MI355X-first, not the reference's launch shape):
//   * one lane per (point, level); a wavefront = 4 points x 16 levels, level fastest.  A point's
//     32 features are produced by 16 adjacent lanes -> every store instruction writes 4 x 128 B
//     contiguous rows of the point-major [M, L*2] output (what the MFMA field kernel consumes);
//     no [L,B,C] intermediate and no permute.
//   * level metadata (offsets, resolutions) travels by value in the kernarg segment (SGPRs);
//     resolutions are host-computed float32 ceil(exp2f(l*S)*H) so oracle and kernel agree.
//   * backward recomputes corner indices/weights (no [M, L*3*2] dy_dx tensor: 805 MB at
//     M = 2.1M in the reference); d/dx is reduced over the 16 level-lanes of a point with DPP
//     row operations, embedding gradients go out as fp32 atomics (same as the reference).
//   * both tables (3.2 MB each) are L2-resident per XCD; gathers are 8-byte float2 loads.
#include "common.h"

// explicit fmaf where the reference kernel (nvcc -fmad=true) fuses, nothing else contracted:
// the CPU oracle does the same, so features agree to the last bit of the interpolation.
#pragma clang fp contract(off)

struct GridMeta {
    int32_t offsets[MH_MAX_LEVELS + 1];
    int32_t res[MH_MAX_LEVELS];
};

struct Corner8 {
    uint32_t row[8];
    float w[8];
};

__device__ __forceinline__ uint32_t grid_row(uint32_t cx, uint32_t cy, uint32_t cz, uint32_t res, uint32_t T,
                                             bool dense, bool pow2) {
    // get_grid_index (gridencoder.cu:61-79): dense x + y*res + z*res^2 while the running stride
    // fits, else the xor-prime hash; for D=3 "stride > T after the loop" <=> res^3 > T.
    uint32_t idx;
    if (dense) {
        idx = cx + cy * res + cz * res * res;  // < res^3 <= T: the modulo is the identity
        return idx;
    }
    idx = (cx * 1u) ^ (cy * 2654435761u) ^ (cz * 805459861u);
    return pow2 ? (idx & (T - 1)) : (idx % T);
}

// position inside level: returns false if the point is outside [0,1]^3
__device__ __forceinline__ bool grid_locate(const float *__restrict__ x, int64_t p, float bound, float two_bound,
                                            uint32_t res, uint32_t g[3], float f[3]) {
    bool inb = true;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        float u = (x[p * 3 + d] + bound) / two_bound;  // grid.py:157, fp32 add then IEEE divide
        inb = inb && !(u < 0.0f || u > 1.0f);
        float pos = fminf(fmaxf(fmaf(u, (float)res, -0.5f), 0.0f), (float)(res - 1));
        float fl = floorf(pos);
        g[d] = (uint32_t)fl;
        f[d] = pos - fl;
    }
    return inb;
}

__global__ __launch_bounds__(256) void grid_fwd_kernel(const float *__restrict__ x, const float2 *__restrict__ emb,
                                                       GridMeta meta, float2 *__restrict__ out, int64_t M, int L,
                                                       int n_levels, float bound, float two_bound) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t p = gid / L;
    const int l = (int)(gid - p * L);
    if (p >= M) return;
    float2 r = make_float2(0.f, 0.f);
    if (l < n_levels) {
        const uint32_t res = (uint32_t)meta.res[l];
        const uint32_t T = (uint32_t)(meta.offsets[l + 1] - meta.offsets[l]);
        const bool dense = (uint64_t)res * res * res <= (uint64_t)T;
        const bool pow2 = (T & (T - 1)) == 0;
        uint32_t g[3];
        float f[3];
        if (grid_locate(x, p, bound, two_bound, res, g, f)) {
            const float2 *tab = emb + meta.offsets[l];
            const uint32_t g1x = min(g[0] + 1, res - 1), g1y = min(g[1] + 1, res - 1), g1z = min(g[2] + 1, res - 1);
            float2 v[8];
#pragma unroll
            for (int c = 0; c < 8; c++) {
                uint32_t cx = (c & 1) ? g1x : g[0], cy = (c & 2) ? g1y : g[1], cz = (c & 4) ? g1z : g[2];
                v[c] = tab[grid_row(cx, cy, cz, res, T, dense, pow2)];
            }
#pragma unroll
            for (int c = 0; c < 8; c++) {
                float w = ((c & 1) ? f[0] : 1.f - f[0]) * ((c & 2) ? f[1] : 1.f - f[1]) * ((c & 4) ? f[2] : 1.f - f[2]);
                r.x = fmaf(w, v[c].x, r.x);
                r.y = fmaf(w, v[c].y, r.y);
            }
        }
    }
    out[gid] = r;
}

// sum over the 16 level-lanes of a point (aligned groups of 16 lanes) -- DPP-free portable form
__device__ __forceinline__ float sum16(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    return v;
}

template <bool NEED_DX>
__global__ __launch_bounds__(256) void grid_bwd_kernel(const float2 *__restrict__ grad, const float *__restrict__ x,
                                                       const float2 *__restrict__ emb, GridMeta meta,
                                                       float *__restrict__ grad_emb, float *__restrict__ grad_x,
                                                       int64_t M, int L, int n_levels, float bound, float two_bound) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t p = gid / L;
    const int l = (int)(gid - p * L);
    const bool live = p < M;
    if (!live) p = M - 1;  // keep whole 16-lane groups converged for the shuffles
    float dx[3] = {0.f, 0.f, 0.f};
    if (live && l < n_levels) {
        const uint32_t res = (uint32_t)meta.res[l];
        const uint32_t T = (uint32_t)(meta.offsets[l + 1] - meta.offsets[l]);
        const bool dense = (uint64_t)res * res * res <= (uint64_t)T;
        const bool pow2 = (T & (T - 1)) == 0;
        uint32_t g[3];
        float f[3];
        if (grid_locate(x, p, bound, two_bound, res, g, f)) {
            const float2 gr = grad[gid];
            const uint32_t g1x = min(g[0] + 1, res - 1), g1y = min(g[1] + 1, res - 1), g1z = min(g[2] + 1, res - 1);
            uint32_t row[8];
#pragma unroll
            for (int c = 0; c < 8; c++) {
                uint32_t cx = (c & 1) ? g1x : g[0], cy = (c & 2) ? g1y : g[1], cz = (c & 4) ? g1z : g[2];
                row[c] = grid_row(cx, cy, cz, res, T, dense, pow2);
            }
            float *ge = grad_emb + (size_t)meta.offsets[l] * 2;
#pragma unroll
            for (int c = 0; c < 8; c++) {
                float w = ((c & 1) ? f[0] : 1.f - f[0]) * ((c & 2) ? f[1] : 1.f - f[1]) * ((c & 4) ? f[2] : 1.f - f[2]);
                atomicAdd(ge + (size_t)row[c] * 2 + 0, w * gr.x);
                atomicAdd(ge + (size_t)row[c] * 2 + 1, w * gr.y);
            }
            if (NEED_DX) {
                // dy_dx (gridencoder.cu:205-247): res * sum over the 4 corners of the other two axes of
                // w_other * (table[right] - table[left]); ignores the border clamp on purpose.
                const float2 *tab = emb + meta.offsets[l];
                float2 v[8];
#pragma unroll
                for (int c = 0; c < 8; c++) v[c] = tab[row[c]];
                const float s = (float)res;
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    const int a = (d + 1) % 3, b = (d + 2) % 3;
                    float acc_x = 0.f, acc_y = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int ba = q & 1, bb = (q >> 1) & 1;
                        const float w = s * (ba ? f[a] : 1.f - f[a]) * (bb ? f[b] : 1.f - f[b]);
                        const int lo = (ba << a) | (bb << b);
                        const int hi = lo | (1 << d);
                        acc_x += w * (v[hi].x - v[lo].x);
                        acc_y += w * (v[hi].y - v[lo].y);
                    }
                    dx[d] = gr.x * acc_x + gr.y * acc_y;
                }
            }
        }
    }
    if (NEED_DX) {
        // L == 16: the point's levels are one aligned 16-lane group
        float sx = sum16(dx[0]), sy = sum16(dx[1]), sz = sum16(dx[2]);
        if (live && l == 0) {
            const float inv = 1.0f / two_bound;  // chain factor of u = (x + bound) / (2 bound)
            grad_x[p * 3 + 0] = sx * inv;
            grad_x[p * 3 + 1] = sy * inv;
            grad_x[p * 3 + 2] = sz * inv;
        }
    }
}

static int fill_meta(GridMeta &m, const int32_t *offsets_host, const int32_t *res_host, int L) {
    if (!offsets_host || !res_host || L < 1 || L > MH_MAX_LEVELS) return MH_ERR_ARG;
    for (int i = 0; i <= L; i++) m.offsets[i] = offsets_host[i];
    for (int i = 0; i < L; i++) {
        m.res[i] = res_host[i];
        if (res_host[i] < 2 || offsets_host[i + 1] <= offsets_host[i]) return MH_ERR_ARG;
    }
    return MH_OK;
}

extern "C" int mh_grid_encode_fwd(const float *x, const float *emb, const int32_t *offsets_host,
                                  const int32_t *res_host, float *out, int64_t M, int32_t L, int32_t n_levels,
                                  float bound, void *stream) {
    if (M == 0) return MH_OK;
    if (!x || !emb || !out || M < 0 || n_levels < 0 || n_levels > L || !(bound > 0.f)) return MH_ERR_ARG;
    GridMeta meta;
    int st = fill_meta(meta, offsets_host, res_host, L);
    if (st) return st;
    const int64_t threads = M * L;
    const int64_t blocks = (threads + 255) / 256;
    if (blocks > 0x7fffffffLL) return MH_ERR_ARG;
    hipLaunchKernelGGL(grid_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, mh_stream(stream), x,
                       reinterpret_cast<const float2 *>(emb), meta, reinterpret_cast<float2 *>(out), M, (int)L,
                       (int)n_levels, bound, 2.0f * bound);
    MH_CHECK_LAUNCH();
    return MH_OK;
}

extern "C" int mh_grid_encode_bwd(const float *grad, const float *x, const float *emb, const int32_t *offsets_host,
                                  const int32_t *res_host, float *grad_emb, float *grad_x, int64_t M, int32_t L,
                                  int32_t n_levels, float bound, void *stream) {
    if (M == 0) return MH_OK;
    if (!grad || !x || !emb || !grad_emb || M < 0 || n_levels < 0 || n_levels > L || !(bound > 0.f)) return MH_ERR_ARG;
    if (grad_x && L != 16) return MH_ERR_ARG;  // d/dx reduction is specialised to 16-lane level groups
    GridMeta meta;
    int st = fill_meta(meta, offsets_host, res_host, L);
    if (st) return st;
    const int64_t threads = M * L;
    const int64_t blocks = (threads + 255) / 256;
    if (blocks > 0x7fffffffLL) return MH_ERR_ARG;
    if (grad_x)
        hipLaunchKernelGGL(grid_bwd_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, mh_stream(stream),
                           reinterpret_cast<const float2 *>(grad), x, reinterpret_cast<const float2 *>(emb), meta,
                           grad_emb, grad_x, M, (int)L, (int)n_levels, bound, 2.0f * bound);
    else
        hipLaunchKernelGGL(grid_bwd_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, mh_stream(stream),
                           reinterpret_cast<const float2 *>(grad), x, reinterpret_cast<const float2 *>(emb), meta,
                           grad_emb, grad_x, M, (int)L, (int)n_levels, bound, 2.0f * bound);
    MH_CHECK_LAUNCH();
    return MH_OK;
}
