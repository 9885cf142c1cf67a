// Caller-side glue of the hot path as single launches: the per-sample means of the training losses and the "displace along a
// random direction orthogonal to the normal" step of the normal-smoothness regularisers.
//
// The reference writes these as chains of elementwise torch operators (morpheus.py:518-528 get_ortho_normal_dir, :764-777 the
// in-render perturbation losses, :1090-1145 get_regularization_loss); on this path each chain was 8-25 launches forward and as
// many backward, ~300 of the ~700 launches of a real-view training step whose own kernels take 5.5 ms.
// Every function here is the same arithmetic in fp32, one launch forward (two for a mean: block partials, then one block that
// adds them in a fixed order -- deterministic, no atomics, nothing to zero) and one backward.
#include "common.h"

#define MEAN_BLOCKS 512
#define MEAN_THREADS 256

enum MeanKind { MK_IDENTITY = 0, MK_SQUARE = 1, MK_ABS = 2, MK_ENTROPY = 3, MK_EIKONAL = 4, MK_ABSDIFF = 5, MK_SQDIFF = 6 };

__device__ __forceinline__ float sgnf(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

// f(a[, b]) of one element and its derivative with respect to a (d f / d b = -that for the two difference kinds)
template <bool GRAD>
__device__ __forceinline__ float mean_term(int kind, float a, float b) {
    switch (kind) {
    case MK_IDENTITY: return GRAD ? 1.f : a;
    case MK_SQUARE: return GRAD ? 2.f * a : a * a;
    case MK_ABS: return GRAD ? sgnf(a) : fabsf(a);
    case MK_ABSDIFF: return GRAD ? sgnf(a - b) : fabsf(a - b);
    case MK_SQDIFF: return GRAD ? 2.f * (a - b) : (a - b) * (a - b);
    case MK_ENTROPY: {   // -x log2 x - (1 - x) log2 (1 - x), x = clamp(a, 1e-5, 1 - 1e-5)   (morpheus.py:1093-1096)
        const float lo = 1e-5f, hi = 1.0f - 1e-5f;
        const float x = fminf(fmaxf(a, lo), hi);
        if (!GRAD) return -x * log2f(x) - (1.0f - x) * log2f(1.0f - x);
        // d/dx = -log2 x + log2 (1 - x) (the two 1/ln2 terms cancel); the clamp passes the gradient on [lo, hi]
        return (a >= lo && a <= hi) ? (log2f(1.0f - x) - log2f(x)) : 0.f;
    }
    default: return 0.f;
    }
}

// partial sums [2 * MEAN_BLOCKS]: (sum of f over the valid rows, sum of the row weights)
__global__ __launch_bounds__(MEAN_THREADS) void masked_mean_partial_kernel(int kind, const float *__restrict__ a,
                                                                           const float *__restrict__ b,
                                                                           const float *__restrict__ w_row, int64_t M, int C,
                                                                           const int32_t *__restrict__ n_valid,
                                                                           float *__restrict__ ws) {
    const int64_t rows = n_valid ? min((int64_t)max(*n_valid, 0), M) : M;
    float sf = 0.f, sw = 0.f;
    const int64_t stride = (int64_t)gridDim.x * MEAN_THREADS;
    if (kind == MK_EIKONAL) {   // one value per row of [M, 3]: (|row| - 1)^2   (morpheus.py:1117-1119)
        for (int64_t r = (int64_t)blockIdx.x * MEAN_THREADS + threadIdx.x; r < rows; r += stride) {
            const float x = a[3 * r], y = a[3 * r + 1], z = a[3 * r + 2];
            const float d = sqrtf(x * x + y * y + z * z) - 1.0f;
            const float w = w_row ? w_row[r] : 1.0f;
            sf += d * d * w;
            sw += w;
        }
    } else {
        const int64_t n = rows * C;
        for (int64_t i = (int64_t)blockIdx.x * MEAN_THREADS + threadIdx.x; i < n; i += stride) {
            const float f = mean_term<false>(kind, a[i], b ? b[i] : 0.f);
            if (w_row) {
                const int64_t r = i / C;
                const float w = w_row[r];
                sf += f * w;
                if (i - r * C == 0) sw += w;
            } else {
                sf += f;
            }
        }
    }
    __shared__ float red[2][MEAN_THREADS / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sf += __shfl_xor(sf, o);
        sw += __shfl_xor(sw, o);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = sf;
        red[1][threadIdx.x >> 6] = sw;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float f = 0.f, w = 0.f;
#pragma unroll
        for (int k = 0; k < MEAN_THREADS / 64; k++) {
            f += red[0][k];
            w += red[1][k];
        }
        ws[2 * blockIdx.x] = f;
        ws[2 * blockIdx.x + 1] = w;
    }
}

// out[0] = sum / den, out[1] = den.  den = per_row * max(n_valid, 1) without row weights (the reference's .mean() over the
// samples), max(per_row * sum of the weights, 1) with them (morpheus.py:556: `.sum() / (3 * keep.sum()).clamp(min=1)`)
__global__ __launch_bounds__(MEAN_THREADS) void masked_mean_final_kernel(const float *__restrict__ ws, int n_blocks, int64_t M,
                                                                         int per_row, bool weighted,
                                                                         const int32_t *__restrict__ n_valid,
                                                                         float *__restrict__ out) {
    float sf = 0.f, sw = 0.f;
    for (int k = threadIdx.x; k < n_blocks; k += MEAN_THREADS) {
        sf += ws[2 * k];
        sw += ws[2 * k + 1];
    }
    __shared__ float red[2][MEAN_THREADS / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sf += __shfl_xor(sf, o);
        sw += __shfl_xor(sw, o);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = sf;
        red[1][threadIdx.x >> 6] = sw;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float f = 0.f, w = 0.f;
#pragma unroll
        for (int k = 0; k < MEAN_THREADS / 64; k++) {
            f += red[0][k];
            w += red[1][k];
        }
        const int64_t rows = n_valid ? min((int64_t)max(*n_valid, 0), M) : M;
        const float den = weighted ? fmaxf((float)per_row * w, 1.0f) : (float)per_row * (float)max(rows, (int64_t)1);
        out[0] = f / den;
        out[1] = den;
    }
}

// g_a[i] = g * f'(a_i) * w_row / den inside the valid rows, 0 behind them; g_b = -g_a for the difference kinds
__global__ __launch_bounds__(256) void masked_mean_bwd_kernel(int kind, const float *__restrict__ a, const float *__restrict__ b,
                                                              const float *__restrict__ w_row, int64_t M, int C,
                                                              const int32_t *__restrict__ n_valid, const float *__restrict__ out,
                                                              const float *__restrict__ g, float *__restrict__ g_a,
                                                              float *__restrict__ g_b) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * C) return;
    const int64_t rows = n_valid ? min((int64_t)max(*n_valid, 0), M) : M;
    const int64_t r = i / C;
    float v = 0.f;
    if (r < rows) {
        const float scale = *g / out[1] * (w_row ? w_row[r] : 1.0f);
        if (kind == MK_EIKONAL) {
            const float x = a[3 * r], y = a[3 * r + 1], z = a[3 * r + 2];
            const float n = sqrtf(x * x + y * y + z * z);
            v = n > 0.f ? scale * 2.0f * (n - 1.0f) * a[i] / n : 0.f;
        } else {
            v = scale * mean_term<true>(kind, a[i], b ? b[i] : 0.f);
        }
    }
    if (g_a) g_a[i] = v;
    if (g_b) g_b[i] = -v;
}

extern "C" int64_t mh_masked_mean_workspace_floats(void) { return 2 * MEAN_BLOCKS; }

static inline bool mean_args_ok(int32_t kind, const float *a, const float *b, int64_t M, int32_t C) {
    if (kind < MK_IDENTITY || kind > MK_SQDIFF || M < 0 || C <= 0 || (M > 0 && !a)) return false;
    if ((kind == MK_ABSDIFF || kind == MK_SQDIFF) && M > 0 && !b) return false;
    if (kind == MK_EIKONAL && C != 3) return false;
    return true;
}

extern "C" int mh_masked_mean_fwd(int32_t kind, const float *a, const float *b, const float *w_row, int64_t M, int32_t C,
                                  const int32_t *n_valid, float *ws, float *out, void *stream) {
    if (!mean_args_ok(kind, a, b, M, C) || !ws || !out) return MH_ERR_ARG;
    const int64_t n = kind == MK_EIKONAL ? M : M * C;
    int blocks = (int)((n + 4 * MEAN_THREADS - 1) / (4 * MEAN_THREADS));
    blocks = blocks < 1 ? 1 : (blocks > MEAN_BLOCKS ? MEAN_BLOCKS : blocks);
    hipLaunchKernelGGL(masked_mean_partial_kernel, dim3(blocks), dim3(MEAN_THREADS), 0, mh_stream(stream), (int)kind, a, b, w_row, M,
                       (int)C, n_valid, ws);
    MH_CHECK_LAUNCH();
    hipLaunchKernelGGL(masked_mean_final_kernel, dim3(1), dim3(MEAN_THREADS), 0, mh_stream(stream), (const float *)ws, blocks, M,
                       kind == MK_EIKONAL ? 1 : (int)C, w_row != nullptr, n_valid, out);
    MH_CHECK_LAUNCH();
    return MH_OK;
}

extern "C" int mh_masked_mean_bwd(int32_t kind, const float *a, const float *b, const float *w_row, int64_t M, int32_t C,
                                  const int32_t *n_valid, const float *out, const float *g, float *g_a, float *g_b, void *stream) {
    if (!mean_args_ok(kind, a, b, M, C) || !out || !g || (!g_a && !g_b)) return MH_ERR_ARG;
    if (M == 0) return MH_OK;
    hipLaunchKernelGGL(masked_mean_bwd_kernel, dim3((unsigned)((M * C + 255) / 256)), dim3(256), 0, mh_stream(stream), (int)kind, a, b,
                       w_row, M, (int)C, n_valid, out, g, g_a, g_b);
    MH_CHECK_LAUNCH();
    return MH_OK;
}

// ---- x + scale * (cos(phi) u + sin(phi) v),  u = normalize((n^_y, -n^_x, 0)),  v = n^ x u,  n^ = normalize(n)  -------------
// (morpheus.py:518-528; torch.nn.functional.normalize divides by max(|.|, 1e-12))
struct Ortho {
    float nh[3], u[3], v[3], m, r;
};

__device__ __forceinline__ Ortho ortho_frame(const float *__restrict__ n) {
    Ortho o;
    o.m = fmaxf(sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]), 1e-12f);
    o.nh[0] = n[0] / o.m;
    o.nh[1] = n[1] / o.m;
    o.nh[2] = n[2] / o.m;
    const float ur0 = o.nh[1], ur1 = -o.nh[0];
    o.r = fmaxf(sqrtf(ur0 * ur0 + ur1 * ur1), 1e-12f);
    o.u[0] = ur0 / o.r;
    o.u[1] = ur1 / o.r;
    o.u[2] = 0.f;
    o.v[0] = o.nh[1] * o.u[2] - o.nh[2] * o.u[1];
    o.v[1] = o.nh[2] * o.u[0] - o.nh[0] * o.u[2];
    o.v[2] = o.nh[0] * o.u[1] - o.nh[1] * o.u[0];
    return o;
}

__global__ __launch_bounds__(256) void ortho_perturb_kernel(const float *__restrict__ x, const float *__restrict__ n,
                                                            const float *__restrict__ phi, float scale, int64_t M,
                                                            float *__restrict__ out) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const Ortho o = ortho_frame(n + 3 * m);
    const float c = cosf(phi[m]), s = sinf(phi[m]);
#pragma unroll
    for (int d = 0; d < 3; d++) out[3 * m + d] = x[3 * m + d] + (c * o.u[d] + s * o.v[d]) * scale;
}

__global__ __launch_bounds__(256) void ortho_perturb_bwd_kernel(const float *__restrict__ n, const float *__restrict__ phi,
                                                                const float *__restrict__ g_out, float scale, int64_t M,
                                                                float *__restrict__ g_n) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const float *nn = n + 3 * m;
    const Ortho o = ortho_frame(nn);
    const float c = cosf(phi[m]), s = sinf(phi[m]);
    const float g[3] = {g_out[3 * m] * scale, g_out[3 * m + 1] * scale, g_out[3 * m + 2] * scale};   // d L / d w
    // w = c u + s (n^ x u):  dL/du = c g + s (g x n^),  dL/dn^ = s (u x g)
    float gu[3] = {c * g[0] + s * (g[1] * o.nh[2] - g[2] * o.nh[1]), c * g[1] + s * (g[2] * o.nh[0] - g[0] * o.nh[2]),
                   c * g[2] + s * (g[0] * o.nh[1] - g[1] * o.nh[0])};
    float gn[3] = {s * (o.u[1] * g[2] - o.u[2] * g[1]), s * (o.u[2] * g[0] - o.u[0] * g[2]), s * (o.u[0] * g[1] - o.u[1] * g[0])};
    // u = u_raw / max(|u_raw|, eps), u_raw = (n^_y, -n^_x, 0 * n^_z)
    const float ur0 = o.nh[1], ur1 = -o.nh[0];
    float gur0, gur1;
    if (sqrtf(ur0 * ur0 + ur1 * ur1) > 1e-12f) {
        const float dot = o.u[0] * gu[0] + o.u[1] * gu[1];
        gur0 = (gu[0] - o.u[0] * dot) / o.r;
        gur1 = (gu[1] - o.u[1] * dot) / o.r;
    } else {
        gur0 = gu[0] / o.r;
        gur1 = gu[1] / o.r;
    }
    gn[1] += gur0;
    gn[0] -= gur1;
    // n^ = n / max(|n|, eps)
    if (sqrtf(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]) > 1e-12f) {
        const float dot = o.nh[0] * gn[0] + o.nh[1] * gn[1] + o.nh[2] * gn[2];
#pragma unroll
        for (int d = 0; d < 3; d++) g_n[3 * m + d] = (gn[d] - o.nh[d] * dot) / o.m;
    } else {
#pragma unroll
        for (int d = 0; d < 3; d++) g_n[3 * m + d] = gn[d] / o.m;
    }
}

extern "C" int mh_ortho_perturb_fwd(const float *x, const float *n, const float *phi, float scale, int64_t M, float *out,
                                    void *stream) {
    if (M == 0) return MH_OK;
    if (M < 0 || !x || !n || !phi || !out) return MH_ERR_ARG;
    hipLaunchKernelGGL(ortho_perturb_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, mh_stream(stream), x, n, phi, scale, M, out);
    MH_CHECK_LAUNCH();
    return MH_OK;
}

extern "C" int mh_ortho_perturb_bwd(const float *n, const float *phi, const float *g_out, float scale, int64_t M, float *g_n,
                                    void *stream) {
    if (M == 0) return MH_OK;
    if (M < 0 || !n || !phi || !g_out || !g_n) return MH_ERR_ARG;
    hipLaunchKernelGGL(ortho_perturb_bwd_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, mh_stream(stream), n, phi, g_out, scale,
                       M, g_n);
    MH_CHECK_LAUNCH();
    return MH_OK;
}
