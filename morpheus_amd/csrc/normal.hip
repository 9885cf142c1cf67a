// Finite-difference normals: the glue around the six SDF taps, and the per-sample position assembly.
//
// Reference (plain PyTorch there, ~25 elementwise launches forward and ~40 backward per call; a real-view training step
// calls it four times):
//   finite_difference_normal  models/model.py:367-385   taps = clamp(x +- eps e_k, -bound, bound), k = x,y,z;
//                                                        raw_k = 0.5 * (sdf(tap_k+) - sdf(tap_k-)) / eps
//   normal                    models/model.py:387-398   normal = nan_to_num(safe_normalize(raw))
//   safe_normalize            utils.py:70-71            x / sqrt(clamp(sum x^2, min = 1e-20))
//   sample assembly           morpheus.py:644-647       xyz = rays_o[ri] + rays_d[ri] * (t_starts + t_ends) / 2
// Each piece is one launch forward and one backward here; the arithmetic is the reference's, operation for operation (no
// FMA contraction in this file): taps and positions are bit-identical to the torch expressions, the normalisation differs
// from torch only in the order of its three-term sum.
#include "common.h"

#pragma clang fp contract(off)

// ---- taps ------------------------------------------------------------------------------------------------------------
// point-major tap order (a sample's six taps adjacent): +x, -x, +y, -y, +z, -z.  topo (optional, [M,C]) is replicated to
// the taps in the same launch.
__global__ __launch_bounds__(256) void fd_taps_kernel(const float *__restrict__ x, const float *__restrict__ topo, int C,
                                                      float eps, float bound, int64_t M, float *__restrict__ taps,
                                                      float *__restrict__ topo6) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per tap
    if (i >= 6 * M) return;
    const int64_t m = i / 6;
    const int k = (int)(i - 6 * m);
    const int axis = k >> 1;
    const float off = (k & 1) ? -eps : eps;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float v = x[m * 3 + a] + (a == axis ? off : 0.0f);
        taps[i * 3 + a] = fminf(fmaxf(v, -bound), bound);
    }
    if (topo)
        for (int c = 0; c < C; c++) topo6[i * C + c] = topo[m * C + c];
}

__global__ __launch_bounds__(256) void fd_taps_bwd_kernel(const float *__restrict__ x, const float *__restrict__ g_taps,
                                                          const float *__restrict__ g_topo6, int C, float eps, float bound,
                                                          int64_t M, float *__restrict__ g_x, float *__restrict__ g_topo) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per sample
    if (m >= M) return;
    float gx[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < 6; k++) {
        const int axis = k >> 1;
        const float off = (k & 1) ? -eps : eps;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float v = x[m * 3 + a] + (a == axis ? off : 0.0f);
            if (g_taps && v >= -bound && v <= bound) gx[a] += g_taps[(m * 6 + k) * 3 + a];   // clamp passes its gradient inside [-b, b]
        }
    }
    if (g_x) {
        g_x[m * 3 + 0] = gx[0];
        g_x[m * 3 + 1] = gx[1];
        g_x[m * 3 + 2] = gx[2];
    }
    if (g_topo)
        for (int c = 0; c < C; c++) {
            float s = 0.f;
            for (int k = 0; k < 6; k++) s += g_topo6[(m * 6 + k) * C + c];
            g_topo[m * C + c] = s;
        }
}

// ---- central differences + normalisation -------------------------------------------------------------------------------
__device__ __forceinline__ bool mh_finite(float v) { return fabsf(v) <= 3.4028234663852886e38f; }

__global__ __launch_bounds__(256) void fd_normal_kernel(const float *__restrict__ sdf6, float eps, int64_t M,
                                                        float *__restrict__ normal, float *__restrict__ raw) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    float r[3];
    // `tensor / python_scalar` on the GPU is a multiplication by the fp32 reciprocal of the scalar (torch's CUDA
    // div_true kernel): 0.5 * (a - b) / eps  ==  (0.5 * (a - b)) * (1.0f / eps)
    const float inv_eps = 1.0f / eps;
#pragma unroll
    for (int k = 0; k < 3; k++) r[k] = (0.5f * (sdf6[m * 6 + 2 * k] - sdf6[m * 6 + 2 * k + 1])) * inv_eps;
    const float ss = fmaxf((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2], 1e-20f);
    const float len = sqrtf(ss);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        raw[m * 3 + k] = r[k];
        float n = r[k] / len;
        if (n != n) n = 0.f;                                            // nan_to_num: nan -> 0, +-inf -> +-float max
        else if (!mh_finite(n)) n = n > 0.f ? 3.4028234663852886e38f : -3.4028234663852886e38f;
        normal[m * 3 + k] = n;
    }
}

__global__ __launch_bounds__(256) void fd_normal_bwd_kernel(const float *__restrict__ sdf6, const float *__restrict__ g_normal,
                                                            const float *__restrict__ g_raw, float eps, int64_t M,
                                                            float *__restrict__ g_sdf6) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    float r[3], gr[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        r[k] = (0.5f * (sdf6[m * 6 + 2 * k] - sdf6[m * 6 + 2 * k + 1])) * (1.0f / eps);
        if (g_raw) gr[k] = g_raw[m * 3 + k];
    }
    if (g_normal) {
        const float ss0 = (r[0] * r[0] + r[1] * r[1]) + r[2] * r[2];
        const bool clamped = !(ss0 >= 1e-20f);                          // clamp(min): no gradient to ss when ss < min
        const float len = sqrtf(fmaxf(ss0, 1e-20f));
        float gn[3], dot = 0.f;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float n = r[k] / len;
            gn[k] = mh_finite(n) ? g_normal[m * 3 + k] : 0.f;           // nan_to_num passes its gradient where the input is finite
            dot += gn[k] * (r[k] / len);
        }
#pragma unroll
        for (int k = 0; k < 3; k++) gr[k] += clamped ? gn[k] / len : (gn[k] - (r[k] / len) * dot) / len;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float g = (gr[k] * (1.0f / eps)) * 0.5f;
        g_sdf6[m * 6 + 2 * k] = g;
        g_sdf6[m * 6 + 2 * k + 1] = -g;
    }
}

// ---- sample positions ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sample_positions_kernel(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                               const int32_t *__restrict__ ray_idx, const float *__restrict__ ts,
                                                               const float *__restrict__ te, int64_t M, float *__restrict__ xyz) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const int r = ray_idx[m];
    const float tm = (ts[m] + te[m]) / 2.0f;
#pragma unroll
    for (int a = 0; a < 3; a++) xyz[m * 3 + a] = rays_o[r * 3 + a] + rays_d[r * 3 + a] * tm;
}

// gradient of the gather: samples are packed ray-major, so the sum over a ray's samples is a segment sum -- one wavefront
// per ray, no atomics, no sort (torch's index backward sorts the M indices: 0.33 ms per gather at M = 137 k)
__global__ __launch_bounds__(256) void sample_positions_bwd_kernel(const float *__restrict__ g_xyz, const float *__restrict__ ts,
                                                                   const float *__restrict__ te, const int32_t *__restrict__ ray_start,
                                                                   const int32_t *__restrict__ ray_cnt, int N,
                                                                   float *__restrict__ g_o, float *__restrict__ g_d) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= N) return;
    const int64_t base = ray_start[r];
    const int n = ray_cnt[r];
    float so[3] = {0.f, 0.f, 0.f}, sd[3] = {0.f, 0.f, 0.f};
    for (int i = lane; i < n; i += 64) {
        const int64_t m = base + i;
        const float tm = (ts[m] + te[m]) / 2.0f;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float g = g_xyz[m * 3 + a];
            so[a] += g;
            sd[a] += g * tm;
        }
    }
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            so[a] += __shfl_xor(so[a], o);
            sd[a] += __shfl_xor(sd[a], o);
        }
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            g_o[r * 3 + a] = so[a];
            g_d[r * 3 + a] = sd[a];
        }
    }
}

// ---- MultiCode.sample (models/deform_code.py:20-38) --------------------------------------------------------------------------
// Per level: a [C, size] table sampled linearly in time with F.grid_sample(align_corners=True, border padding): the
// normalised coordinate 2t-1 is mapped back by ((x+1)/2)*(size-1).  One thread per (time, level, channel); the reference
// (and the torch form this replaces) spends a dozen launches per call and as many again in backward.
struct CodeLevels {
    const float *v[3];
    float *g[3];
    int size[3];
};

__device__ __forceinline__ void code_taps(float t, int size, int &i0, int &i1, float &fr) {
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    const float r = (((t * 2.0f - 1.0f) + 1.0f) / 2.0f) * (float)(size - 1);
    const float r0 = floorf(r);
    fr = r - r0;
    i0 = min(max((int)r0, 0), size - 1);
    i1 = min(i0 + 1, size - 1);
}

__global__ __launch_bounds__(256) void multicode_fwd_kernel(const float *__restrict__ t, CodeLevels lv, int C, int F,
                                                            float *__restrict__ out) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= F * 3 * C) return;
    const int f = gid / (3 * C), l = (gid / C) % 3, c = gid % C;
    int i0, i1;
    float fr;
    code_taps(t[f], lv.size[l], i0, i1, fr);
    const float *v = lv.v[l] + (int64_t)c * lv.size[l];
    out[gid] = v[i0] * (1.0f - fr) + v[i1] * fr;
}

__global__ __launch_bounds__(256) void multicode_bwd_kernel(const float *__restrict__ t, const float *__restrict__ g_out,
                                                            CodeLevels lv, int C, int F) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= F * 3 * C) return;
    const int f = gid / (3 * C), l = (gid / C) % 3, c = gid % C;
    int i0, i1;
    float fr;
    code_taps(t[f], lv.size[l], i0, i1, fr);
    float *g = lv.g[l] + (int64_t)c * lv.size[l];
    const float go = g_out[gid];
    atomicAdd(g + i0, go * (1.0f - fr));
    atomicAdd(g + i1, go * fr);
}

// ---- free-space / near-surface SDF losses on packed samples (utils.py:91-113, called at morpheus.py:789) ---------------------
// Per sample m of ray r: z = (ts + te)/2, target = rays_depth[r];  front = z < target - trunc  or  (target < 0 and z < 3.5);
// bnd = target - z (10 if target < 0);  smask = |bnd| <= trunc and target > 0 (and rays_mask[r] > 0.5);  n = front + smask + 1e-8;
//   fs += max(max(exp(-5 p) - 1, p - bnd), 0) * front / n        sl += |p - bnd| * smask / n        nd += (target != 0)
// and the caller divides both sums by nd.  The torch form is ~25 launches forward, ~20 backward and two [M] gathers.
struct SdfLossTerm {
    float fs, sl, dfs, dsl;   // the sample's two loss terms and their derivatives w.r.t. the predicted sdf
    bool nz;
};

__device__ __forceinline__ SdfLossTerm sdf_loss_term(float z, float target, float p, float trunc, bool masked_in) {
    SdfLossTerm o;
    const bool front = (z < (target - trunc)) || ((target < 0.0f) && (z < 3.5f));
    const float bnd = (target < 0.0f) ? 10.0f : (target - z);
    const bool smask = (fabsf(bnd) <= trunc) && (target > 0.0f) && masked_in;
    const float n = (front ? 1.0f : 0.0f) + (smask ? 1.0f : 0.0f) + 1e-8f;
    const float a = expf(-5.0f * p) - 1.0f, b = p - bnd;
    const float mx = fmaxf(a, b);
    const float dmx = (a > b) ? (-5.0f * expf(-5.0f * p)) : ((a < b) ? 1.0f : 0.5f * (1.0f - 5.0f * expf(-5.0f * p)));
    const bool pos = mx > 0.0f;
    o.fs = front ? (pos ? mx : 0.0f) / n : 0.0f;
    o.dfs = (front && pos) ? dmx / n : 0.0f;
    const float d = p - bnd;
    o.sl = smask ? fabsf(d) / n : 0.0f;
    o.dsl = smask ? ((d > 0.0f) ? 1.0f : ((d < 0.0f) ? -1.0f : 0.0f)) / n : 0.0f;
    o.nz = target != 0.0f;
    return o;
}

__global__ __launch_bounds__(256) void sdf_losses_kernel(const float *__restrict__ pred, const float *__restrict__ ts,
                                                         const float *__restrict__ te, const int32_t *__restrict__ ray_idx,
                                                         const float *__restrict__ rays_depth, const float *__restrict__ rays_mask,
                                                         float trunc, int64_t M, const int32_t *__restrict__ n_valid,
                                                         float *__restrict__ sums /*[3]: fs, sl, nd*/) {
    // grid-stride over the samples, then ONE set of atomics per workgroup (<= 128 workgroups): a wave-level atomicAdd per 64
    // samples serialised 6 900 same-address atomics at the 140 000 samples of a training step (80 us; now 6)
    const int64_t rows = n_valid ? min((int64_t)max(*n_valid, 0), M) : M;
    float fs = 0.f, sl = 0.f, nd = 0.f;
    for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < rows; m += (int64_t)gridDim.x * blockDim.x) {
        const int r = ray_idx[m];
        const SdfLossTerm o = sdf_loss_term((ts[m] + te[m]) / 2.0f, rays_depth[r], pred[m], trunc, rays_mask ? rays_mask[r] > 0.5f : true);
        fs += o.fs;
        sl += o.sl;
        nd += o.nz ? 1.0f : 0.0f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        fs += __shfl_xor(fs, o);
        sl += __shfl_xor(sl, o);
        nd += __shfl_xor(nd, o);
    }
    __shared__ float red[3][4];
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = fs;
        red[1][threadIdx.x >> 6] = sl;
        red[2][threadIdx.x >> 6] = nd;
    }
    __syncthreads();
    if (threadIdx.x < 3) atomicAdd(sums + threadIdx.x, (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]));
}

// g_pred[m] = (g_fs * dfs_m + g_sl * dsl_m) / nd;  g = [g_fs, g_sl] on the device, nd = sums[2] of the forward launch
__global__ __launch_bounds__(256) void sdf_losses_bwd_kernel(const float *__restrict__ pred, const float *__restrict__ ts,
                                                             const float *__restrict__ te, const int32_t *__restrict__ ray_idx,
                                                             const float *__restrict__ rays_depth,
                                                             const float *__restrict__ rays_mask, float trunc, int64_t M,
                                                             const int32_t *__restrict__ n_valid,
                                                             const float *__restrict__ sums, const float *__restrict__ g_fs,
                                                             const float *__restrict__ g_sl, float *__restrict__ g_pred) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    if (n_valid && m >= (int64_t)*n_valid) {      // padding behind the packed samples (fixed-capacity sampling): no loss, no gradient
        g_pred[m] = 0.0f;
        return;
    }
    const int r = ray_idx[m];
    const SdfLossTerm o = sdf_loss_term((ts[m] + te[m]) / 2.0f, rays_depth[r], pred[m], trunc, rays_mask ? rays_mask[r] > 0.5f : true);
    const float nd = sums[2];
    g_pred[m] = ((g_fs ? *g_fs : 0.0f) * o.dfs + (g_sl ? *g_sl : 0.0f) * o.dsl) / nd;
}

// ---- C ABI -----------------------------------------------------------------------------------------------------------------
static inline unsigned blocks_for(int64_t n) { return (unsigned)((n + 255) / 256); }

extern "C" int mh_fd_taps(const float *x, const float *topo, int32_t topo_dim, float eps, float bound, int64_t M, float *taps,
                          float *topo6, void *stream) {
    if (M == 0) return MH_OK;
    if (M < 0 || !x || !taps || !(eps > 0.f) || !(bound > 0.f) || (topo && (!topo6 || topo_dim <= 0)) || 6 * M > 0x7fffffffLL * 256)
        return MH_ERR_ARG;
    hipLaunchKernelGGL(fd_taps_kernel, dim3(blocks_for(6 * M)), dim3(256), 0, mh_stream(stream), x, topo, (int)topo_dim, eps,
                       bound, M, taps, topo6);
    MH_CHECK_LAUNCH();
    return MH_OK;
}

extern "C" int mh_fd_taps_bwd(const float *x, const float *g_taps, const float *g_topo6, int32_t topo_dim, float eps,
                              float bound, int64_t M, float *g_x, float *g_topo, void *stream) {
    if (M == 0) return MH_OK;
    if (M < 0 || !x || !(eps > 0.f) || !(bound > 0.f) || (g_x && !g_taps) || (g_topo && (!g_topo6 || topo_dim <= 0)))
        return MH_ERR_ARG;
    hipLaunchKernelGGL(fd_taps_bwd_kernel, dim3(blocks_for(M)), dim3(256), 0, mh_stream(stream), x, g_taps, g_topo6,
                       (int)topo_dim, eps, bound, M, g_x, g_topo);
    MH_CHECK_LAUNCH();
    return MH_OK;
}

extern "C" int mh_fd_normal_fwd(const float *sdf6, float eps, int64_t M, float *normal, float *raw, void *stream) {
    if (M == 0) return MH_OK;
    if (M < 0 || !sdf6 || !normal || !raw || !(eps > 0.f)) return MH_ERR_ARG;
    hipLaunchKernelGGL(fd_normal_kernel, dim3(blocks_for(M)), dim3(256), 0, mh_stream(stream), sdf6, eps, M, normal, raw);
    MH_CHECK_LAUNCH();
    return MH_OK;
}

extern "C" int mh_fd_normal_bwd(const float *sdf6, const float *g_normal, const float *g_raw, float eps, int64_t M,
                                float *g_sdf6, void *stream) {
    if (M == 0) return MH_OK;
    if (M < 0 || !sdf6 || !g_sdf6 || !(eps > 0.f)) return MH_ERR_ARG;
    hipLaunchKernelGGL(fd_normal_bwd_kernel, dim3(blocks_for(M)), dim3(256), 0, mh_stream(stream), sdf6, g_normal, g_raw, eps, M,
                       g_sdf6);
    MH_CHECK_LAUNCH();
    return MH_OK;
}

extern "C" int mh_sample_positions(const float *rays_o, const float *rays_d, const int32_t *ray_idx, const float *t_starts,
                                   const float *t_ends, int64_t M, float *xyz, void *stream) {
    if (M == 0) return MH_OK;
    if (M < 0 || !rays_o || !rays_d || !ray_idx || !t_starts || !t_ends || !xyz) return MH_ERR_ARG;
    hipLaunchKernelGGL(sample_positions_kernel, dim3(blocks_for(M)), dim3(256), 0, mh_stream(stream), rays_o, rays_d, ray_idx,
                       t_starts, t_ends, M, xyz);
    MH_CHECK_LAUNCH();
    return MH_OK;
}

extern "C" int mh_sample_positions_bwd(const float *g_xyz, const float *t_starts, const float *t_ends, const int32_t *ray_start,
                                       const int32_t *ray_cnt, int32_t N, float *g_o, float *g_d, void *stream) {
    if (N == 0) return MH_OK;
    if (N < 0 || !g_xyz || !t_starts || !t_ends || !ray_start || !ray_cnt || !g_o || !g_d) return MH_ERR_ARG;
    hipLaunchKernelGGL(sample_positions_bwd_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, mh_stream(stream), g_xyz,
                       t_starts, t_ends, ray_start, ray_cnt, (int)N, g_o, g_d);
    MH_CHECK_LAUNCH();
    return MH_OK;
}

extern "C" int mh_multicode_fwd(const float *t, const float *v0, const float *v1, const float *v2, int32_t s0, int32_t s1,
                                int32_t s2, int32_t C, int32_t F, float *out, void *stream) {
    if (F == 0) return MH_OK;
    if (!t || !v0 || !v1 || !v2 || !out || F < 0 || C <= 0 || s0 <= 0 || s1 <= 0 || s2 <= 0) return MH_ERR_ARG;
    CodeLevels lv = {{v0, v1, v2}, {nullptr, nullptr, nullptr}, {(int)s0, (int)s1, (int)s2}};
    hipLaunchKernelGGL(multicode_fwd_kernel, dim3(blocks_for((int64_t)F * 3 * C)), dim3(256), 0, mh_stream(stream), t, lv, (int)C,
                       (int)F, out);
    MH_CHECK_LAUNCH();
    return MH_OK;
}

extern "C" int mh_multicode_bwd(const float *t, const float *g_out, float *g0, float *g1, float *g2, int32_t s0, int32_t s1,
                                int32_t s2, int32_t C, int32_t F, void *stream) {
    if (F == 0) return MH_OK;
    if (!t || !g_out || !g0 || !g1 || !g2 || F < 0 || C <= 0 || s0 <= 0 || s1 <= 0 || s2 <= 0) return MH_ERR_ARG;
    CodeLevels lv = {{nullptr, nullptr, nullptr}, {g0, g1, g2}, {(int)s0, (int)s1, (int)s2}};
    hipLaunchKernelGGL(multicode_bwd_kernel, dim3(blocks_for((int64_t)F * 3 * C)), dim3(256), 0, mh_stream(stream), t, g_out, lv,
                       (int)C, (int)F);
    MH_CHECK_LAUNCH();
    return MH_OK;
}

extern "C" int mh_sdf_losses_fwd(const float *pred_sdf, const float *t_starts, const float *t_ends, const int32_t *ray_idx,
                                 const float *rays_depth, const float *rays_mask, float trunc, int64_t M,
                                 const int32_t *n_valid, float *sums, void *stream) {
    if (!sums) return MH_ERR_ARG;
    if (!mh_zero_async(sums, 3 * sizeof(float), mh_stream(stream))) return MH_ERR_LAUNCH;
    if (M == 0) return MH_OK;
    if (M < 0 || !pred_sdf || !t_starts || !t_ends || !ray_idx || !rays_depth) return MH_ERR_ARG;
    const unsigned blocks = blocks_for(M) < 128u ? blocks_for(M) : 128u;
    hipLaunchKernelGGL(sdf_losses_kernel, dim3(blocks), dim3(256), 0, mh_stream(stream), pred_sdf, t_starts, t_ends, ray_idx,
                       rays_depth, rays_mask, trunc, M, n_valid, sums);
    MH_CHECK_LAUNCH();
    return MH_OK;
}

extern "C" int mh_sdf_losses_bwd(const float *pred_sdf, const float *t_starts, const float *t_ends, const int32_t *ray_idx,
                                 const float *rays_depth, const float *rays_mask, float trunc, int64_t M,
                                 const int32_t *n_valid, const float *sums, const float *g_fs, const float *g_sl, float *g_pred,
                                 void *stream) {
    if (M == 0) return MH_OK;
    if (M < 0 || !pred_sdf || !t_starts || !t_ends || !ray_idx || !rays_depth || !sums || !g_pred) return MH_ERR_ARG;
    hipLaunchKernelGGL(sdf_losses_bwd_kernel, dim3(blocks_for(M)), dim3(256), 0, mh_stream(stream), pred_sdf, t_starts, t_ends,
                       ray_idx, rays_depth, rays_mask, trunc, M, n_valid, sums, g_fs, g_sl, g_pred);
    MH_CHECK_LAUNCH();
    return MH_OK;
}
