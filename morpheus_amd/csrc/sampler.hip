// Ray generation + uniform stratified sampler.
//
// Ray generation: reference datasets/utils.py:28-65 (OpenGL pinhole, directions NOT normalised)
// and datasets/dataset.py:363-366 (rays_d = sum_k dirs_k * R[:,k]; rays_o = c2w[:3,3]).
// Sampler: replaces the nerfacc OccGridEstimator.sampling call site (morpheus.py:628-638) with the
// benchmark sampler of SURVEY 8(d); definition shared with oracle/field.py:uniform_samples:
//   slab clip to [-bound,bound]^3 -> [t_near>=0, t_far];  dt = (t_far - t_near) / (S+1);
//   ts_i = t_near + (i + u) * dt;  te_i = t_near + (i + 1 + u) * dt;   misses -> zero-width samples.
// Arithmetic is un-contracted round-to-nearest mul/add/div (no FMA) so the packed
// samples are bit-identical to the oracle's -- samples are *inputs* to every parity test.
#include "common.h"

// No FMA contraction in this file: every mul/add/div below is a separate IEEE round-to-nearest
// operation, exactly the sequence torch executes on the CPU for the oracle.  (HIP's __fadd_rn /
// __fmul_rn are inline header functions compiled BEFORE this pragma and still get fused, so
// plain operators are used instead.)
#pragma clang fp contract(off)

struct Pose {
    float r[9];
    float t[3];
};

// one pinhole ray: pixel idx = j*W + i -> camera direction ((i+.5-cx)/fx, -(j+.5-cy)/fy, -1) rotated by R
__device__ __forceinline__ void pixel_ray(float fx, float fy, float cx, float cy, const Pose &pose, int W, int idx,
                                          float *o, float *d) {
    const int j = idx / W, i = idx - j * W;
    const float d0 = (((float)i + 0.5f) - cx) / fx;
    const float d1 = -((((float)j + 0.5f) - cy) / fy);
    const float d2 = -1.0f;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        d[a] = (d0 * pose.r[a * 3 + 0] + d1 * pose.r[a * 3 + 1]) + d2 * pose.r[a * 3 + 2];
        o[a] = pose.t[a];
    }
}

__global__ __launch_bounds__(256) void generate_rays_kernel(float fx, float fy, float cx, float cy, Pose pose, int H,
                                                            int W, float *__restrict__ rays_o, float *__restrict__ rays_d) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * W) return;
    float o[3], d[3];
    pixel_ray(fx, fy, cx, cy, pose, W, idx, o, d);
#pragma unroll
    for (int a = 0; a < 3; a++) {
        rays_d[idx * 3 + a] = d[a];
        rays_o[idx * 3 + a] = o[a];
    }
}

// slab clip + the i-th stratified interval of one ray (shared by the two uniform-sampler kernels)
__device__ __forceinline__ void uniform_interval(const float *o, const float *d, float bound, int S, int i, float u,
                                                 float *ts, float *te) {
    float tmin = -INFINITY, tmax = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float ta = (-bound - o[a]) / d[a];
        const float tb = (bound - o[a]) / d[a];
        tmin = fmaxf(tmin, fminf(ta, tb));
        tmax = fminf(tmax, fmaxf(ta, tb));
    }
    tmin = fmaxf(tmin, 0.0f);
    if (!(tmax > tmin)) {
        tmin = 0.f;
        tmax = 0.f;
    }
    const float dt = (tmax - tmin) / (float)(S + 1);
    *ts = tmin + ((float)i + u) * dt;
    *te = tmin + (((float)i + 1.0f) + u) * dt;
}

__global__ __launch_bounds__(256) void sample_uniform_kernel(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                             const float *__restrict__ jitter, int N, int S, float bound,
                                                             int32_t *__restrict__ ray_idx, float *__restrict__ t_starts,
                                                             float *__restrict__ t_ends, float *__restrict__ xyz,
                                                             int32_t *__restrict__ ray_start, int32_t *__restrict__ ray_cnt) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)N * S;
    if (gid >= total) return;
    const int r = (int)(gid / S);
    const int i = (int)(gid - (int64_t)r * S);
    float o[3], d[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        o[a] = rays_o[r * 3 + a];
        d[a] = rays_d[r * 3 + a];
    }
    float ts, te;
    uniform_interval(o, d, bound, S, i, jitter[r], &ts, &te);
    ray_idx[gid] = r;
    t_starts[gid] = ts;
    t_ends[gid] = te;
    if (xyz) {
        const float tm = (ts + te) / 2.0f;
#pragma unroll
        for (int a = 0; a < 3; a++) xyz[gid * 3 + a] = o[a] + d[a] * tm;
    }
    if (i == 0) {
        ray_start[r] = (int32_t)((int64_t)r * S);
        ray_cnt[r] = S;
    }
}

// Ray generation fused with the uniform sampler (BASELINE.json north_star: "fused ray-generate + stratified sampler"):
// ray r looks through pixel pix[r] (NULL -> r, the whole image) of the pinhole camera; every sample thread rebuilds
// its ray from the 12 pose floats instead of reading a rays buffer, thread i == 0 of a ray also writes rays_o/rays_d
// for the renderer.  The per-iteration pixel draw is dataset.py:412-423 (index = randint(0, H*W, (ray_num,)),
// rays[:, index]).  Same operation sequence as the two kernels above -> bit-identical outputs.
__global__ __launch_bounds__(256) void rays_sample_uniform_kernel(float fx, float fy, float cx, float cy, Pose pose, int W,
                                                                  const int32_t *__restrict__ pix,
                                                                  const float *__restrict__ jitter, int N, int S, float bound,
                                                                  float *__restrict__ rays_o, float *__restrict__ rays_d,
                                                                  int32_t *__restrict__ ray_idx, float *__restrict__ t_starts,
                                                                  float *__restrict__ t_ends, float *__restrict__ xyz,
                                                                  int32_t *__restrict__ ray_start, int32_t *__restrict__ ray_cnt) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)N * S;
    if (gid >= total) return;
    const int r = (int)(gid / S);
    const int i = (int)(gid - (int64_t)r * S);
    float o[3], d[3];
    pixel_ray(fx, fy, cx, cy, pose, W, pix ? pix[r] : r, o, d);
    float ts, te;
    uniform_interval(o, d, bound, S, i, jitter[r], &ts, &te);
    ray_idx[gid] = r;
    t_starts[gid] = ts;
    t_ends[gid] = te;
    if (xyz) {
        const float tm = (ts + te) / 2.0f;
#pragma unroll
        for (int a = 0; a < 3; a++) xyz[gid * 3 + a] = o[a] + d[a] * tm;
    }
    if (i == 0) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            rays_o[r * 3 + a] = o[a];
            rays_d[r * 3 + a] = d[a];
        }
        ray_start[r] = (int32_t)((int64_t)r * S);
        ray_cnt[r] = S;
    }
}

// ---- occupancy-grid ray marcher ----------------------------------------------------------------
// Stands where nerfacc's OccGridEstimator.sampling stands in the reference (morpheus.py:628-638: fixed
// render_step_size, one stratified near-plane jitter per ray, alpha_thre = 0, early_stop_eps = 0, sigma_fn =
// None => the output depends only on the rays and the binary grid).  nerfacc's source is not in the reference
// tree, so the interval placement is defined here (and mirrored by oracle/field.py:march_samples):
//   [t_near, t_far] = slab clip to the AABB, t_near >= 0;   t0 = t_near + u * step;
//   interval k: ts = t0 + k*step, te = min(ts + step, t_far), kept while ts < t_far and only if the grid cell
//   containing the interval's midpoint is occupied.
// The packed layout is ragged: march into per-ray slot rows, scan the counts, pack (march_wave_kernel, march_pack_kernel).
struct MarchRay {
    float o[3], d[3];
    float t0, tfar;
};

__device__ __forceinline__ MarchRay march_setup(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                const float *__restrict__ jitter, int r, float step, float bound) {
    MarchRay m;
    float tmin = -INFINITY, tmax = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        m.o[a] = rays_o[r * 3 + a];
        m.d[a] = rays_d[r * 3 + a];
        const float ta = (-bound - m.o[a]) / m.d[a];
        const float tb = (bound - m.o[a]) / m.d[a];
        tmin = fmaxf(tmin, fminf(ta, tb));
        tmax = fminf(tmax, fmaxf(ta, tb));
    }
    tmin = fmaxf(tmin, 0.0f);
    if (!(tmax > tmin)) {
        tmin = 0.f;
        tmax = 0.f;
    }
    m.t0 = tmin + (jitter ? jitter[r] : 0.0f) * step;
    m.tfar = tmax;
    return m;
}

__device__ __forceinline__ bool march_occupied(const MarchRay &m, float ts, float te, float bound, int R,
                                               const uint8_t *__restrict__ binary) {
    const float tm = (ts + te) / 2.0f;
    int c[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float x = m.o[a] + m.d[a] * tm;
        const float u = (x + bound) / (2.0f * bound);
        c[a] = min(max((int)floorf(u * (float)R), 0), R - 1);
    }
    return binary[((int64_t)c[0] * R + c[1]) * R + c[2]] != 0;
}

// Wave-per-ray marcher: one wavefront walks one ray, 64 consecutive steps per iteration (a lane = one step: setup once,
// one occupancy byte per lane in flight instead of a 350-iteration dependent-load loop per thread), ballot + prefix
// popcount compacts the occupied steps into the ray's slot row [cap]; 2 048 rays fill 2 048 waves (8 per CU) where the
// thread-per-ray form filled 32 workgroups.  A second launch copies the slot rows to their packed positions once the
// exclusive scan of the counts is known -- the march itself runs ONCE (a count pass + a fill pass would run it twice).
__global__ __launch_bounds__(256) void march_wave_kernel(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                         const float *__restrict__ jitter, int N, float step, float bound,
                                                         int R, const uint8_t *__restrict__ binary, int cap,
                                                         int32_t *__restrict__ ray_cnt, float *__restrict__ slot_ts,
                                                         float *__restrict__ slot_te, int32_t *__restrict__ overflow) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= N) return;
    const MarchRay m = march_setup(rays_o, rays_d, jitter, r, step, bound);
    float *row_s = slot_ts + (int64_t)r * cap, *row_e = slot_te + (int64_t)r * cap;
    int n = 0, k0 = 0;
    for (; k0 < cap; k0 += 64) {
        const int k = k0 + lane;
        const float ts = m.t0 + (float)k * step;
        const bool in = ts < m.tfar;
        if (__ballot(in) == 0ull) break;
        const float te = fminf(ts + step, m.tfar);
        const bool occ = in && march_occupied(m, ts, te, bound, R, binary);
        const unsigned long long mask = __ballot(occ);
        const int pos = n + __popcll(mask & ((1ull << lane) - 1ull));
        if (occ && pos < cap) {
            row_s[pos] = ts;
            row_e[pos] = te;
        }
        n += __popcll(mask);
    }
    if (lane == 0) {
        ray_cnt[r] = n < cap ? n : cap;
        // a ray with more steps than the slot row holds (directions much shorter than unit length): tell the host, which
        // re-runs the batch with a longer slot row
        if (n > cap || (m.t0 + (float)k0 * step) < m.tfar) atomicOr(overflow, 1);
    }
}

__global__ __launch_bounds__(256) void march_pack_kernel(const int32_t *__restrict__ ray_start, const int32_t *__restrict__ ray_cnt,
                                                         const float *__restrict__ slot_ts, const float *__restrict__ slot_te,
                                                         int N, int cap, int32_t *__restrict__ ray_idx,
                                                         float *__restrict__ t_starts, float *__restrict__ t_ends) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= N) return;
    const int n = ray_cnt[r];
    const int64_t base = ray_start[r];
    const float *row_s = slot_ts + (int64_t)r * cap, *row_e = slot_te + (int64_t)r * cap;
    for (int i = lane; i < n; i += 64) {
        ray_idx[base + i] = r;
        t_starts[base + i] = row_s[i];
        t_ends[base + i] = row_e[i];
    }
}

static Pose pose_from_c2w(const float *c2w_host) {
    Pose p;
    for (int a = 0; a < 3; a++) {
        for (int b = 0; b < 3; b++) p.r[a * 3 + b] = c2w_host[a * 4 + b];
        p.t[a] = c2w_host[a * 4 + 3];
    }
    return p;
}

extern "C" int mh_generate_rays(float fx, float fy, float cx, float cy, const float *c2w_host, int32_t H, int32_t W,
                                float *rays_o, float *rays_d, void *stream) {
    if (!c2w_host || !rays_o || !rays_d || H <= 0 || W <= 0) return MH_ERR_ARG;
    const Pose p = pose_from_c2w(c2w_host);
    const int n = H * W;
    hipLaunchKernelGGL(generate_rays_kernel, dim3((n + 255) / 256), dim3(256), 0, mh_stream(stream), fx, fy, cx, cy, p,
                       (int)H, (int)W, rays_o, rays_d);
    MH_CHECK_LAUNCH();
    return MH_OK;
}

extern "C" int mh_rays_sample_uniform(float fx, float fy, float cx, float cy, const float *c2w_host, int32_t H, int32_t W,
                                      const int32_t *pix, const float *jitter, int32_t N, int32_t S, float bound,
                                      float *rays_o, float *rays_d, int32_t *ray_idx, float *t_starts, float *t_ends,
                                      float *xyz, int32_t *ray_start, int32_t *ray_cnt, void *stream) {
    if (N == 0) return MH_OK;
    if (!c2w_host || !jitter || !rays_o || !rays_d || !ray_idx || !t_starts || !t_ends || !ray_start || !ray_cnt ||
        H <= 0 || W <= 0 || N < 0 || S <= 0)
        return MH_ERR_ARG;
    if (!pix && (int64_t)N > (int64_t)H * W) return MH_ERR_ARG;   // whole-image mode: ray r is pixel r
    const int64_t total = (int64_t)N * S;
    if (total > 0x7fffffffLL) return MH_ERR_ARG;
    const Pose p = pose_from_c2w(c2w_host);
    hipLaunchKernelGGL(rays_sample_uniform_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, mh_stream(stream),
                       fx, fy, cx, cy, p, (int)W, pix, jitter, (int)N, (int)S, bound, rays_o, rays_d, ray_idx, t_starts,
                       t_ends, xyz, ray_start, ray_cnt);
    MH_CHECK_LAUNCH();
    return MH_OK;
}

extern "C" int mh_sample_uniform(const float *rays_o, const float *rays_d, const float *jitter, int32_t N, int32_t S,
                                 float bound, int32_t *ray_idx, float *t_starts, float *t_ends, float *xyz,
                                 int32_t *ray_start, int32_t *ray_cnt, void *stream) {
    if (N == 0) return MH_OK;
    if (!rays_o || !rays_d || !jitter || !ray_idx || !t_starts || !t_ends || !ray_start || !ray_cnt || N < 0 || S <= 0)
        return MH_ERR_ARG;
    const int64_t total = (int64_t)N * S;
    if (total > 0x7fffffffLL) return MH_ERR_ARG;
    hipLaunchKernelGGL(sample_uniform_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, mh_stream(stream),
                       rays_o, rays_d, jitter, (int)N, (int)S, bound, ray_idx, t_starts, t_ends, xyz, ray_start, ray_cnt);
    MH_CHECK_LAUNCH();
    return MH_OK;
}

// upper bound of the steps one ray can take inside the AABB: its longest chord is the diagonal 2*bound*sqrt(3)
extern "C" int32_t mh_march_cap(float step, float bound) {
    if (!(step > 0.f) || !(bound > 0.f)) return 0;
    const double n = 2.0 * (double)bound * 1.7320508075688772 / (double)step;
    return (int32_t)n + 4;
}

extern "C" int mh_march_slots(const float *rays_o, const float *rays_d, const float *jitter, int32_t N, float step,
                              float bound, int32_t R, const uint8_t *binary, int32_t cap, int32_t *ray_cnt,
                              float *slot_ts, float *slot_te, int32_t *overflow, void *stream) {
    if (N == 0) return MH_OK;
    if (!rays_o || !rays_d || !binary || !ray_cnt || !slot_ts || !slot_te || !overflow || N < 0 || R <= 0 ||
        !(step > 0.f) || !(bound > 0.f) || cap <= 0)
        return MH_ERR_ARG;
    hipLaunchKernelGGL(march_wave_kernel, dim3((N + 3) / 4), dim3(256), 0, mh_stream(stream), rays_o, rays_d, jitter, (int)N,
                       step, bound, (int)R, binary, (int)cap, ray_cnt, slot_ts, slot_te, overflow);
    MH_CHECK_LAUNCH();
    return MH_OK;
}

extern "C" int mh_march_pack(const int32_t *ray_start, const int32_t *ray_cnt, const float *slot_ts, const float *slot_te,
                             int32_t N, int32_t cap, int32_t *ray_idx, float *t_starts, float *t_ends, void *stream) {
    if (N == 0) return MH_OK;
    if (!ray_start || !ray_cnt || !slot_ts || !slot_te || !ray_idx || !t_starts || !t_ends || N < 0 || cap <= 0)
        return MH_ERR_ARG;
    hipLaunchKernelGGL(march_pack_kernel, dim3((N + 3) / 4), dim3(256), 0, mh_stream(stream), ray_start, ray_cnt, slot_ts,
                       slot_te, (int)N, (int)cap, ray_idx, t_starts, t_ends);
    MH_CHECK_LAUNCH();
    return MH_OK;
}
