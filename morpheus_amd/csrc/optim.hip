// Adam over ONE flat fp32 parameter bucket: the step after the hot path.
//
// Replaces torch.optim.Adam(model.get_params_all(lr), betas=(0.9, 0.99), eps=1e-15) of the reference
// (morpheus.py:154-155) as it is stepped at morpheus.py:1401-1424: no weight decay, no amsgrad.  The reference's
// optimiser walks ~60 parameter tensors in 10 groups (the two 3.2 MB hash tables included); here every parameter,
// gradient and moment lives in one flat buffer (morpheus_amd/optim.py), so a step is one launch streaming
// 7 x 7.45 MB.  Groups are contiguous segments of the bucket; their learning rates travel by value.
//   m = m + (g - m)(1 - b1);  v = b2 v + (1 - b2) g g;  p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// -- the operation order of torch's fused Adam kernel, so the two agree to fp32 round-off.
#include "common.h"

#define ADAM_MAX_SEGS 160
struct AdamSegs {
    int n;
    int64_t end[ADAM_MAX_SEGS];
    float step_size[ADAM_MAX_SEGS];  // lr / (1 - b1^t_s);  < 0: the segment is skipped (its parameter had no gradient)
    float bc2_sqrt[ADAM_MAX_SEGS];   // sqrt(1 - b2^t_s)
};

// A segment is ONE parameter tensor (or a group's 4-element alignment pad): torch.optim.Adam keeps a step count per
// parameter and leaves a parameter whose gradient is None untouched -- no moment decay, no move, no step increment
// (the reference steps the pose group only on real-view iterations, morpheus.py:1399-1424 with freeze_lr) -- so the
// bias corrections and the skip flag travel per segment.
// dyn: NULL (step sizes / bias corrections travel by value in `segs`), or [2 * n_segs] DEVICE floats
// (step_size | bc2_sqrt) written by adam_steps_kernel from device-side flags and counters (mh_adam_step_dev).
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                   float *__restrict__ v, AdamSegs segs, float beta1, float beta2, float eps,
                                                   int64_t n, const float *__restrict__ dyn) {
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i0 >= n) return;
#define SEG_STEP(s) (dyn ? dyn[(s)] : segs.step_size[(s)])
#define SEG_BC2(s) (dyn ? dyn[segs.n + (s)] : segs.bc2_sqrt[(s)])
    const int cnt = (n - i0) < 4 ? (int)(n - i0) : 4;
    // first segment whose end is beyond i0 (binary search over <= 160 ends)
    int lo = 0, hi = segs.n - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (i0 >= segs.end[mid]) lo = mid + 1; else hi = mid;
    }
    int seg = lo;
    const bool one_seg = (i0 + cnt) <= segs.end[seg];
    if (one_seg && SEG_STEP(seg) < 0.f) return;             // skipped parameter: nothing is read or written
    float pv[4], gv[4], mv[4], vv[4];
    if (cnt == 4) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(p + i0), b = *reinterpret_cast<const f32x4 *>(g + i0);
        const f32x4 c = *reinterpret_cast<const f32x4 *>(m + i0), d = *reinterpret_cast<const f32x4 *>(v + i0);
#pragma unroll
        for (int k = 0; k < 4; k++) pv[k] = a[k], gv[k] = b[k], mv[k] = c[k], vv[k] = d[k];
    } else {
        for (int k = 0; k < cnt; k++) pv[k] = p[i0 + k], gv[k] = g[i0 + k], mv[k] = m[i0 + k], vv[k] = v[i0 + k];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (k < cnt) {
            while (seg < segs.n - 1 && i0 + k >= segs.end[seg]) seg++;
            const float ss = SEG_STEP(seg);
            if (ss >= 0.f) {
                mv[k] = mv[k] + (gv[k] - mv[k]) * (1.0f - beta1);
                vv[k] = beta2 * vv[k] + (1.0f - beta2) * gv[k] * gv[k];
                const float denom = sqrtf(vv[k]) / SEG_BC2(seg) + eps;
                pv[k] -= ss * mv[k] / denom;
            }
        }
    }
    if (cnt == 4) {
        f32x4 a, c, d;
#pragma unroll
        for (int k = 0; k < 4; k++) a[k] = pv[k], c[k] = mv[k], d[k] = vv[k];
        *reinterpret_cast<f32x4 *>(p + i0) = a;
        *reinterpret_cast<f32x4 *>(m + i0) = c;
        *reinterpret_cast<f32x4 *>(v + i0) = d;
    } else {
        for (int k = 0; k < cnt; k++) p[i0 + k] = pv[k], m[i0 + k] = mv[k], v[i0 + k] = vv[k];
    }
#undef SEG_STEP
#undef SEG_BC2
}

// Device-side bookkeeping of a data-parallel step: whether a parameter "has a gradient" is then a property of ALL ranks (the
// all-reduced has-gradient flags of dist.GradBucket), known on the device only -- reading it back would cost the host a
// synchronisation per step.  One thread per segment: flag > 0 -> the segment's step count goes up and its step size /
// second-moment correction are written for adam_kernel; otherwise step size -1 (skipped: value, moments, count untouched).
struct AdamLrs {
    float lr[ADAM_MAX_SEGS];
};
__global__ void adam_steps_kernel(const float *__restrict__ flag, int64_t *__restrict__ step, float *__restrict__ dyn, AdamLrs lrs,
                                  int n_segs, float beta1, float beta2) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_segs) return;
    if (flag[s] > 0.0f) {
        const int64_t t = step[s] + 1;
        step[s] = t;
        dyn[s] = (float)((double)lrs.lr[s] / (1.0 - pow((double)beta1, (double)t)));
        dyn[n_segs + s] = (float)sqrt(1.0 - pow((double)beta2, (double)t));
    } else {
        dyn[s] = -1.0f;
        dyn[n_segs + s] = 1.0f;
    }
}

extern "C" int mh_adam_step(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n, int32_t n_segs,
                            const int64_t *seg_end_host, const float *seg_lr_host, const int64_t *seg_step_host, float beta1,
                            float beta2, float eps, void *stream) {
    if (n == 0) return MH_OK;
    if (!params || !grads || !exp_avg || !exp_avg_sq || n < 0 || n_segs <= 0 || n_segs > ADAM_MAX_SEGS || !seg_end_host ||
        !seg_lr_host || !seg_step_host || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f))
        return MH_ERR_ARG;
    if ((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) != 0) return MH_ERR_ARG;
    AdamSegs segs;
    segs.n = n_segs;
    int64_t prev = 0;
    for (int s = 0; s < n_segs; s++) {
        if (seg_end_host[s] < prev || seg_end_host[s] > n || seg_step_host[s] < 0) return MH_ERR_ARG;
        prev = segs.end[s] = seg_end_host[s];
        if (seg_step_host[s] == 0) {            // step 0: no gradient for this parameter this time -> untouched
            segs.step_size[s] = -1.0f;
            segs.bc2_sqrt[s] = 1.0f;
        } else {
            const double bc1 = 1.0 - pow((double)beta1, (double)seg_step_host[s]);
            const double bc2 = 1.0 - pow((double)beta2, (double)seg_step_host[s]);
            segs.step_size[s] = (float)((double)seg_lr_host[s] / bc1);
            segs.bc2_sqrt[s] = (float)sqrt(bc2);
        }
    }
    if (prev != n) return MH_ERR_ARG;
    const int64_t threads = (n + 3) / 4;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, mh_stream(stream), params, grads,
                       exp_avg, exp_avg_sq, segs, beta1, beta2, eps, n, (const float *)nullptr);
    MH_CHECK_LAUNCH();
    return MH_OK;
}

extern "C" int mh_adam_step_dev(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n, int32_t n_segs,
                                const int64_t *seg_end_host, const float *seg_lr_host, const float *seg_flag_dev,
                                int64_t *seg_step_dev, float *seg_scratch_dev, float beta1, float beta2, float eps, void *stream) {
    if (n == 0) return MH_OK;
    if (!params || !grads || !exp_avg || !exp_avg_sq || n < 0 || n_segs <= 0 || n_segs > ADAM_MAX_SEGS || !seg_end_host ||
        !seg_lr_host || !seg_flag_dev || !seg_step_dev || !seg_scratch_dev || !(beta1 >= 0.f && beta1 < 1.f) ||
        !(beta2 >= 0.f && beta2 < 1.f))
        return MH_ERR_ARG;
    if ((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) != 0) return MH_ERR_ARG;
    AdamSegs segs;
    AdamLrs lrs;
    segs.n = n_segs;
    int64_t prev = 0;
    for (int s = 0; s < n_segs; s++) {
        if (seg_end_host[s] < prev || seg_end_host[s] > n) return MH_ERR_ARG;
        prev = segs.end[s] = seg_end_host[s];
        segs.step_size[s] = -1.0f;      // unused: the kernel reads the device values
        segs.bc2_sqrt[s] = 1.0f;
        lrs.lr[s] = seg_lr_host[s];
    }
    if (prev != n) return MH_ERR_ARG;
    hipLaunchKernelGGL(adam_steps_kernel, dim3(1), dim3(ADAM_MAX_SEGS), 0, mh_stream(stream), seg_flag_dev, seg_step_dev,
                       seg_scratch_dev, lrs, (int)n_segs, beta1, beta2);
    MH_CHECK_LAUNCH();
    const int64_t threads = (n + 3) / 4;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, mh_stream(stream), params, grads,
                       exp_avg, exp_avg_sq, segs, beta1, beta2, eps, n, (const float *)seg_scratch_dev);
    MH_CHECK_LAUNCH();
    return MH_OK;
}
