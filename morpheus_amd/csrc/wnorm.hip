// Weight-norm parametrisation of ALL the hot path's weight-normed layers in one launch each way.
//
// Reference: models/decoders.py:51-52 wraps every Linear of deform_net / topo_net / color_net in
// nn.utils.weight_norm, i.e. W = g * v / ||v||_row (torch._weight_norm, dim=0), recomputed on every forward and
// differentiated on every backward: 15 layers x (norm, divide, multiply) + their autograd graph = ~200 tiny launches
// per training step around the MLP kernels.  Here a wavefront owns one weight row (<= 128 columns... any width works):
//   fwd:  W[r,:]  = v[r,:] * (g[r] / ||v[r,:]||)
//   bwd:  dg[r]   = <dW[r,:], v[r,:]> / ||v[r,:]||
//         dv[r,:] = (g[r] / ||v[r,:]||) * (dW[r,:] - v[r,:] * <dW[r,:], v[r,:]> / ||v[r,:]||^2)
// (the formulas of torch's _weight_norm_interface / _backward).  Layer descriptors travel by value.
#include "common.h"

#define WN_MAX_LAYERS 32
struct WnLayers {
    int n;
    int row_end[WN_MAX_LAYERS];   // exclusive prefix of rows
    int cols[WN_MAX_LAYERS];
    const float *v[WN_MAX_LAYERS];
    const float *g[WN_MAX_LAYERS];
    const float *dw[WN_MAX_LAYERS];  // bwd only (NULL: no gradient reached this layer)
    float *out0[WN_MAX_LAYERS];      // fwd: W      bwd: dv
    float *out1[WN_MAX_LAYERS];      // fwd: unused bwd: dg
};

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

template <bool BWD>
__global__ __launch_bounds__(256) void wn_kernel(WnLayers L) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= L.row_end[L.n - 1]) return;
    int l = 0;
    while (row >= L.row_end[l]) l++;
    const int r = row - (l ? L.row_end[l - 1] : 0);
    const int C = L.cols[l];
    const float *v = L.v[l] + (int64_t)r * C;
    const float gr = L.g[l][r];
    float ss = 0.f, dot = 0.f;
    const float *dw = BWD && L.dw[l] ? L.dw[l] + (int64_t)r * C : nullptr;
    for (int c = lane; c < C; c += 64) {
        const float x = v[c];
        ss += x * x;
        if (dw) dot += dw[c] * x;
    }
    ss = wave_sum(ss);
    const float norm = sqrtf(ss);
    if (!BWD) {
        const float s = gr / norm;
        float *w = L.out0[l] + (int64_t)r * C;
        for (int c = lane; c < C; c += 64) w[c] = v[c] * s;
    } else {
        dot = wave_sum(dot);
        float *dv = L.out0[l] + (int64_t)r * C;
        const float s = gr / norm, k = dot / (norm * norm);
        for (int c = lane; c < C; c += 64) dv[c] = dw ? s * (dw[c] - v[c] * k) : 0.f;
        if (lane == 0) L.out1[l][r] = dw ? dot / norm : 0.f;
    }
}

static int wn_fill(WnLayers &L, int32_t n_layers, const int32_t *rows_host, const int32_t *cols_host) {
    if (n_layers <= 0 || n_layers > WN_MAX_LAYERS || !rows_host || !cols_host) return MH_ERR_ARG;
    L.n = n_layers;
    int acc = 0;
    for (int l = 0; l < n_layers; l++) {
        if (rows_host[l] <= 0 || cols_host[l] <= 0) return MH_ERR_ARG;
        acc += rows_host[l];
        L.row_end[l] = acc;
        L.cols[l] = cols_host[l];
    }
    return MH_OK;
}

extern "C" int mh_weight_norm_fwd(int32_t n_layers, const float *const *v_host, const float *const *g_host, float *const *w_host,
                                  const int32_t *rows_host, const int32_t *cols_host, void *stream) {
    if (n_layers == 0) return MH_OK;
    WnLayers L;
    if (!v_host || !g_host || !w_host || wn_fill(L, n_layers, rows_host, cols_host) != MH_OK) return MH_ERR_ARG;
    for (int l = 0; l < n_layers; l++) {
        if (!v_host[l] || !g_host[l] || !w_host[l]) return MH_ERR_ARG;
        L.v[l] = v_host[l], L.g[l] = g_host[l], L.out0[l] = w_host[l], L.dw[l] = nullptr, L.out1[l] = nullptr;
    }
    hipLaunchKernelGGL(wn_kernel<false>, dim3((L.row_end[n_layers - 1] + 3) / 4), dim3(256), 0, mh_stream(stream), L);
    MH_CHECK_LAUNCH();
    return MH_OK;
}

extern "C" int mh_weight_norm_bwd(int32_t n_layers, const float *const *v_host, const float *const *g_host,
                                  const float *const *dw_host, float *const *dv_host, float *const *dg_host,
                                  const int32_t *rows_host, const int32_t *cols_host, void *stream) {
    if (n_layers == 0) return MH_OK;
    WnLayers L;
    if (!v_host || !g_host || !dw_host || !dv_host || !dg_host || wn_fill(L, n_layers, rows_host, cols_host) != MH_OK)
        return MH_ERR_ARG;
    for (int l = 0; l < n_layers; l++) {
        if (!v_host[l] || !g_host[l] || !dv_host[l] || !dg_host[l]) return MH_ERR_ARG;
        L.v[l] = v_host[l], L.g[l] = g_host[l], L.dw[l] = dw_host[l], L.out0[l] = dv_host[l], L.out1[l] = dg_host[l];
    }
    hipLaunchKernelGGL(wn_kernel<true>, dim3((L.row_end[n_layers - 1] + 3) / 4), dim3(256), 0, mh_stream(stream), L);
    MH_CHECK_LAUNCH();
    return MH_OK;
}
