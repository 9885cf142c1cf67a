// Fused tiny-MLP evaluators on exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) for gfx950.
//
// Replaces, for the render_rays hot path, what the reference runs as ~150 separate PyTorch ops:
//   warp            models/model.py:412-437  (freq-enc + deform code -> deform_net, topo_net)
//   get_sigma_albedo models/model.py:273-307 (freq-enc + hash + topo -> sdf_net -> Laplace sigma;
//                                             hash_c + geo -> color_net -> sigmoid)
//   MLP.forward     models/decoders.py:59-64 ; FreqEncoder_torch models/encodings.py:35-57 ;
//   LaplaceDensity  models/density.py:22-31
//
// Design (CDNA4-first):
//  * A wavefront owns a tile of 32 sample points for the whole network.  With the 32x32x2 fp32
//    MFMA the accumulator layout is  D[row = (r&3)+8(r>>2)+4(lane>>5)][col = lane&31]  and the B
//    operand of the NEXT layer wants  B[k = lane>>5][col = lane&31]: choosing the k-pairing
//    (row, row+4) makes every accumulator register directly the next layer's B operand -- the
//    activations never leave the register file between layers (no LDS transpose, no HBM round
//    trip; the reference materialises [M,128] per layer).  The matching k-permutation is baked
//    into the host-side packing of the weight (A) operands (morpheus_amd/packing.py).
//  * Weights are staged per layer into LDS once per 128-point block in A-fragment order, read
//    back with conflict-free ds_read_b128 (one read feeds 4 MFMAs per output tile).
//  * fp32 in / fp32 accumulate: bitwise a k-ordered fmaf chain, so parity with the PyTorch
//    reference is round-off only (summation order differs).
//  * Forward (training) parks post-ReLU activations per 32-point tile feature-major [F][32] --
//    the layout the weight-gradient MFMA wants (K = points) -- instead of recomputing the
//    forward in backward: 128 MACs per stored float makes the store cheaper than the recompute.
//  * Backward is two kernels: backward-data (same register-resident chain with transposed packs,
//    ReLU derivative from sign masks the forward parks next to the activations, writes dPre tiles) and
//    mh_mlp_wgrad (dW = dPre . act^T as MFMA over the point axis, per-chunk partials summed by a
//    reduction launch; no atomics anywhere).
//  * The kernels are ISSUE-bound: on gfx950 one wave's VALU / store / DMA instructions do not issue under
//    the co-resident wave's fp32 MFMAs (tools/phase_trace.py), so every non-MFMA instruction of the layer
//    loop is paid in full and the design effort goes into having few of them.
#include "mlp_dev.h"
#include <stdlib.h>


__shared__ f32x4 lds_w[4096];  // 64 KB: one layer's A fragments  [mt][q][lane] float4

// Weight staging by LDS-DMA (global_load_lds_dwordx4): the packed layout is lane-linear ([tile][quad][lane] float4),
// exactly the "wave-uniform LDS base + lane*16" destination the instruction wants, so a layer is N_F4/256 DMA
// instructions per thread, all in flight together, with no VGPR round trip and no ds_write pass.  N_F4 is a
// compile-time constant: with a run-time trip count hipcc emits load -> s_waitcnt vmcnt(0) -> ds_write per
// iteration, i.e. 16 serial L2 round trips per layer -- as long as the layer's whole MFMA phase (seen in the .s).
template <int N_F4>
__device__ __forceinline__ void stage_weights(const float *__restrict__ g) {
    static_assert(N_F4 % 256 == 0, "whole rounds of the 256-thread block");
    __syncthreads();  // everyone is done reading the previous layer
    const f32x4 *src = reinterpret_cast<const f32x4 *>(g);
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < N_F4 / 256; k++)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + k * 256 + threadIdx.x),
                                         (__attribute__((address_space(3))) void *)(lds_w + k * 256 + wave * 64), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// Split form of stage_weights for a layer pipeline: stage_issue<N>() right after the barrier that ends layer l's MFMA
// phase (every wave is done reading lds_w) starts layer l+1's DMA, the layer-l epilogue (ReLU, tile stores) runs under its
// latency, stage_wait() then covers the DMA and the epilogue's stores together (vmcnt counts both on gfx9).
template <int N_F4>
__device__ __forceinline__ void stage_issue(const float *__restrict__ g) {
    static_assert(N_F4 % 256 == 0, "whole rounds of the 256-thread block");
    const f32x4 *src = reinterpret_cast<const f32x4 *>(g);
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < N_F4 / 256; k++)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + k * 256 + threadIdx.x),
                                         (__attribute__((address_space(3))) void *)(lds_w + k * 256 + wave * 64), 16, 0, 0);
}
__device__ __forceinline__ void stage_wait() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

#ifdef MH_PHASE_TRACE
// phase trace for tools/phase_trace.py (never compiled into the product library): wave 0 of every 64th workgroup
// stamps s_memtime at the phase boundaries of warp_fwd_kernel
__device__ long long mh_trace[256 * 64];
#define MH_STAMP(slot)                                                                     \
    do {                                                                                   \
        if ((threadIdx.x == 0) && (blockIdx.x % 64 == 0) && (blockIdx.x / 64 < 256))       \
            mh_trace[(blockIdx.x / 64) * 64 + (slot)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
// constant-rate (100 MHz) timestamp next to a shader-clock one: slot 62 at the first stamp, 63 at the last -- the ratio of
// the two spans is the EFFECTIVE SHADER CLOCK the kernel ran at (the chip clocks to its power budget)
#define MH_STAMP_REAL(slot)                                                                \
    do {                                                                                   \
        if ((threadIdx.x == 0) && (blockIdx.x % 64 == 0) && (blockIdx.x / 64 < 256))       \
            mh_trace[(blockIdx.x / 64) * 64 + (slot)] = (long long)__builtin_amdgcn_s_memrealtime(); \
    } while (0)
extern "C" int mh_trace_read(long long *dst_host) {
    return hipMemcpyFromSymbol(dst_host, HIP_SYMBOL(mh_trace), sizeof(long long) * 256 * 64) == hipSuccess ? 0 : 2;
}
#else
#define MH_STAMP(slot) do { } while (0)
#define MH_STAMP_REAL(slot) do { } while (0)
#endif


// acc[mt] += W_tile[mt] . bin   (KS k-steps of 2)
template <int KS, int MT>
__device__ __forceinline__ void mfma_layer(const float (&bin)[KS], f32x16 (&acc)[MT], int lane) {
    static_assert(KS % 4 == 0, "k-steps come in quads");
#pragma unroll
    for (int q = 0; q < KS / 4; q++) {
        f32x4 a[MT];
#pragma unroll
        for (int t = 0; t < MT; t++) a[t] = lds_w[(t * (KS / 4) + q) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int t = 0; t < MT; t++)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][j], bin[4 * q + j], acc[t], 0, 0, 0);
    }
}

// acc[mt] = W_tile[mt] . bin, accumulators NOT pre-initialised: the first k-step's MFMA takes the inline constant 0 as its
// C operand instead of 16 * MT v_mov instructions zeroing the accumulators beforehand (backward-data layers have no bias)
template <int KS, int MT>
__device__ __forceinline__ void mfma_layer_z(const f32x4 *__restrict__ w, const float (&bin)[KS], f32x16 (&acc)[MT], int lane) {
    static_assert(KS % 4 == 0, "k-steps come in quads");
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < KS / 4; q++) {
        f32x4 a[MT];
#pragma unroll
        for (int t = 0; t < MT; t++) a[t] = w[(t * (KS / 4) + q) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int t = 0; t < MT; t++)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][j], bin[4 * q + j], (q == 0 && j == 0) ? zero : acc[t], 0, 0, 0);
    }
}

// Same contraction with the layer's A fragments at an arbitrary LDS address (resident-weight kernels below).
template <int KS, int MT>
__device__ __forceinline__ void mfma_layer_at(const f32x4 *__restrict__ w, const float (&bin)[KS], f32x16 (&acc)[MT],
                                              int lane) {
    static_assert(KS % 4 == 0, "k-steps come in quads");
#pragma unroll
    for (int q = 0; q < KS / 4; q++) {
        f32x4 a[MT];
#pragma unroll
        for (int t = 0; t < MT; t++) a[t] = w[(t * (KS / 4) + q) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int t = 0; t < MT; t++)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][j], bin[4 * q + j], acc[t], 0, 0, 0);
    }
}

// The field networks are small enough (94 KB fwd / 98 KB transposed) to keep EVERY layer's fragments in LDS for the
// life of a persistent block: one staging pass per block, then no barrier at all between layers or tiles.
#define FIELD_THREADS 512
extern __shared__ f32x4 lds_res[];
template <int N_F4>
__device__ __forceinline__ void stage_resident(const float *__restrict__ g, int dst_f4) {
    const f32x4 *src = reinterpret_cast<const f32x4 *>(g);
    constexpr int R = (N_F4 + FIELD_THREADS - 1) / FIELD_THREADS;
    f32x4 v[R];
#pragma unroll
    for (int k = 0; k < R; k++) {
        const int i = k * FIELD_THREADS + threadIdx.x;
        if (i < N_F4) v[k] = src[i];
    }
#pragma unroll
    for (int k = 0; k < R; k++) {
        const int i = k * FIELD_THREADS + threadIdx.x;
        if (i < N_F4) lds_res[dst_f4 + i] = v[k];
    }
}


// =====================================================================================
// warp: deform_net + topo_net
// =====================================================================================
__global__ __launch_bounds__(256, 2) void warp_fwd_kernel(const float *__restrict__ x, const int32_t *__restrict__ slot,
                                                          const float *__restrict__ bias0_d, const float *__restrict__ bias0_t,
                                                          const float *__restrict__ wpack_d, const float *__restrict__ wpack_t,
                                                          const float *__restrict__ bias_d, const float *__restrict__ bias_t,
                                                          int n_bands, float *__restrict__ out_deform,
                                                          float *__restrict__ out_topo, float *__restrict__ acts, int64_t M) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pt = lane & 31, h = lane >> 5;
    const int64_t tile_id = (int64_t)blockIdx.x * 4 + wave;
    const int64_t p = tile_id * TILE + pt;
    const int64_t pc = p < M ? p : M - 1;
    float xv[3] = {x[pc * 3 + 0], x[pc * 3 + 1], x[pc * 3 + 2]};
    const int sl = slot ? slot[pc] : 0;
    float *tile = acts ? acts + tile_id * (int64_t)(WARP_ACT_ROWS * TILE) : nullptr;

    float bin0[20];
    enc_bin(xv, h, n_bands, bin0);
    if (tile) {
        store_kk_rows<20>(tile, bin0, pt, h);
#pragma unroll
        for (int k = 20; k < 32; k++) tile[(2 * k + h) * TILE + pt] = 0.f;  // pad rows 40..63
    }
    uint2 *mk = tile ? reinterpret_cast<uint2 *>(tile + WARP_HID_ROWS * TILE) : nullptr;
    MH_STAMP(0);
    MH_STAMP_REAL(62);
    __syncthreads();
    stage_issue<1280>(wpack_d);
    for (int net = 0; net < 2; net++) {
        const float *wp = net ? wpack_t : wpack_d;
        const float *bs = net ? bias_t : bias_d;
        const float *b0 = (net ? bias0_t : bias0_d) + (int64_t)sl * 128;
        float *ht = tile ? tile + (64 + net * 640) * TILE : nullptr;
        f32x16 acc[4];
        float bin[64];
        // layer 0: 40 -> 128, bias row chosen by the point's frame slot
        acc_bias<4>(acc, b0, h);   // bias loads are issued BEFORE the wait: their latency hides under the DMA's
        stage_wait();
        MH_STAMP(1 + (net * 6 + 0) * 5 + 0);
        mfma_layer<20, 4>(bin0, acc, lane);
        MH_STAMP(1 + (net * 6 + 0) * 5 + 1);
        __syncthreads();
        MH_STAMP(1 + (net * 6 + 0) * 5 + 2);
        wp += 5120;
        stage_issue<4096>(wp);
        MH_STAMP(1 + (net * 6 + 0) * 5 + 3);
        acc_to_bin<4, true>(acc, bin);
        if (ht) {
            store_acc_rows<4>(ht, bin, pt, h);
            mk[(net * 5 + 0) * 64 + lane] = relu_mask64(bin);
        }
        MH_STAMP(1 + (net * 6 + 0) * 5 + 4);
        // layers 1..4: 128 -> 128
        for (int l = 1; l <= 4; l++) {
            acc_bias<4>(acc, bs + (l - 1) * 128, h);
            stage_wait();
            MH_STAMP(1 + (net * 6 + l) * 5 + 0);
            mfma_layer<64, 4>(bin, acc, lane);
            MH_STAMP(1 + (net * 6 + l) * 5 + 1);
            __syncthreads();
            MH_STAMP(1 + (net * 6 + l) * 5 + 2);
            wp += 16384;
            if (l < 4)
                stage_issue<4096>(wp);
            else
                stage_issue<1024>(wp);
            MH_STAMP(1 + (net * 6 + l) * 5 + 3);
            acc_to_bin<4, true>(acc, bin);
            if (ht) {
                store_acc_rows<4>(ht + l * 128 * TILE, bin, pt, h);
                mk[(net * 5 + l) * 64 + lane] = relu_mask64(bin);
            }
            MH_STAMP(1 + (net * 6 + l) * 5 + 4);
        }
        // layer 5: 128 -> 3 | 2 (one padded tile)
        f32x16 o[1];
        acc_bias<1>(o, bs + 4 * 128, h);
        stage_wait();
        MH_STAMP(1 + (net * 6 + 5) * 5 + 0);
        mfma_layer<64, 1>(bin, o, lane);
        MH_STAMP(1 + (net * 6 + 5) * 5 + 1);
        __syncthreads();
        MH_STAMP(1 + (net * 6 + 5) * 5 + 2);
        if (net == 0) stage_issue<1280>(wpack_t);
        if (h == 0 && p < M) {
            if (net == 0) {
                out_deform[p * 3 + 0] = o[0][0];
                out_deform[p * 3 + 1] = o[0][1];
                out_deform[p * 3 + 2] = o[0][2];
            } else {
                out_topo[p * 2 + 0] = o[0][0];
                out_topo[p * 2 + 1] = o[0][1];
            }
        }
        MH_STAMP(1 + (net * 6 + 5) * 5 + 4);
        if (net == 1) MH_STAMP_REAL(63);
    }
}

__global__ __launch_bounds__(256, 2) void warp_bwd_kernel(const float *__restrict__ x, const float *__restrict__ g_deform,
                                                          const float *__restrict__ g_topo, const float *__restrict__ wpackT_d,
                                                          const float *__restrict__ wpackT_t, int n_bands,
                                                          const float *__restrict__ acts, float *__restrict__ dpre,
                                                          float *__restrict__ g_x, int64_t M) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pt = lane & 31, h = lane >> 5;
    const int64_t tile_id = (int64_t)blockIdx.x * 4 + wave;
    const int64_t p = tile_id * TILE + pt;
    const bool live = p < M;
    const int64_t pc = live ? p : M - 1;
    float xv[3] = {x[pc * 3 + 0], x[pc * 3 + 1], x[pc * 3 + 2]};
    const float *atile = acts + tile_id * (int64_t)(WARP_ACT_ROWS * TILE);
    float *dtile = dpre + tile_id * (int64_t)(WARP_DPRE_ROWS * TILE);
    float gx[3] = {0.f, 0.f, 0.f};

    for (int net = 0; net < 2; net++) {
        const float *wt = net ? wpackT_t : wpackT_d;
        const float *g = net ? g_topo : g_deform;
        const int nout = net ? 2 : 3;
        float *dt = dtile + net * 672 * TILE;
        // ReLU masks of the net's five hidden layers: 8 bytes per lane and layer, all fetched up front
        const uint2 *mk = reinterpret_cast<const uint2 *>(atile + WARP_HID_ROWS * TILE) + net * 5 * 64 + lane;
        uint2 msk[5];
#pragma unroll
        for (int l = 0; l < 5; l++) msk[l] = mk[l * 64];
        // dPre5: rows 0..nout-1 carry the incoming gradient (no activation on the last layer)
        float d5[16];
#pragma unroll
        for (int r = 0; r < 16; r++) d5[r] = 0.f;
        if (g && live && h == 0) {
            d5[0] = g[p * nout + 0];
            d5[1] = g[p * nout + 1];
            if (nout == 3) d5[2] = g[p * nout + 2];
        }
        store_acc_rows<1>(dt + 640 * TILE, d5, pt, h);
        // dH5 = W5^T dPre5
        f32x16 acc[4];
        float dbin[64];
        stage_weights<1024>(wt);
        mfma_layer_z<16, 4>(lds_w, d5, acc, lane);
        wt += 4096;
        for (int l = 4; l >= 0; l--) {
            // output of layer l is H_{l+1}; mask by its ReLU (sign bits parked by the forward kernel) and park dPre_l
            const uint2 m = msk[l];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const uint32_t mw = (t < 2 ? m.x : m.y) >> (16 * (t & 1));
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float v = mask_bit(mw, r, acc[t][r]);
                    dbin[16 * t + r] = v;
                    dt[(l * 128 + 32 * t + acc_row(r, h)) * TILE + pt] = v;
                }
            }
            if (l > 0) {
                stage_weights<4096>(wt);
                mfma_layer_z<64, 4>(lds_w, dbin, acc, lane);
                wt += 16384;
            } else if (g_x) {
                // d(enc features) = W0^T dPre0; rows ordered (kk = 16t + r, h = lane>>5).  Skipped when nobody asks
                // for d/dx (sample positions without gradient: every step that does not optimise the camera pose) --
                // 128 of the net's 1216 MFMAs
                stage_weights<2048>(wt);
                f32x16 e[2];
                mfma_layer_z<64, 2>(lds_w, dbin, e, lane);
                // the encoding derivatives come from the parked encoding (rows 0..35 of the tile), fetched here, once per
                // net, instead of living in registers across the whole layer chain (the kernel sits at the 256-VGPR limit)
                float dsc[18];
                enc_deriv_parked(atile, pt, h, dsc);
#pragma unroll
                for (int k = 0; k < 18; k++) {
                    const float de = k < 16 ? e[0][k] : e[1][k - 16];
                    gx[k % 3] += de * dsc[k];
                }
                // kk 18: (x0 | x1), kk 19: (x2 | -)
                if (h == 0) {
                    gx[0] += e[1][2];
                    gx[2] += e[1][3];
                } else {
                    gx[1] += e[1][2];
                }
            }
        }
    }
#pragma unroll
    for (int d = 0; d < 3; d++) gx[d] += __shfl_xor(gx[d], 32);
    if (g_x && live && h == 0) {
        g_x[p * 3 + 0] = gx[0];
        g_x[p * 3 + 1] = gx[1];
        g_x[p * 3 + 2] = gx[2];
    }
}

// =====================================================================================
// canonical field: sdf_net (+ Laplace density) and color_net
// =====================================================================================
__device__ __forceinline__ float laplace_sigma(float s, float beta) {
    // density.py:22-31, cancellation-free form (mlp_dev.h: laplace_unit)
    return (1.0f / beta) * laplace_unit(s, beta);
}

__global__ __launch_bounds__(FIELD_THREADS, 1) void field_fwd_kernel(const float *__restrict__ xc, const float *__restrict__ feat_s,
                                                           const float *__restrict__ feat_c, const float *__restrict__ topo,
                                                           const float *__restrict__ wpack, const float *__restrict__ bias,
                                                           const float *__restrict__ beta_p, int n_bands, int with_color, float *__restrict__ sdf,
                                                           float *__restrict__ sigma, float *__restrict__ albedo,
                                                           float *__restrict__ acts, int64_t M, int64_t n_tiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pt = lane & 31, h = lane >> 5;
    if (with_color)
        stage_resident<FIELD_WPACK / 4>(wpack, 0);
    else
        stage_resident<(5120 + 2 * 4096) / 4>(wpack, 0);
    __syncthreads();
    for (int64_t tile_id = (int64_t)blockIdx.x * (FIELD_THREADS / 64) + wave; tile_id < n_tiles;
         tile_id += (int64_t)gridDim.x * (FIELD_THREADS / 64)) {
    const int64_t p = tile_id * TILE + pt;
    const bool live = p < M;
    const int64_t pc = live ? p : M - 1;
    float xv[3] = {xc[pc * 3 + 0], xc[pc * 3 + 1], xc[pc * 3 + 2]};
    float *tile = acts ? acts + tile_id * (int64_t)(FIELD_ACT_ROWS * TILE) : nullptr;

    float bin0[40];
    enc_bin(xv, h, n_bands, bin0);
    {
        const f32x4 *fs = reinterpret_cast<const f32x4 *>(feat_s + pc * 32 + 16 * h);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 v = fs[q];
#pragma unroll
            for (int c = 0; c < 4; c++) bin0[20 + 4 * q + c] = v[c];
        }
    }
    bin0[36] = topo ? topo[pc * 2 + h] : 0.f;
    bin0[37] = bin0[38] = bin0[39] = 0.f;
    if (tile) {
        store_kk_rows<40>(tile, bin0, pt, h);      // rows 80..95 (padding of the 96-row k extent) are neither written nor read
    }
    const f32x4 *wp = lds_res;
    f32x16 acc[2];
    float bin[32];
    // sdf L0: 73 -> 64
    acc_bias<2>(acc, bias, h);
    mfma_layer_at<40, 2>(wp, bin0, acc, lane);
    acc_to_bin<2, true>(acc, bin);
    if (tile) {
        store_acc_rows<2>(tile + 96 * TILE, bin, pt, h);
        reinterpret_cast<uint32_t *>(tile + FIELD_HID_ROWS * TILE)[0 * 64 + lane] = relu_mask32(bin);
    }
    wp += 1280;
    // sdf L1: 64 -> 64
    acc_bias<2>(acc, bias + 64, h);
    mfma_layer_at<32, 2>(wp, bin, acc, lane);
    acc_to_bin<2, true>(acc, bin);
    if (tile) {
        store_acc_rows<2>(tile + 160 * TILE, bin, pt, h);
        reinterpret_cast<uint32_t *>(tile + FIELD_HID_ROWS * TILE)[1 * 64 + lane] = relu_mask32(bin);
    }
    wp += 1024;
    // sdf L2: 64 -> [geo(32) | sdf], no activation
    if (!with_color) {
        // sdf-only pass (finite-difference taps): the geo tile feeds nothing and of the other tile only row 0, the sdf, is wanted:
        // a 64-term dot product per point -- 32 fused multiply-adds per lane on the k-steps it holds (row 0's A-fragment values sit
        // in lane 32 h of tile 1: the same LDS address for every lane of a half) instead of 32 MFMAs of 64 cycles
        float sv = 0.f;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const f32x4 w4 = wp[(8 + q) * 64 + 32 * h];
#pragma unroll
            for (int j = 0; j < 4; j++) sv = fmaf(w4[j], bin[4 * q + j], sv);
        }
        sv += __shfl_xor(sv, 32);
        sv += bias[128 + 32];
        if (h == 0 && live) {
            sdf[p] = sv;
            if (sigma) sigma[p] = laplace_sigma(sv, *beta_p);
        }
        continue;
    }
    acc_bias<2>(acc, bias + 128, h);
    mfma_layer_at<32, 2>(wp, bin, acc, lane);
    wp += 1024;
    if (h == 0 && live) {
        const float s = acc[1][0];
        sdf[p] = s;
        if (sigma) sigma[p] = laplace_sigma(s, *beta_p);
    }
    // color L0: [hash_c(32) | geo(32)] -> 64
    float binc[32];
    {
        const f32x4 *fc = reinterpret_cast<const f32x4 *>(feat_c + pc * 32 + 16 * h);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 v = fc[q];
#pragma unroll
            for (int c = 0; c < 4; c++) binc[4 * q + c] = v[c];
        }
#pragma unroll
        for (int r = 0; r < 16; r++) binc[16 + r] = acc[0][r];
    }
    if (tile) store_kk_rows<32>(tile + 224 * TILE, binc, pt, h);
    acc_bias<2>(acc, bias + 192, h);
    mfma_layer_at<32, 2>(wp, binc, acc, lane);
    acc_to_bin<2, true>(acc, bin);
    if (tile) {
        store_acc_rows<2>(tile + 288 * TILE, bin, pt, h);
        reinterpret_cast<uint32_t *>(tile + FIELD_HID_ROWS * TILE)[2 * 64 + lane] = relu_mask32(bin);
    }
    wp += 1024;
    // color L1
    acc_bias<2>(acc, bias + 256, h);
    mfma_layer_at<32, 2>(wp, bin, acc, lane);
    acc_to_bin<2, true>(acc, bin);
    if (tile) {
        store_acc_rows<2>(tile + 352 * TILE, bin, pt, h);
        reinterpret_cast<uint32_t *>(tile + FIELD_HID_ROWS * TILE)[3 * 64 + lane] = relu_mask32(bin);
    }
    wp += 1024;
    // color L2: 64 -> 3, sigmoid
    f32x16 o[1];
    acc_bias<1>(o, bias + 320, h);
    mfma_layer_at<32, 1>(wp, bin, o, lane);
    if (h == 0 && live) {
#pragma unroll
        for (int c = 0; c < 3; c++) albedo[p * 3 + c] = 1.0f / (1.0f + expf(-o[0][c]));
    }
    }  // tile loop
}

// =====================================================================================
// canonical field, FUSED backward: backward-data AND weight gradients in one pass, dPre never leaves the chip.
//
// A split form (backward-data kernel + mh_mlp_wgrad, removed in round 3) wrote every layer's pre-activation gradient
// (1.4 KB per point) and read it back together with the parked activations (3.1 KB per point): 9.7 GB of HBM traffic per
// step at the benchmark size (2.76 ms against 2.13 ms for this form).  Here a wave keeps the weight-gradient
// accumulators of its net IN REGISTERS across all its tiles -- the field nets are small enough: sdf_net 14 336 floats =
// 224 registers per lane, color_net 10 240 = 160 (the warp nets' 77 824 per net are not) -- so per
// tile and layer it (1) starts the loads of the layer's parked input activations H_l (feature-major rows = the B operand
// of the K = points MFMA, straight from HBM), (2) drops dPre_l (column form, in registers) into a private 9 KB LDS
// scratch and runs the backward-data MFMAs of the layer under the loads' latency, (3) reads dPre_l back ROW-wise (the A
// operand; row stride 36 floats keeps ds_read_b128 conflict-free and 16-byte aligned) and accumulates dW_l.
// One wave per SIMD (up to 512 registers), persistent 4-wave blocks, transposed weight packs resident in LDS; the two
// nets are two launches (color first: it hands d(geo) to the sdf launch through a 4 KB-per-tile scratch) so that either
// accumulator set fits.  Per-WORKGROUP partial sums (the four waves add up through LDS) go to the workspace in wgrad_kernel's [chunk][out x in] format and
// wgrad_reduce_kernel finishes as before.
// =====================================================================================
#define FUSED_THREADS 256
#define SCR_STRIDE 36                         // floats per scratch row (32 points + 4 pad)
#define SCR_FLOATS (64 * SCR_STRIDE)          // per wave: 64 rows
extern __shared__ f32x4 lds_fused[];

template <int N_F4>
__device__ __forceinline__ void stage_fused(const float *__restrict__ g, int dst_f4) {
    const f32x4 *src = reinterpret_cast<const f32x4 *>(g);
    for (int i = threadIdx.x; i < N_F4; i += FUSED_THREADS) lds_fused[dst_f4 + i] = src[i];
}

// column form (lane = point, registers = rows in accumulator order) -> scratch rows [row][point]
template <int MT>
__device__ __forceinline__ void scr_put(float *__restrict__ scr, const float (&v)[16 * MT], int pt, int h) {
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) scr[(32 * t + acc_row(r, h)) * SCR_STRIDE + pt] = v[16 * t + r];
}

struct RowFrag {
    f32x4 v[4];   // one feature row, 16 points of this lane half
};

__device__ __forceinline__ void scr_get(RowFrag &f, const float *__restrict__ scr, int row, int h) {
    const f32x4 *p = reinterpret_cast<const f32x4 *>(scr + row * SCR_STRIDE + 16 * h);
#pragma unroll
    for (int j = 0; j < 4; j++) f.v[j] = p[j];
}

// parked activation row (feature-major tile [F][32]) -> registers, asynchronously (the caller waits with fused_wait())
__device__ __forceinline__ void row_load_async(RowFrag &f, const float *__restrict__ tile, int row, int h) {
    const float *a = tile + (int64_t)row * TILE + 16 * h;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(f.v[0]) : "v"(a) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(f.v[1]) : "v"(a) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off offset:32" : "=v"(f.v[2]) : "v"(a) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off offset:48" : "=v"(f.v[3]) : "v"(a) : "memory");
}

// waits for the rows loaded by row_load_async and passes their registers through an empty volatile asm, so that no
// register-only use of them can be moved above the wait (see landed() further down)
template <int N>
__device__ __forceinline__ void fused_wait(RowFrag *__restrict__ B) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int n = 0; n < N; n++) asm volatile("" : "+v"(B[n].v[0]), "+v"(B[n].v[1]), "+v"(B[n].v[2]), "+v"(B[n].v[3]));
    __builtin_amdgcn_sched_barrier(0);
}

// dW[out tile][in tile n] += dPre rows (A, from the scratch) x H rows (B, from HBM) over the tile's 32 points;
// bsum += row sums of dPre (the bias gradient)
template <int NI>
__device__ __forceinline__ void dw_mma(const RowFrag &a, const RowFrag *__restrict__ b, f32x16 (&acc)[NI], float &bsum) {
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            bsum += a.v[j][q];
#pragma unroll
            for (int n = 0; n < NI; n++) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[j][q], b[n].v[j][q], acc[n], 0, 0, 0);
        }
}

// partial of this wave in wgrad_kernel's layout: [chunk][out_pad x in_pad], D[row = out (acc_row)][col = in (lane & 31)]
template <int NI>
__device__ __forceinline__ void dw_store(float *__restrict__ dw, const f32x16 (&acc)[NI], int mt, int in_pad, int i, int h) {
#pragma unroll
    for (int n = 0; n < NI; n++)
#pragma unroll
        for (int r = 0; r < 16; r++) dw[(int64_t)(32 * mt + acc_row(r, h)) * in_pad + 32 * n + i] = acc[n][r];
}

// Block-level sum of the four waves' weight-gradient accumulators, through the LDS the transposed weights no longer need once the
// tile loop is over: ONE partial per workgroup goes to the workspace instead of four (the partial sums of a 140 000-point call were
// 57 MB written and read back by wgrad_reduce_kernel, 23-38 us of reduction per call; waves add in the fixed order 1, 2, 3).
template <int N>
__device__ __forceinline__ float *acc_to_lds(const f32x16 (&a)[N], float *__restrict__ p, int lane) {
#pragma unroll
    for (int n = 0; n < N; n++)
#pragma unroll
        for (int r = 0; r < 16; r++) p[(n * 16 + r) * 64 + lane] = a[n][r];
    return p + N * 1024;
}
template <int N>
__device__ __forceinline__ const float *acc_add_lds(f32x16 (&a)[N], const float *__restrict__ p, int lane) {
#pragma unroll
    for (int n = 0; n < N; n++)
#pragma unroll
        for (int r = 0; r < 16; r++) a[n][r] += p[(n * 16 + r) * 64 + lane];
    return p + N * 1024;
}

// ---- the same two primitives with exact fp32 products on the bf16 matrix pipe (mlp_b3.hip's arithmetic) -------------------
#ifndef FUSED_FILL
#define FUSED_FILL 0        // measured round 5: 4 fillers per MFMA made the fused backward SLOWER (1.78 -> 1.84-1.93 ms: 6 more spilled registers in the sdf launch)
#endif
// weights: [plane hi|mid|lo][out tile][k16 step][lane][8 bf16] (packing.py: bwd3 blocks of the field packer); `bin` holds what
// the fp32 chain carries per 2-wide k-step, so k16 step s takes bin[8s .. 8s+7]
// S_RUN < KS / 8: the caller knows that bin[8 S_RUN ..] is all zeros (the short last layers: dP2 = [d geo | d sdf | zeros], dQ2 = three
// rows): those k16 steps are not multiplied -- 12 MFMAs and 44 slicing instructions per skipped step (round 6; the sums are the same
// numbers: a step of zeros adds +0 to every accumulator)
template <int KS, int MT, int S_RUN = KS / 8>
__device__ __forceinline__ void mfma_layer_z_b3(const f32x4 *__restrict__ w, const float (&bin)[KS], f32x16 (&acc)[MT], int lane) {
    static_assert(KS % 8 == 0 && S_RUN >= 1 && S_RUN <= KS / 8, "k16 steps");
    constexpr int S = KS / 8, PL = MT * S * 64;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < S_RUN; s++) {
        Frag bh, bm, bl;
#pragma unroll
        for (int e2 = 0; e2 < 4; e2++) split2(bin[8 * s + 2 * e2], bin[8 * s + 2 * e2 + 1], bh.u[e2], bm.u[e2], bl.u[e2]);
        Frag ah[MT], am[MT], al[MT];
#pragma unroll
        for (int t = 0; t < MT; t++) {
            ah[t].f = w[0 * PL + (t * S + s) * 64 + lane];
            am[t].f = w[1 * PL + (t * S + s) * 64 + lane];
            al[t].f = w[2 * PL + (t * S + s) * 64 + lane];
        }
        // slice products in ascending magnitude, the MT accumulators in rotation (no back-to-back dependent MFMAs)
#pragma unroll
        for (int t = 0; t < MT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[t].h, bh.h, s == 0 ? zero : acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < MT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[t].h, bm.h, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < MT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t].h, bl.h, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < MT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[t].h, bh.h, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < MT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t].h, bm.h, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < MT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t].h, bh.h, acc[t], 0, 0, 0);
    }
    // round 5: step s + 1's slicing (44 VALU) rides in the shadow of step s's 6 MT MFMAs instead of running between them
    // (tools/micro/mfma_valu_gap.hip: 5-6 single-issue instructions per MFMA gap are free inside a wave; this kernel is one wave per SIMD)
#if FUSED_FILL > 0
#pragma unroll
    for (int g_ = 0; g_ < S_RUN * 6 * MT; g_++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, FUSED_FILL, 0);
    }
#endif
}

// ONE k16 step `s` of a layer of S steps, accumulators = that step's products alone (every other step's B operand is zero)
template <int S, int MT>
__device__ __forceinline__ void mfma_kstep_z_b3(const f32x4 *__restrict__ w, int s, const float *__restrict__ bin8, f32x16 (&acc)[MT], int lane) {
    constexpr int PL = MT * S * 64;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    Frag bh, bm, bl;
#pragma unroll
    for (int e2 = 0; e2 < 4; e2++) split2(bin8[2 * e2], bin8[2 * e2 + 1], bh.u[e2], bm.u[e2], bl.u[e2]);
#pragma unroll
    for (int t = 0; t < MT; t++) {
        Frag ah, am, al;
        ah.f = w[0 * PL + (t * S + s) * 64 + lane];
        am.f = w[1 * PL + (t * S + s) * 64 + lane];
        al.f = w[2 * PL + (t * S + s) * 64 + lane];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al.h, bh.h, zero, 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am.h, bm.h, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.h, bl.h, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am.h, bh.h, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.h, bm.h, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.h, bh.h, acc[t], 0, 0, 0);
    }
}

// K = points: k16 step s, lane half g, element e <-> point 16 g + 8 s + e, i.e. v[2s], v[2s+1] of a RowFrag (both operands)
struct RowSl {
    Frag h[2], m[2], l[2];
};
__device__ __forceinline__ void row_slices(const RowFrag &f, RowSl &o) {
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
        for (int e2 = 0; e2 < 4; e2++)
            split2(f.v[2 * s + (e2 >> 1)][2 * (e2 & 1)], f.v[2 * s + (e2 >> 1)][2 * (e2 & 1) + 1], o.h[s].u[e2], o.m[s].u[e2], o.l[s].u[e2]);
}
// the B rows (a layer's parked input activations) are sliced ONCE per tile and layer by the caller and serve every out tile
template <int NI>
__device__ __forceinline__ void dw_mma_b3(const RowFrag &a, const RowSl *__restrict__ b, f32x16 (&acc)[NI], float &bsum) {
#pragma unroll
    for (int j = 0; j < 4; j++) bsum += (a.v[j][0] + a.v[j][1]) + (a.v[j][2] + a.v[j][3]);
    RowSl as;
    row_slices(a, as);
#pragma unroll
    for (int s = 0; s < 2; s++) {
#pragma unroll
        for (int n = 0; n < NI; n++) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as.l[s].h, b[n].h[s].h, acc[n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NI; n++) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as.m[s].h, b[n].m[s].h, acc[n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NI; n++) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as.h[s].h, b[n].l[s].h, acc[n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NI; n++) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as.m[s].h, b[n].h[s].h, acc[n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NI; n++) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as.h[s].h, b[n].m[s].h, acc[n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NI; n++) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as.h[s].h, b[n].h[s].h, acc[n], 0, 0, 0);
    }
#if FUSED_FILL > 0
#pragma unroll
    for (int g_ = 0; g_ < 12 * NI; g_++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, FUSED_FILL, 0);
    }
#endif
}
// arithmetic selector of the fused kernels (B3 is their template parameter): LAYER(KS, MT, w, bin, acc); SLICE(N, B, Bs) after the
// parked rows landed; DW(NI, A, B, Bs, acc, bsum)
#define FUSED_LAYER(KS, MT, w, bin, acc) do { if (B3) mfma_layer_z_b3<KS, MT>(w, bin, acc, lane); else mfma_layer_z<KS, MT>(w, bin, acc, lane); } while (0)
#define FUSED_SLICE(N, B, Bs) do { if (B3) { _Pragma("unroll") for (int n_ = 0; n_ < N; n_++) row_slices(B[n_], Bs[n_]); } } while (0)
#define FUSED_DW(NI, A, B, Bs, acc, bsum) do { if (B3) dw_mma_b3<NI>(A, Bs, acc, bsum); else dw_mma<NI>(A, B, acc, bsum); } while (0)
// float4 sizes of the transposed blocks in LDS: fp32 fragments / bf16x3 slices (packing.py pads every block to whole 512s)
#define FUSED_TC2(B3) ((B3) ? 1024 : 512)
#define FUSED_TC1(B3) ((B3) ? 1536 : 1024)
#define FUSED_TC0(B3) ((B3) ? 1536 : 1024)
#define FUSED_TS2(B3) ((B3) ? 1536 : 1024)
#define FUSED_TS1(B3) ((B3) ? 1536 : 1024)
#define FUSED_TS0(B3) ((B3) ? 2560 : 1536)

#ifdef MH_PHASE_TRACE
// phase trace for tools/phase_trace_field_bwd.py (never compiled into the product library): wave 0 of every 8th workgroup stamps
// s_memtime at the phase boundaries of its THIRD tile in field_fused_sdf_kernel (16 slots), s_memrealtime in 30/31
__device__ long long mh_fused_trace[64 * 32];
#define FUSED_STAMP(slot)                                                                                   \
    do {                                                                                                    \
        if (trace_it == 2 && threadIdx.x == 0 && (blockIdx.x % 8 == 0) && (blockIdx.x / 8 < 64))            \
            mh_fused_trace[(blockIdx.x / 8) * 32 + (slot)] = (long long)__builtin_amdgcn_s_memtime();       \
    } while (0)
#define FUSED_STAMP_REAL(slot)                                                                              \
    do {                                                                                                    \
        if (trace_it == 2 && threadIdx.x == 0 && (blockIdx.x % 8 == 0) && (blockIdx.x / 8 < 64))            \
            mh_fused_trace[(blockIdx.x / 8) * 32 + (slot)] = (long long)__builtin_amdgcn_s_memrealtime();   \
    } while (0)
extern "C" int mh_fused_trace_read(long long *dst_host) {
    return hipMemcpyFromSymbol(dst_host, HIP_SYMBOL(mh_fused_trace), sizeof(long long) * 64 * 32) == hipSuccess ? 0 : 2;
}
#else
#define FUSED_STAMP(slot) do { } while (0)
#define FUSED_STAMP_REAL(slot) do { } while (0)
#endif

struct FusedPart {            // where this launch's per-wave partial sums go (float offsets into the workspace)
    int64_t dw[3];            // layer-major: [n_chunks][out_pad * in_pad]
    int64_t db[3];            // [n_chunks][out_pad]
    int64_t gb;               // sdf launch: [n_chunks] partial sums of d(loss)/d(beta)
    int64_t dw_s2, db_s2;     // colour launch: where the sdf net's last layer keeps its partials (the geo rows' are computed here)
};

// ---- color_net: Q2 (3 rows) <- g_albedo, Q1, Q0; hands d(geo) to the sdf launch ------------------------------------
template <bool B3>
__global__ __launch_bounds__(FUSED_THREADS, 1) void field_fused_color_kernel(
    const float *__restrict__ albedo, const float *__restrict__ g_albedo, const float *__restrict__ wpackT,
    const float *__restrict__ acts, float *__restrict__ dgeo_scr, float *__restrict__ g_feat_c, float *__restrict__ ws,
    FusedPart part, uint32_t *__restrict__ gmax, int64_t M, int64_t n_tiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pt = lane & 31, h = lane >> 5, i = lane & 31;
    constexpr int W_F4 = FUSED_TC2(B3) + FUSED_TC1(B3) + FUSED_TC0(B3);
    stage_fused<W_F4>(wpackT, 0);     // TC2 | TC1 | TC0 (fp32 fragments, or their bf16x3 slices)
    float *scr = reinterpret_cast<float *>(lds_fused + W_F4) + wave * SCR_FLOATS;
    __syncthreads();
    f32x16 w2[2], w1[2][2], w0[2][2];                      // dW of c2 [1 out tile][2 in], c1, c0 [2][2]: 160 registers
    // + the geo rows of the SDF net's last layer, dW_s2[geo][:] = d(geo) H2^T: d(geo) is produced HERE (this kernel's last
    // backward-data tile), and the sdf launch -- 160 accumulator registers of its own -- has no room for these 32 in its sliced form
    f32x16 wg[2];
    float bg = 0.f;
    acc_zero<2>(wg);
    acc_zero<2>(w2);
    acc_zero<2>(w1[0]);
    acc_zero<2>(w1[1]);
    acc_zero<2>(w0[0]);
    acc_zero<2>(w0[1]);
    float b2 = 0.f, b1[2] = {0.f, 0.f}, b0[2] = {0.f, 0.f};
    uint32_t max_c = 0;
    const int chunk = blockIdx.x * (FUSED_THREADS / 64) + wave, n_chunks = gridDim.x * (FUSED_THREADS / 64);
    RowFrag B[2];
    for (int64_t tile_id = chunk; tile_id < n_tiles; tile_id += n_chunks) {
        const int64_t p = tile_id * TILE + pt;
        const bool live = p < M;
        const float *atile = acts + tile_id * (int64_t)(FIELD_ACT_ROWS * TILE);
        const uint32_t *masks = reinterpret_cast<const uint32_t *>(atile + FIELD_HID_ROWS * TILE);
        const f32x4 *wt = lds_fused;
        RowFrag A;
        RowSl Bs[2];
        f32x16 acc[2];
        float dbin[32];
        // dQ2 = g_albedo * a * (1 - a)
        float d2[16];
#pragma unroll
        for (int r = 0; r < 16; r++) d2[r] = 0.f;
        if (g_albedo && live && h == 0) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float a = albedo[p * 3 + c];
                d2[c] = g_albedo[p * 3 + c] * a * (1.0f - a);
            }
        }
        const uint32_t mw3 = masks[3 * 64 + lane], mw2 = masks[2 * 64 + lane];
        // ---- layer c2: input C2 = rows 352..415
        row_load_async(B[0], atile, 352 + i, h);
        row_load_async(B[1], atile, 384 + i, h);
        scr_put<1>(scr, d2, pt, h);
        if (B3) mfma_layer_z_b3<16, 2, 1>(wt, d2, acc, lane);      // (rows 8.. of dQ2 are zeros: k16 step 0 only)
        else mfma_layer_z<16, 2>(wt, d2, acc, lane);
        wt += FUSED_TC2(B3);
#pragma unroll
        for (int j = 0; j < 32; j++) dbin[j] = mask_bit(mw3, j, acc[j >> 4][j & 15]);   // mask C2 -> dQ1
        scr_get(A, scr, i, h);
        fused_wait<2>(B);
        FUSED_SLICE(2, B, Bs);
        FUSED_DW(2, A, B, Bs, w2, b2);
        __builtin_amdgcn_sched_barrier(0);
        // ---- layer c1: input C1 = rows 288..351
        row_load_async(B[0], atile, 288 + i, h);
        row_load_async(B[1], atile, 320 + i, h);
        scr_put<2>(scr, dbin, pt, h);
        FUSED_LAYER(32, 2, wt, dbin, acc);
        wt += FUSED_TC1(B3);
        {
            float q1[32];
#pragma unroll
            for (int j = 0; j < 32; j++) q1[j] = mask_bit(mw2, j, acc[j >> 4][j & 15]);   // mask C1 -> dQ0
            fused_wait<2>(B);
            FUSED_SLICE(2, B, Bs);
#pragma unroll
            for (int mt = 0; mt < 2; mt++) {
                scr_get(A, scr, 32 * mt + i, h);
                FUSED_DW(2, A, B, Bs, w1[mt], b1[mt]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 32; j++) dbin[j] = q1[j];
        }
        // ---- layer c0: input [hash_c | geo] = rows 224..287
        row_load_async(B[0], atile, 224 + i, h);
        row_load_async(B[1], atile, 256 + i, h);
        scr_put<2>(scr, dbin, pt, h);
        FUSED_LAYER(32, 2, wt, dbin, acc);
        if (live) {
            if (g_feat_c) {
                f32x4 *o = reinterpret_cast<f32x4 *>(g_feat_c + p * 32 + 16 * h);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    f32x4 v;
#pragma unroll
                    for (int c = 0; c < 4; c++) v[c] = acc[0][4 * q + c];
                    o[q] = v;
                }
#pragma unroll
                for (int r = 0; r < 16; r++) max_c = max(max_c, __float_as_uint(fabsf(acc[0][r])));
            }
        }
        {   // d(geo), accumulator order, for the sdf launch: [tile][lane][16]
            f32x4 *o = reinterpret_cast<f32x4 *>(dgeo_scr + (tile_id * 64 + lane) * 16);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                f32x4 v;
#pragma unroll
                for (int c = 0; c < 4; c++) v[c] = acc[1][4 * q + c];
                o[q] = v;
            }
        }
        fused_wait<2>(B);
        FUSED_SLICE(2, B, Bs);
        RowFrag Bg[2];                                    // the sdf net's parked S2 rows (160..223): land under the c0 weight gradient
        row_load_async(Bg[0], atile, 160 + i, h);
        row_load_async(Bg[1], atile, 192 + i, h);
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
            scr_get(A, scr, 32 * mt + i, h);
            FUSED_DW(2, A, B, Bs, w0[mt], b0[mt]);
        }
        __builtin_amdgcn_sched_barrier(0);
        {   // dW_s2, geo rows: A = d(geo) rows through the scratch (free now), B = S2 rows
            float dg[16];
#pragma unroll
            for (int r = 0; r < 16; r++) dg[r] = acc[1][r];
            scr_put<1>(scr, dg, pt, h);
            fused_wait<2>(Bg);
            FUSED_SLICE(2, Bg, Bs);
            scr_get(A, scr, i, h);
            FUSED_DW(2, A, Bg, Bs, wg, bg);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (gmax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) max_c = max(max_c, (uint32_t)__shfl_xor((int)max_c, o));
        if (lane == 0 && max_c) atomicMax(gmax + 1, max_c);
    }
    bg += __shfl_xor(bg, 32);
    b2 += __shfl_xor(b2, 32);
#pragma unroll
    for (int mt = 0; mt < 2; mt++) {
        b1[mt] += __shfl_xor(b1[mt], 32);
        b0[mt] += __shfl_xor(b0[mt], 32);
    }
    // the workgroup's partial = the sum of its four waves' (acc_to_lds above): 12 accumulator tiles + 6 bias rows = ~50 KB of LDS
    float *red = reinterpret_cast<float *>(lds_fused);
    for (int src = 1; src < FUSED_THREADS / 64; src++) {
        __syncthreads();
        if (wave == src) {
            float *q = acc_to_lds<2>(w2, red, lane);
            q = acc_to_lds<2>(w1[0], q, lane);
            q = acc_to_lds<2>(w1[1], q, lane);
            q = acc_to_lds<2>(w0[0], q, lane);
            q = acc_to_lds<2>(w0[1], q, lane);
            q = acc_to_lds<2>(wg, q, lane);
            q[5 * 64 + lane] = bg;
            q[0 * 64 + lane] = b2;
            q[1 * 64 + lane] = b1[0];
            q[2 * 64 + lane] = b1[1];
            q[3 * 64 + lane] = b0[0];
            q[4 * 64 + lane] = b0[1];
        }
        __syncthreads();
        if (wave == 0) {
            const float *q = acc_add_lds<2>(w2, red, lane);
            q = acc_add_lds<2>(w1[0], q, lane);
            q = acc_add_lds<2>(w1[1], q, lane);
            q = acc_add_lds<2>(w0[0], q, lane);
            q = acc_add_lds<2>(w0[1], q, lane);
            q = acc_add_lds<2>(wg, q, lane);
            bg += q[5 * 64 + lane];
            b2 += q[0 * 64 + lane];
            b1[0] += q[1 * 64 + lane];
            b1[1] += q[2 * 64 + lane];
            b0[0] += q[3 * 64 + lane];
            b0[1] += q[4 * 64 + lane];
        }
    }
    if (wave != 0) return;
    // partial sums of this workgroup: layers in the launch's order c0, c1, c2 = part.dw[0..2]
    const int64_t pchunk = blockIdx.x;
    dw_store<2>(ws + part.dw[2] + pchunk * 32 * 64, w2, 0, 64, i, h);
#pragma unroll
    for (int mt = 0; mt < 2; mt++) {
        dw_store<2>(ws + part.dw[1] + pchunk * 64 * 64, w1[mt], mt, 64, i, h);
        dw_store<2>(ws + part.dw[0] + pchunk * 64 * 64, w0[mt], mt, 64, i, h);
    }
    dw_store<2>(ws + part.dw_s2 + pchunk * 64 * 64, wg, 0, 64, i, h);      // out tile 0 (the geo rows) of the sdf net's last layer
    if (h == 0) {
        ws[part.db[2] + pchunk * 32 + i] = b2;
        ws[part.db_s2 + pchunk * 64 + i] = bg;
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
            ws[part.db[1] + pchunk * 64 + 32 * mt + i] = b1[mt];
            ws[part.db[0] + pchunk * 64 + 32 * mt + i] = b0[mt];
        }
    }
}

// ---- sdf_net (+ Laplace density): P2 <- [d(geo) | g_sdf, g_sigma], P1, P0, d(inputs) ---------------------------------
template <bool WITH_COLOR, bool B3>
__global__ __launch_bounds__(FUSED_THREADS, 1) void field_fused_sdf_kernel(
    const float *__restrict__ xc, const float *__restrict__ sdf, const float *__restrict__ g_sdf,
    const float *__restrict__ g_sigma, const float *__restrict__ wpackT, const float *__restrict__ beta_p, int n_bands,
    const float *__restrict__ acts, const float *__restrict__ dgeo_scr, float *__restrict__ g_xc,
    float *__restrict__ g_feat_s, float *__restrict__ g_topo, float *__restrict__ ws,
    FusedPart part, uint32_t *__restrict__ gmax, int64_t M, int64_t n_tiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pt = lane & 31, h = lane >> 5, i = lane & 31;
    constexpr int W_F4 = FUSED_TS2(B3) + FUSED_TS1(B3) + FUSED_TS0(B3);
    // TS2 | TS1 | TS0 behind the colour net's three blocks (fp32 fragments, or their bf16x3 slices)
    stage_fused<W_F4>(wpackT + 4 * (FUSED_TC2(B3) + FUSED_TC1(B3) + FUSED_TC0(B3)), 0);
    float *scr = reinterpret_cast<float *>(lds_fused + W_F4) + wave * SCR_FLOATS;
    __syncthreads();
    // dP2 = [d geo (32 rows, tile 0; zero on the sdf-only pass) | d sdf (ONE row: tile 1, row 0)].  The sdf row's weight gradient
    // dW2[sdf][in] = sum_pt g[pt] H2[in][pt] is 2 x 16 fused multiply-adds per lane on the rows the lane holds anyway (`wsdf`),
    // not a 32 x 64 matrix tile of which one row is not zero: 32 accumulator registers and 32 (24) MFMAs per tile less
    // (the geo rows' weight gradient is the colour launch's: it produces d(geo))
    f32x16 w1[2][2], w0[2][3];                             // 64 + 96 = 160 registers
    float wsdf[2] = {0.f, 0.f}, bsdf = 0.f;
    acc_zero<2>(w1[0]);
    acc_zero<2>(w1[1]);
    acc_zero<3>(w0[0]);
    acc_zero<3>(w0[1]);
    float b1[2] = {0.f, 0.f}, b0[2] = {0.f, 0.f};
    uint32_t max_s = 0;
    float gb_acc = 0.f;                                    // d(loss)/d(beta) of this lane's points (lanes h == 0 carry it)
    const float beta = *beta_p;
    const int chunk = blockIdx.x * (FUSED_THREADS / 64) + wave, n_chunks = gridDim.x * (FUSED_THREADS / 64);
    RowFrag B[3];                     // a layer's parked input rows
    int trace_it = -1;
    (void)trace_it;
    for (int64_t tile_id = chunk; tile_id < n_tiles; tile_id += n_chunks) {
        trace_it++;
        FUSED_STAMP_REAL(30);
        FUSED_STAMP(0);
        const int64_t p = tile_id * TILE + pt;
        const bool live = p < M;
        const int64_t pc = live ? p : M - 1;
        const float *atile = acts + tile_id * (int64_t)(FIELD_ACT_ROWS * TILE);
        const uint32_t *masks = reinterpret_cast<const uint32_t *>(atile + FIELD_HID_ROWS * TILE);
        const f32x4 *wt = lds_fused;
        RowFrag A;
        f32x16 acc[2];
        float dbin[32];
        // dP2 = [d geo | d sdf]
        float gs = 0.f, gbeta = 0.f;
        if (live && h == 0) {
            const float s = sdf[p];
            if (g_sdf) gs = g_sdf[p];
            if (g_sigma) {
                const float gsg = g_sigma[p];
                const float sg = (s > 0.f) ? 1.f : ((s < 0.f) ? -1.f : 0.f);
                const float a = fabsf(s) / beta;
                const float ex = expf(-a);
                gs += gsg * (-(0.5f / (beta * beta)) * sg * sg * ex);
                gbeta = gsg * (-(1.0f / (beta * beta)) * laplace_unit(s, beta) +
                               (1.0f / beta) * (0.5f * sg * ex * (fabsf(s) / (beta * beta))));
            }
        }
        gb_acc += gbeta;
        const uint32_t mw1 = masks[1 * 64 + lane], mw0 = masks[0 * 64 + lane];
        float d2[32];
        if (WITH_COLOR) {
            const f32x4 *gsrc = reinterpret_cast<const f32x4 *>(dgeo_scr + (tile_id * 64 + lane) * 16);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const f32x4 v = gsrc[q];
#pragma unroll
                for (int c = 0; c < 4; c++) d2[4 * q + c] = v[c];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++) d2[r] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; r++) d2[16 + r] = 0.f;
        d2[16] = gs;  // tile 1, row 0 (only h == 0 lanes carry a non-zero gs)
        // ---- layer s2: input S2 = rows 160..223
        {
            RowSl Bs[2];
            row_load_async(B[0], atile, 160 + i, h);
            row_load_async(B[1], atile, 192 + i, h);
            FUSED_STAMP(1);
            scr_put<2>(scr, d2, pt, h);
            if (WITH_COLOR) {
                if (B3) mfma_layer_z_b3<32, 2, 3>(wt, d2, acc, lane);      // (dP2 rows 24.. are zeros: k16 steps 0, 1, 2)
                else mfma_layer_z<32, 2>(wt, d2, acc, lane);
            } else if (B3) {
                // d2 is zero but for g_sdf = d2[16]: the k16 step 2 of the layer's 4 (12 MFMAs instead of 48)
                mfma_kstep_z_b3<4, 2>(wt, 2, &d2[16], acc, lane);
            } else {
                // dH2 = W2[sdf,:]^T g_sdf is the single k-step 16 of the 32 (2 MFMAs instead of 64)
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const f32x4 a = wt[(t * 8 + 4) * 64 + lane];
                    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], gs, z, 0, 0, 0);
                }
            }
            wt += FUSED_TS2(B3);
#pragma unroll
            for (int j = 0; j < 32; j++) dbin[j] = mask_bit(mw1, j, acc[j >> 4][j & 15]);   // mask S2 -> dP1
            FUSED_STAMP(2);
            fused_wait<2>(B);
            FUSED_STAMP(3);
            {   // the sdf row (scratch row 32: every lane reads the same 16 points of its half -- an LDS broadcast)
                RowFrag G;
                scr_get(G, scr, 32, h);
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const float g = G.v[j][q];
                        wsdf[0] = fmaf(g, B[0].v[j][q], wsdf[0]);
                        wsdf[1] = fmaf(g, B[1].v[j][q], wsdf[1]);
                        bsdf += g;
                    }
            }
            FUSED_STAMP(4);
            __builtin_amdgcn_sched_barrier(0);
            FUSED_STAMP(5);
        }
        // ---- layer s1: input S1 = rows 96..159
        {
            RowSl Bs[2];
            row_load_async(B[0], atile, 96 + i, h);
            row_load_async(B[1], atile, 128 + i, h);
            scr_put<2>(scr, dbin, pt, h);
            FUSED_LAYER(32, 2, wt, dbin, acc);
            wt += FUSED_TS1(B3);
            float q0[32];
#pragma unroll
            for (int j = 0; j < 32; j++) q0[j] = mask_bit(mw0, j, acc[j >> 4][j & 15]);     // mask S1 -> dP0
            FUSED_STAMP(6);
            fused_wait<2>(B);
            FUSED_STAMP(7);
            FUSED_SLICE(2, B, Bs);
            FUSED_STAMP(8);
#pragma unroll
            for (int mt = 0; mt < 2; mt++) {
                scr_get(A, scr, 32 * mt + i, h);
                FUSED_DW(2, A, B, Bs, w1[mt], b1[mt]);
            }
            __builtin_amdgcn_sched_barrier(0);
            FUSED_STAMP(9);
#pragma unroll
            for (int j = 0; j < 32; j++) dbin[j] = q0[j];
        }
        // ---- layer s0: input [enc | hash | topo] = rows 0..95 (k-step order)
        {
            RowSl Bs[3];
            row_load_async(B[0], atile, 0 + i, h);
            row_load_async(B[1], atile, 32 + i, h);
            row_load_async(B[2], atile, 64 + min(i, 15), h);     // rows 80..95 are unwritten padding: their lanes take row 79 (k-step 39: zeros)
            scr_put<2>(scr, dbin, pt, h);
            // d(inputs) = W0^T dP0: tile0 = enc kk 0..15, tile1 = enc kk 16..19 + topo at r=4, tile2 = hash (16h + r)
            f32x16 e[3];
            if (g_xc || B3) {
                FUSED_LAYER(32, 3, wt, dbin, e);       // (the bf16x3 planes of the three out tiles are not contiguous per tile)
            } else {
                f32x16 e12[2];
                mfma_layer_z<32, 2>(wt + 8 * 64, dbin, e12, lane);
                e[1] = e12[0];
                e[2] = e12[1];
            }
            // the weight gradient first (it frees the 48 registers of the activation rows), the sincos stage after it
            FUSED_STAMP(10);
            fused_wait<3>(B);
            FUSED_STAMP(11);
            FUSED_SLICE(3, B, Bs);
            FUSED_STAMP(12);
#pragma unroll
            for (int mt = 0; mt < 2; mt++) {
                scr_get(A, scr, 32 * mt + i, h);
                FUSED_DW(3, A, B, Bs, w0[mt], b0[mt]);
            }
            __builtin_amdgcn_sched_barrier(0);
            FUSED_STAMP(13);
            float gx[3] = {0.f, 0.f, 0.f};
            if (g_xc) {
                float dsc[18];
                enc_deriv_parked(atile, pt, h, dsc);       // S0 rows 0..35 hold the encoding of xc
#pragma unroll
                for (int k = 0; k < 18; k++) {
                    const float de = k < 16 ? e[0][k] : e[1][k - 16];
                    gx[k % 3] += de * dsc[k];
                }
                if (h == 0) {
                    gx[0] += e[1][2];
                    gx[2] += e[1][3];
                } else {
                    gx[1] += e[1][2];
                }
#pragma unroll
                for (int d = 0; d < 3; d++) gx[d] += __shfl_xor(gx[d], 32);
            }
            if (live) {
                if (g_xc && h == 0) {
                    g_xc[p * 3 + 0] = gx[0];
                    g_xc[p * 3 + 1] = gx[1];
                    g_xc[p * 3 + 2] = gx[2];
                }
                if (g_topo) g_topo[p * 2 + h] = e[1][4];
                if (g_feat_s) {
                    f32x4 *o = reinterpret_cast<f32x4 *>(g_feat_s + p * 32 + 16 * h);
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        f32x4 v;
#pragma unroll
                        for (int c = 0; c < 4; c++) v[c] = e[2][4 * q + c];
                        o[q] = v;
                    }
#pragma unroll
                    for (int r = 0; r < 16; r++) max_s = max(max_s, __float_as_uint(fabsf(e[2][r])));
                }
            }
        }
        FUSED_STAMP(14);
        FUSED_STAMP_REAL(31);
    }
    if (gmax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) max_s = max(max_s, (uint32_t)__shfl_xor((int)max_s, o));
        if (lane == 0 && max_s) atomicMax(gmax + 0, max_s);
    }
    wsdf[0] += __shfl_xor(wsdf[0], 32);
    wsdf[1] += __shfl_xor(wsdf[1], 32);
    bsdf += __shfl_xor(bsdf, 32);
#pragma unroll
    for (int mt = 0; mt < 2; mt++) {
        b1[mt] += __shfl_xor(b1[mt], 32);
        b0[mt] += __shfl_xor(b0[mt], 32);
    }
    // the workgroup's partial = the sum of its four waves' (acc_to_lds above): up to 14 accumulator tiles + 6 bias rows = 57.5 KB
    float *red = reinterpret_cast<float *>(lds_fused);
    for (int src = 1; src < FUSED_THREADS / 64; src++) {
        __syncthreads();
        if (wave == src) {
            float *q = red;
            q = acc_to_lds<2>(w1[0], q, lane);
            q = acc_to_lds<2>(w1[1], q, lane);
            q = acc_to_lds<3>(w0[0], q, lane);
            q = acc_to_lds<3>(w0[1], q, lane);
            q[1 * 64 + lane] = bsdf;
#pragma unroll
            for (int mt = 0; mt < 2; mt++) {
                q[(2 + mt) * 64 + lane] = b1[mt];
                q[(4 + mt) * 64 + lane] = b0[mt];
                q[(7 + mt) * 64 + lane] = wsdf[mt];
            }
            q[6 * 64 + lane] = gb_acc;
        }
        __syncthreads();
        if (wave == 0) {
            const float *q = red;
            q = acc_add_lds<2>(w1[0], q, lane);
            q = acc_add_lds<2>(w1[1], q, lane);
            q = acc_add_lds<3>(w0[0], q, lane);
            q = acc_add_lds<3>(w0[1], q, lane);
            bsdf += q[1 * 64 + lane];
#pragma unroll
            for (int mt = 0; mt < 2; mt++) {
                b1[mt] += q[(2 + mt) * 64 + lane];
                b0[mt] += q[(4 + mt) * 64 + lane];
                wsdf[mt] += q[(7 + mt) * 64 + lane];
            }
            gb_acc += q[6 * 64 + lane];
        }
    }
    if (wave != 0) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) gb_acc += __shfl_xor(gb_acc, o);
    if (lane == 0) ws[part.gb + blockIdx.x] = gb_acc;
    // partial sums of this workgroup: layers s0, s1, s2 = part.dw[0..2]; the sdf-only pass fills tile 1 of s2 only (tile 0 = zeros)
    const int64_t pchunk = blockIdx.x;
#pragma unroll
    for (int mt = 0; mt < 2; mt++) {
        dw_store<3>(ws + part.dw[0] + pchunk * 64 * 96, w0[mt], mt, 96, i, h);
        dw_store<2>(ws + part.dw[1] + pchunk * 64 * 64, w1[mt], mt, 64, i, h);
    }
    {
        f32x16 z[2];
        acc_zero<2>(z);
        if (!WITH_COLOR) dw_store<2>(ws + part.dw[2] + pchunk * 64 * 64, z, 0, 64, i, h);      // (with colour: the colour launch's)
        if (h == 0) {                 // tile 1: row 0 = the sdf row (accumulator row r = 0 of the lanes h == 0), the other 31 are zero
            z[0][0] = wsdf[0];
            z[1][0] = wsdf[1];
        }
        dw_store<2>(ws + part.dw[2] + pchunk * 64 * 64, z, 1, 64, i, h);
    }
    if (h == 0) {
        if (!WITH_COLOR) ws[part.db[2] + pchunk * 64 + i] = 0.f;
        ws[part.db[2] + pchunk * 64 + 32 + i] = i == 0 ? bsdf : 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
            ws[part.db[1] + pchunk * 64 + 32 * mt + i] = b1[mt];
            ws[part.db[0] + pchunk * 64 + 32 * mt + i] = b0[mt];
        }
    }
}

// =====================================================================================
// weight gradients: dW[out][in] = sum_pt dPre[out][pt] * act[in][pt]  (MFMA, K = points)
// one wave per 32-row out tile; a block = OT waves sharing the act tiles through L1.  Fragments of
// tile t+1 are fetched into a second register set while tile t's MFMAs issue (the operands come
// straight from HBM: 2.15 GB per 128x128 layer at the benchmark size).
// =====================================================================================
template <int IT>
struct WgFrag {
    f32x4 a[4];
    f32x4 b[IT][4];
};

// Pad rows are neither written nor read (round 6): the first layer's input tile has 64 rows for 40 encoding values, the last
// layer's dPre tile 32 rows for 3 | 2 outputs -- 1.4 GB of a cfg3 step's parked traffic.  A lane whose row lies behind the layer's
// live rows re-reads the LAST live row (the line is being fetched for its owner anyway: no extra traffic, the same number of load
// instructions, so the counted waits stand): its values only reach dW rows / columns of pad features, which nobody gathers.
__device__ __forceinline__ int wg_row(int row, int live) { return row < live ? row : live - 1; }

template <int IT>
__device__ __forceinline__ void wg_load(WgFrag<IT> &f, const float *__restrict__ acts, const float *__restrict__ dpre,
                                        int64_t t, int64_t acts_tile_floats, int64_t dpre_tile_floats, int act_off,
                                        int dpre_off, int mt, int i, int h, int in_live, int out_live) {
    const f32x4 *a = reinterpret_cast<const f32x4 *>(dpre + t * dpre_tile_floats + dpre_off + (int64_t)wg_row(32 * mt + i, out_live) * TILE + 16 * h);
#pragma unroll
    for (int j = 0; j < 4; j++) f.a[j] = a[j];
#pragma unroll
    for (int n = 0; n < IT; n++) {
        const f32x4 *b = reinterpret_cast<const f32x4 *>(acts + t * acts_tile_floats + act_off + (int64_t)wg_row(32 * n + i, in_live) * TILE + 16 * h);
#pragma unroll
        for (int j = 0; j < 4; j++) f.b[n][j] = b[j];
    }
}

// Same loads, but issued through inline asm so that hipcc neither waits for them nor re-rolls the software
// pipeline (cdna_hip_programming.md 5.7): the caller owns the s_waitcnt vmcnt(N) accounting.
template <int IT>
__device__ __forceinline__ void wg_load_async(WgFrag<IT> &f, const float *__restrict__ acts, const float *__restrict__ dpre,
                                              int64_t t, int64_t acts_tile_floats, int64_t dpre_tile_floats, int act_off,
                                              int dpre_off, int mt, int i, int h, int in_live, int out_live) {
    const float *a = dpre + t * dpre_tile_floats + dpre_off + (int64_t)wg_row(32 * mt + i, out_live) * TILE + 16 * h;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(f.a[0]) : "v"(a) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(f.a[1]) : "v"(a) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off offset:32" : "=v"(f.a[2]) : "v"(a) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off offset:48" : "=v"(f.a[3]) : "v"(a) : "memory");
#pragma unroll
    for (int n = 0; n < IT; n++) {
        const float *b = acts + t * acts_tile_floats + act_off + (int64_t)wg_row(32 * n + i, in_live) * TILE + 16 * h;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(f.b[n][0]) : "v"(b) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(f.b[n][1]) : "v"(b) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:32" : "=v"(f.b[n][2]) : "v"(b) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:48" : "=v"(f.b[n][3]) : "v"(b) : "memory");
    }
}

template <int N>
__device__ __forceinline__ void wg_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);  // keep register-only MFMAs below the wait (guide rule 18)
}
// The compiler takes an inline-asm load's output as valid from the asm statement on, so it may move register-only work on
// it (a slice conversion, a copy) ABOVE the s_waitcnt that actually makes the data valid: the "memory" clobber orders memory
// operations, the scheduling fence orders the machine scheduler, neither orders IR-level code motion of pure arithmetic.
// (Seen: with the round-to-nearest split2 the merged b3 weight-gradient kernel sliced a register set before it had landed.)
// Passing the registers through an empty volatile asm right after the wait makes every later use depend on that point.
__device__ __forceinline__ void landed(f32x4 (&v)[4]) { asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])); }
template <int IT>
__device__ __forceinline__ void landed(WgFrag<IT> &f) {
    landed(f.a);
#pragma unroll
    for (int n = 0; n < IT; n++) landed(f.b[n]);
}

template <int IT>
__device__ __forceinline__ void wg_mma(const WgFrag<IT> &f, f32x16 (&acc)[IT], float &bsum) {
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            bsum += f.a[j][q];
#pragma unroll
            for (int n = 0; n < IT; n++)
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[j][q], f.b[n][j][q], acc[n], 0, 0, 0);
        }
}

template <int IT>
__device__ __forceinline__ void wgrad_body(const float *__restrict__ acts, const float *__restrict__ dpre,
                                           int64_t acts_tile_floats, int64_t dpre_tile_floats, int act_off, int dpre_off,
                                           int out_pad, float *__restrict__ dw_part, float *__restrict__ db_part,
                                           int64_t n_tiles, int n_chunks, int chunk, int in_live, int out_live) {
    const int lane = threadIdx.x & 63, mt = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    // tiles are dealt round-robin: at any instant the resident workgroups read NEIGHBOURING tiles (172 KB apart,
    // spread over all HBM channels) rather than addresses a fixed 2^20-multiple apart (channel camping)
    const int64_t st = n_chunks;
    f32x16 acc[IT];
    acc_zero<IT>(acc);
    float bsum = 0.f;
    WgFrag<IT> f0;
    // Three register sets, software-pipelined by hand, ONE wave per SIMD: two tiles' loads (2 x (4+4*IT) x 1 KB per
    // wave) are in flight while a third tile's 16*IT MFMAs issue.  Two design facts measured on gfx950: (i) a second
    // wave on the SIMD does not hide a wave's non-MFMA issue time, only memory latency (tools/phase_trace.py; 8.5 ms at
    // two waves per SIMD vs 6.4 at one, for the warp nets), so this kernel wants exactly one wave per SIMD and hides
    // latency by prefetch depth instead of occupancy; (ii) the loads must be inline asm with counted s_waitcnt -- hipcc
    // re-rolls a C++-level multi-buffer loop into load -> vmcnt(0) -> MFMA -- and their 3 x (4+4*IT) x 4 destination
    // registers must all be architectural VGPRs (240 of 256 at IT = 4: there is no room for a second out-tile per wave).
    // Index-clamped prefetches past the end re-read a tile.
    constexpr int NL = 4 + 4 * IT;  // loads per tile
    const int64_t n_my = chunk < n_tiles ? (n_tiles - chunk + st - 1) / st : 0;
    const int64_t last = chunk + (n_my > 0 ? n_my - 1 : 0) * st;
#define WG_TILE(k) ((chunk + (k) * st) <= last ? (chunk + (k) * st) : last)
#define WG_LOAD(f, k) wg_load_async<IT>(f, acts, dpre, WG_TILE(k), acts_tile_floats, dpre_tile_floats, act_off, dpre_off, mt, i, h, in_live, out_live)
    if (n_my > 0) {
        WgFrag<IT> f1, f2;
        WG_LOAD(f0, 0);
        WG_LOAD(f1, 1);
        for (int64_t k = 0; k < n_my; k += 3) {
            WG_LOAD(f2, k + 2);
            wg_wait<2 * NL>();  // f0 landed
            landed(f0);
            wg_mma<IT>(f0, acc, bsum);
            __builtin_amdgcn_sched_barrier(0);
            WG_LOAD(f0, k + 3);
            wg_wait<2 * NL>();  // f1 landed
            landed(f1);
            if (k + 1 < n_my) wg_mma<IT>(f1, acc, bsum);
            __builtin_amdgcn_sched_barrier(0);
            WG_LOAD(f1, k + 4);
            wg_wait<2 * NL>();  // f2 landed
            landed(f2);
            if (k + 2 < n_my) wg_mma<IT>(f2, acc, bsum);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#undef WG_LOAD
#undef WG_TILE
    // D[row = out (acc_row), col = in (lane&31)];  partial of this chunk: [out_pad][32*IT]
    const int in_pad = 32 * IT;
    float *dw = dw_part + (int64_t)chunk * out_pad * in_pad;
#pragma unroll
    for (int n = 0; n < IT; n++)
#pragma unroll
        for (int r = 0; r < 16; r++) dw[(int64_t)(32 * mt + acc_row(r, h)) * in_pad + 32 * n + i] = acc[n][r];
    bsum += __shfl_xor(bsum, 32);
    if (h == 0) db_part[(int64_t)chunk * out_pad + 32 * mt + i] = bsum;
}

// ---- the same GEMM with exact fp32 products on the bf16 matrix pipe (see mlp_b3.hip) ---------------------------------
// K = points is a summation index, so the assignment of the tile's 32 points to (k16 step s, lane half g, element e) is
// free: point 16 g + 8 s + e keeps wg_load_async's pattern (lane (i, g) holds points 16g..16g+15 of row i of BOTH
// operands).  Each lane cuts its 16 + 16*IT loaded values into bf16 slices (the 4 waves of a block redo the split of the
// shared act tiles: the kernel is HBM-bound either way) and issues six slice products per (in tile, step).
template <int IT>
__device__ __forceinline__ void wg_mma_b3(const WgFrag<IT> &f, f32x16 (&acc)[IT], float &bsum) {
    Frag ah[2], am[2], al[2];
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
        for (int e2 = 0; e2 < 4; e2++) {
            const float v0 = f.a[2 * s + (e2 >> 1)][2 * (e2 & 1)], v1 = f.a[2 * s + (e2 >> 1)][2 * (e2 & 1) + 1];
            bsum += v0 + v1;
            split2(v0, v1, ah[s].u[e2], am[s].u[e2], al[s].u[e2]);
        }
#pragma unroll
    for (int np = 0; np < IT; np += 2) {
        constexpr int NP_MAX = 2;
        const int nn = (IT - np) < NP_MAX ? (IT - np) : NP_MAX;
        Frag bh[NP_MAX][2], bm[NP_MAX][2], bl[NP_MAX][2];
#pragma unroll
        for (int t = 0; t < NP_MAX; t++)
            if (t < nn)
#pragma unroll
                for (int s = 0; s < 2; s++)
#pragma unroll
                    for (int e2 = 0; e2 < 4; e2++)
                        split2(f.b[np + t][2 * s + (e2 >> 1)][2 * (e2 & 1)], f.b[np + t][2 * s + (e2 >> 1)][2 * (e2 & 1) + 1],
                               bh[t][s].u[e2], bm[t][s].u[e2], bl[t][s].u[e2]);
#pragma unroll
        for (int s = 0; s < 2; s++) {
#define WG_B3(A, B)                                                                                                      \
    _Pragma("unroll") for (int t = 0; t < NP_MAX; t++) if (t < nn)                                                       \
        acc[np + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[s].h, B[t][s].h, acc[np + t], 0, 0, 0)
            WG_B3(al, bh);
            WG_B3(am, bm);
            WG_B3(ah, bl);
            WG_B3(am, bh);
            WG_B3(ah, bm);
            WG_B3(ah, bh);
#undef WG_B3
        }
    }
}

template <int IT>
__device__ __forceinline__ void wgrad_body_b3(const float *__restrict__ acts, const float *__restrict__ dpre,
                                              int64_t acts_tile_floats, int64_t dpre_tile_floats, int act_off, int dpre_off,
                                              int out_pad, float *__restrict__ dw_part, float *__restrict__ db_part,
                                              int64_t n_tiles, int n_chunks, int chunk, int in_live, int out_live) {
    const int lane = threadIdx.x & 63, mt = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int64_t st = n_chunks;
    f32x16 acc[IT];
    acc_zero<IT>(acc);
    float bsum = 0.f;
    // two register sets (the slices take the room of wgrad_body's third): one tile's loads in flight under the other's
    // slicing + MFMAs
    constexpr int NL = 4 + 4 * IT;
    const int64_t n_my = chunk < n_tiles ? (n_tiles - chunk + st - 1) / st : 0;
    const int64_t last = chunk + (n_my > 0 ? n_my - 1 : 0) * st;
#define WG_TILE(k) ((chunk + (k) * st) <= last ? (chunk + (k) * st) : last)
#define WG_LOAD(f, k) wg_load_async<IT>(f, acts, dpre, WG_TILE(k), acts_tile_floats, dpre_tile_floats, act_off, dpre_off, mt, i, h, in_live, out_live)
    if (n_my > 0) {
        WgFrag<IT> f0, f1;
        WG_LOAD(f0, 0);
        for (int64_t k = 0; k < n_my; k += 2) {
            WG_LOAD(f1, k + 1);
            wg_wait<NL>();  // f0 landed
            landed(f0);
            wg_mma_b3<IT>(f0, acc, bsum);
            __builtin_amdgcn_sched_barrier(0);
            WG_LOAD(f0, k + 2);
            wg_wait<NL>();  // f1 landed
            landed(f1);
            if (k + 1 < n_my) wg_mma_b3<IT>(f1, acc, bsum);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#undef WG_LOAD
#undef WG_TILE
    const int in_pad = 32 * IT;
    float *dw = dw_part + (int64_t)chunk * out_pad * in_pad;
#pragma unroll
    for (int n = 0; n < IT; n++)
#pragma unroll
        for (int r = 0; r < 16; r++) dw[(int64_t)(32 * mt + acc_row(r, h)) * in_pad + 32 * n + i] = acc[n][r];
    bsum += __shfl_xor(bsum, 32);
    if (h == 0) db_part[(int64_t)chunk * out_pad + 32 * mt + i] = bsum;
}

// ---- b3 weight gradients of a 128-row layer, every operand value sliced ONCE per workgroup ---------------------------
// wgrad_body_b3's four waves walk the SAME tile sequence and each slices all IT activation tiles itself (440 VALU
// instructions per tile and wave, serial with its 48 MFMAs on a one-wave SIMD).  In the large-batch kernel below wave mt
// loads ONLY its own dPre row block and activation row block mt, keeps its dPre slices in registers (nobody else needs
// them) and publishes the slices of activation block mt in a double-buffered LDS area from which all four waves take
// ready-made B fragments: one barrier and 176 slicing instructions per wave and tile.  (Forms that were measured and
// removed -- raw tiles through an LDS-DMA ring with every wave slicing everything, 5.4-5.5 ms for the warp group; the same
// ring with slice-once, 5.54 ms; asm-loaded register sets, wrong at large batches -- are described in DESIGN.md section 3.)
struct WgSl {
    Frag h[2], m[2], l[2];
};
__device__ __forceinline__ void wg_slice16(const f32x4 (&v)[4], WgSl &o) {
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
        for (int e2 = 0; e2 < 4; e2++)
            split2(v[2 * s + (e2 >> 1)][2 * (e2 & 1)], v[2 * s + (e2 >> 1)][2 * (e2 & 1) + 1], o.h[s].u[e2], o.m[s].u[e2], o.l[s].u[e2]);
}

// The raw rows are prefetched into REGISTERS by ordinary (compiler-visible) loads: wave mt loads only its own dPre row block
// and activation row block mt (8 KB per tile), WG_REG_SETS register sets deep; hipcc tracks these loads itself (its s_waitcnt
// before a set's first use counts the younger sets), the scheduling fences keep the loads at the head of each step.
struct WgRaw {
    f32x4 a[4], b[4];
};
__device__ __forceinline__ void wg_raw_load(WgRaw &f, const float *__restrict__ a, const float *__restrict__ b) {
    const f32x4 *a4 = reinterpret_cast<const f32x4 *>(a), *b4 = reinterpret_cast<const f32x4 *>(b);
#ifdef WG_NT_LOADS   // A/B, measured: weight gradients 4.2 -> 6.6 ms.  Every parked row is read once by one wave, but a lane takes
                     // its 64 bytes of a row as four consecutive dwordx4 loads, i.e. an instruction touches 32 bytes of each of 32
                     // lines and the other three find the line in the cache -- which a non-temporal load does not leave there.
                     // (A streaming read gains 10 % from nt, tools/micro/hbm_read.hip; using it here needs lane-linear loads,
                     // i.e. the A operand through LDS as well.)
#pragma unroll
    for (int j = 0; j < 4; j++) f.a[j] = __builtin_nontemporal_load(a4 + j);
#pragma unroll
    for (int j = 0; j < 4; j++) f.b[j] = __builtin_nontemporal_load(b4 + j);
#else
#pragma unroll
    for (int j = 0; j < 4; j++) f.a[j] = a4[j];
#pragma unroll
    for (int j = 0; j < 4; j++) f.b[j] = b4[j];
#endif
}
// measured on cfg3 (warp group, same box): 4 sets / one workgroup per CU 5.28 ms, 5 sets 5.25, 3 sets / TWO workgroups per CU
// (240 registers, 2 x 48 KB of LDS: the second workgroup's MFMAs fill the first one's slicing and barrier time) 4.95
#ifndef WG_REG_SETS
#define WG_REG_SETS 3
#endif
#ifndef WG_REG_WAVES
#define WG_REG_WAVES 2        // waves per SIMD the register budget is sized for = workgroups per CU
#endif
template <int IT>
__device__ __forceinline__ void wgrad_regs_b3_body(const float *__restrict__ acts, const float *__restrict__ dpre,
                                                   int64_t acts_tile_floats, int64_t dpre_tile_floats, int act_off, int dpre_off,
                                                   float *__restrict__ dw_part, float *__restrict__ db_part, int64_t n_tiles,
                                                   int n_chunks, int chunk, int in_live) {
    // B slices: [buffer 2][in tile IT][plane 3][step 2][lane 64] float4
    constexpr int BUF_F4 = IT * 6 * 64;
    const int lane = threadIdx.x & 63, mt = threadIdx.x >> 6;
    const int i = lane & 31, g = lane >> 5;
    const int64_t st = n_chunks;
    const int64_t n_my = chunk < n_tiles ? (n_tiles - chunk + st - 1) / st : 0;
    const int64_t last = chunk + (n_my > 0 ? n_my - 1 : 0) * st;
    const int bt = mt & (IT - 1);            // activation block this wave loads (waves >= IT reload one, publish nothing)
    const float *a0 = dpre + dpre_off + (int64_t)(32 * mt + i) * TILE + 16 * g;
    const float *b0 = acts + act_off + (int64_t)wg_row(32 * bt + i, in_live) * TILE + 16 * g;      // (the 128 dPre rows of these layers are all live)
    f32x16 acc[IT];
    acc_zero<IT>(acc);
    float bsum = 0.f;
    constexpr int NS = WG_REG_SETS;           // register sets of raw rows: NS - 1 tiles (8 KB per wave each) in flight
    WgRaw raw[NS];
    WgSl as[2];
#define WG_T(k) ((chunk + (k) * st) <= last ? (chunk + (k) * st) : last)
#define WG_LD(set, k)                                                                \
    do {                                                                             \
        const int64_t t_ = WG_T(k);                                                  \
        wg_raw_load(raw[set], a0 + t_ * dpre_tile_floats, b0 + t_ * acts_tile_floats); \
    } while (0)
    // slice raw set `set` (tile k): dPre slices -> as[k & 1], activation slices -> LDS buffer k & 1
#define WG_SPLIT(set, par, REAL)                                                                                          \
    do {                                                                                                            \
        if (REAL) { _Pragma("unroll") for (int j = 0; j < 4; j++) bsum += (raw[set].a[j][0] + raw[set].a[j][1]) + (raw[set].a[j][2] + raw[set].a[j][3]); } \
        wg_slice16(raw[set].a, as[par]);                                                                            \
        if (IT == 4 || mt < IT) {                                                                                              \
            WgSl bs_;                                                                                               \
            wg_slice16(raw[set].b, bs_);                                                                            \
            f32x4 *dst_ = lds_res + (par) * BUF_F4 + mt * 6 * 64 + lane;                                            \
            _Pragma("unroll") for (int s = 0; s < 2; s++) {                                                         \
                dst_[(0 * 2 + s) * 64] = bs_.h[s].f;                                                                \
                dst_[(1 * 2 + s) * 64] = bs_.m[s].f;                                                                \
                dst_[(2 * 2 + s) * 64] = bs_.l[s].f;                                                                \
            }                                                                                                       \
        }                                                                                                           \
    } while (0)
#define WG_MMA(par)                                                                                                 \
    do {                                                                                                            \
        const f32x4 *src_ = lds_res + (par) * BUF_F4 + lane;                                                        \
        _Pragma("unroll") for (int np = 0; np < IT; np += 2) {                                                      \
            Frag bh_[2][2], bm_[2][2], bl_[2][2];                                                                   \
            _Pragma("unroll") for (int t = 0; t < 2; t++) _Pragma("unroll") for (int s = 0; s < 2; s++) {           \
                bh_[t][s].f = src_[((np + t) * 6 + 0 * 2 + s) * 64];                                                \
                bm_[t][s].f = src_[((np + t) * 6 + 1 * 2 + s) * 64];                                                \
                bl_[t][s].f = src_[((np + t) * 6 + 2 * 2 + s) * 64];                                                \
            }                                                                                                       \
            _Pragma("unroll") for (int s = 0; s < 2; s++) {                                                         \
                _Pragma("unroll") for (int t = 0; t < 2; t++) acc[np + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as[par].l[s].h, bh_[t][s].h, acc[np + t], 0, 0, 0); \
                _Pragma("unroll") for (int t = 0; t < 2; t++) acc[np + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as[par].m[s].h, bm_[t][s].h, acc[np + t], 0, 0, 0); \
                _Pragma("unroll") for (int t = 0; t < 2; t++) acc[np + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as[par].h[s].h, bl_[t][s].h, acc[np + t], 0, 0, 0); \
                _Pragma("unroll") for (int t = 0; t < 2; t++) acc[np + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as[par].m[s].h, bh_[t][s].h, acc[np + t], 0, 0, 0); \
                _Pragma("unroll") for (int t = 0; t < 2; t++) acc[np + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as[par].h[s].h, bm_[t][s].h, acc[np + t], 0, 0, 0); \
                _Pragma("unroll") for (int t = 0; t < 2; t++) acc[np + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as[par].h[s].h, bh_[t][s].h, acc[np + t], 0, 0, 0); \
            }                                                                                                       \
        }                                                                                                           \
    } while (0)
    // one pipeline step for tile k = k0 + J: raw sets rotate mod 4, slice sets / LDS buffers mod 2
#define WG_STEP(J)                                                                          \
    do {                                                                                    \
        __syncthreads();                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        WG_LD((J) % NS, k0 + (J) + NS);                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        if (k0 + (J) < n_my) WG_MMA((J) & 1);                                               \
        WG_SPLIT(((J) + 1) % NS, ((J) + 1) & 1, (k0 + (J) + 1 < n_my));                                             \
        __builtin_amdgcn_sched_barrier(0);                                                  \
    } while (0)
    // The same step with nothing conditional in it (every tile of the trip is real, so MFMAs and the NEXT tile's slicing sit in one
    // basic block) and the two streams INTERLEAVED by scheduling groups: one MFMA, then WG_FILL slicing instructions, 12 IT times.
    // Round 5 (tools/micro/mfma_valu_gap.hip): up to 5-6 single-issue instructions ride free in the 32-cycle shadow of a bf16 MFMA
    // *of the same wave*; the step used to run its 48 MFMAs and then its ~200 slicing instructions back to back and left the
    // overlap to whatever the SIMD's other wave happened to be doing.
#ifndef WG_FILL
#define WG_FILL 4
#endif
#define WG_STEP_FAST(J)                                                                     \
    do {                                                                                    \
        __syncthreads();                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        WG_LD((J) % NS, k0 + (J) + NS);                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        WG_MMA((J) & 1);                                                                    \
        WG_SPLIT(((J) + 1) % NS, ((J) + 1) & 1, true);                                      \
        if (WG_FILL > 0) {                                                                  \
            _Pragma("unroll") for (int g_ = 0; g_ < 12 * IT; g_++) {                        \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                          \
                __builtin_amdgcn_sched_group_barrier(0x002, WG_FILL, 0);                    \
            }                                                                               \
        }                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                  \
    } while (0)
    if (n_my > 0) {
#pragma unroll
        for (int q = 0; q < NS; q++) WG_LD(q, q);
        __builtin_amdgcn_sched_barrier(0);
        WG_SPLIT(0, 0, true);
        // the set index has period NS, the slice / B-buffer parity period 2: one loop trip = lcm(NS, 2) steps
        constexpr int TRIP = (NS % 2 ? 2 * NS : NS);
        int64_t k0 = 0;
        if (IT == 4) {           // (4 waves = 4 activation blocks: every wave publishes, `mt < IT` folds away)
            for (; k0 + TRIP < n_my; k0 += TRIP) {      // every step's tile AND its successor are real
                WG_STEP_FAST(0);
                WG_STEP_FAST(1);
                WG_STEP_FAST(2);
                WG_STEP_FAST(3);
                if (TRIP > 4) {
                    WG_STEP_FAST(4);
                    WG_STEP_FAST(5);
                }
                if (TRIP > 6) {
                    WG_STEP_FAST(6);
                    WG_STEP_FAST(7);
                    WG_STEP_FAST(8);
                    WG_STEP_FAST(9);
                }
            }
        }
        for (; k0 < n_my; k0 += TRIP) {
            WG_STEP(0);
            WG_STEP(1);
            WG_STEP(2);
            WG_STEP(3);
            if ((NS % 2 ? 2 * NS : NS) > 4) {
                WG_STEP(4);
                WG_STEP(5);
            }
            if ((NS % 2 ? 2 * NS : NS) > 6) {
                WG_STEP(6);
                WG_STEP(7);
                WG_STEP(8);
                WG_STEP(9);
            }
        }
    }
#undef WG_STEP
#undef WG_STEP_FAST
#undef WG_MMA
#undef WG_SPLIT
#undef WG_LD
#undef WG_T
    const int in_pad = 32 * IT;
    float *dw = dw_part + (int64_t)chunk * 128 * in_pad;
#pragma unroll
    for (int n = 0; n < IT; n++)
#pragma unroll
        for (int r = 0; r < 16; r++) dw[(int64_t)(32 * mt + acc_row(r, g)) * in_pad + 32 * n + i] = acc[n][r];
    bsum += __shfl_xor(bsum, 32);
    if (g == 0) db_part[(int64_t)chunk * 128 + 32 * mt + i] = bsum;
}

template <int IT>
__global__ __launch_bounds__(256, WG_REG_WAVES) void wgrad_regs_b3_kernel(const float *__restrict__ acts, const float *__restrict__ dpre,
                                                                int64_t acts_tile_floats, int64_t dpre_tile_floats, int act_off,
                                                                int dpre_off, float *__restrict__ dw_part,
                                                                float *__restrict__ db_part, int64_t n_tiles, int n_chunks, int in_live) {
    wgrad_regs_b3_body<IT>(acts, dpre, acts_tile_floats, dpre_tile_floats, act_off, dpre_off, dw_part, db_part, n_tiles, n_chunks,
                           (int)blockIdx.x, in_live);
}

// ---- layer 4 of a warp net with dPre4 REGENERATED instead of read (round 6) ------------------------------------
// dPre4 = relu'(H5) (W5^T dPre5), and dPre5 is the incoming gradient itself: three (two) numbers per point.  The backward-data kernel
// wrote those 128 rows per tile and net (16 KB of its 172) for this kernel alone to read them back: 4.3 GB of a cfg3 step's 57.  Here
// wave mt makes ITS 32 rows of dPre4 from the gradient (12 B per point), the ReLU sign words the forward parked (8 B per point and
// layer) and the T5 slices (loop-invariant registers), with the backward-data kernel's own instruction sequence -- so that what
// reaches the weight-gradient MFMAs is bit for bit what that kernel would have parked:
//   * chain (mlp_b3.hip, b3_layer<2, 4, true> on T5):  D[row = feature][col = point] = sum of six slice products A = W5^T slices,
//     B = dPre5 slices, k16 step 0 (step 1 is all zeros).  Here the SAME registers in swapped roles: A' = the lane's dPre5 slices
//     (lane (i, g) = point i, k = 8g .. 8g + 7: the gradient in lanes g = 0, e = 0..2), B' = the T5 fragment of tile mt (lane (i, g) =
//     feature i, the same k) give D'[row = point][col = feature] = D^T: the same products summed over the same k positions.
//   * D' leaves lane (i, g) with points acc_row(r, g); the weight-gradient MFMAs want points 16 g .. 16 g + 15 (the order in which the
//     parked rows were read): eight v_permlane32_swap exchange the two halves' register groups.
//   * the sign words are parked per chain lane (point, half): 16 ballots hand every feature lane the 32-point word of its row.
struct WgRegenRaw {
    f32x4 b[4];        // the activation row block (H4 rows 32 bt + i, points 16 g ..)
    uint2 m;           // H5's ReLU sign words of chain lane `lane` (point lane & 31, half lane >> 5)
    float gv[3];       // the incoming gradient of point lane & 31 (lanes g = 0)
};

__global__ __launch_bounds__(256, WG_REG_WAVES) void wgrad_regen_b3_kernel(
    const float *__restrict__ acts, const float *__restrict__ g5, int nout, int64_t M, const f32x4 *__restrict__ w3T,
    int64_t acts_tile_floats, int act_off, int mask_off, float *__restrict__ dw_part, float *__restrict__ db_part, int64_t n_tiles,
    int n_chunks) {
    constexpr int IT = 4, NS = 3, BUF_F4 = IT * 6 * 64;
    const int chunk = (int)blockIdx.x;
    const int lane = threadIdx.x & 63, mt = threadIdx.x >> 6;
    const int i = lane & 31, g = lane >> 5;
    const int64_t st = n_chunks;
    const int64_t n_my = chunk < n_tiles ? (n_tiles - chunk + st - 1) / st : 0;
    const int64_t last = chunk + (n_my > 0 ? n_my - 1 : 0) * st;
    const float *b0 = acts + act_off + (int64_t)(32 * mt + i) * TILE + 16 * g;
    const uint2 *m0 = reinterpret_cast<const uint2 *>(acts + mask_off) + lane;
    // T5 fragments of output tile mt, k16 step 0: [plane][tile][step][lane] (packing.py; mlp_b3.hip: b3_layer<2, 4>)
    Frag wh, wm, wl;
    wh.f = w3T[0 * 512 + (mt * 2) * 64 + lane];
    wm.f = w3T[1 * 512 + (mt * 2) * 64 + lane];
    wl.f = w3T[2 * 512 + (mt * 2) * 64 + lane];
    // which ballot carries this lane's row: feature i = acc_row(r', h') of tile mt
    const int my_r = (i & 3) + 4 * (i >> 3), my_h = (i >> 2) & 1;
    f32x16 acc[IT];
    acc_zero<IT>(acc);
    float bsum = 0.f;
    WgRegenRaw raw[NS];
    WgSl as[2];
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#define RG_T(k) ((chunk + (k) * st) <= last ? (chunk + (k) * st) : last)
#define RG_LD(set, k)                                                                                   \
    do {                                                                                                \
        const int64_t t_ = RG_T(k);                                                                     \
        const f32x4 *b4_ = reinterpret_cast<const f32x4 *>(b0 + t_ * acts_tile_floats);                 \
        _Pragma("unroll") for (int j = 0; j < 4; j++) raw[set].b[j] = b4_[j];                           \
        raw[set].m = m0[t_ * (acts_tile_floats / 2)];                                                   \
        const int64_t p_ = t_ * TILE + i;                                                               \
        const bool on_ = g5 && g == 0 && p_ < M;                                                        \
        raw[set].gv[0] = on_ ? g5[p_ * nout + 0] : 0.f;                                                 \
        raw[set].gv[1] = on_ ? g5[p_ * nout + 1] : 0.f;                                                 \
        raw[set].gv[2] = (on_ && nout == 3) ? g5[p_ * nout + 2] : 0.f;                                  \
    } while (0)
    // raw set -> dPre4 rows (regenerated) -> slices as[par]; activation rows -> slices in LDS buffer par
#define RG_SPLIT(set, par, REAL)                                                                                              \
    do {                                                                                                                      \
        Frag dh_, dm_, dl_;                                                                                                   \
        split2(raw[set].gv[0], raw[set].gv[1], dh_.u[0], dm_.u[0], dl_.u[0]);                                                 \
        split2(raw[set].gv[2], 0.f, dh_.u[1], dm_.u[1], dl_.u[1]);                                                            \
        dh_.u[2] = dh_.u[3] = dm_.u[2] = dm_.u[3] = dl_.u[2] = dl_.u[3] = 0u;                                                 \
        f32x16 D_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dh_.h, wl.h, zero16, 0, 0, 0);                                   \
        D_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dm_.h, wm.h, D_, 0, 0, 0);                                               \
        D_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dl_.h, wh.h, D_, 0, 0, 0);                                               \
        D_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dh_.h, wm.h, D_, 0, 0, 0);                                               \
        D_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dm_.h, wh.h, D_, 0, 0, 0);                                               \
        D_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dh_.h, wh.h, D_, 0, 0, 0);                                               \
        /* the 32-point sign word of this lane's row: ballot r carries rows acc_row(r, 0) (low half) and acc_row(r, 1) (high) */ \
        const uint32_t w16_ = ((mt < 2 ? raw[set].m.x : raw[set].m.y) >> (16 * (mt & 1))) & 0xffffu;                          \
        uint32_t rowm_ = 0u;                                                                                                  \
        _Pragma("unroll") for (int r = 0; r < 16; r++) {                                                                      \
            const uint64_t bal_ = __builtin_amdgcn_ballot_w64(((w16_ >> r) & 1u) != 0u);                                      \
            const uint32_t v_ = my_h ? (uint32_t)(bal_ >> 32) : (uint32_t)bal_;                                               \
            rowm_ = my_r == r ? v_ : rowm_;                                                                                   \
        }                                                                                                                     \
        rowm_ >>= 16 * g;                                                                                                     \
        /* points acc_row(r, g) -> points 16 g + 0..15 */                                                                     \
        f32x4 a_[4];                                                                                                          \
        _Pragma("unroll") for (int c = 0; c < 4; c++) {                                                                       \
            const auto s0_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(D_[c]), __float_as_uint(D_[8 + c]), false, false);      \
            const auto s1_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(D_[4 + c]), __float_as_uint(D_[12 + c]), false, false); \
            a_[0][c] = mask_bit(rowm_, 0 + c, __uint_as_float(s0_[0]));                                                       \
            a_[1][c] = mask_bit(rowm_, 4 + c, __uint_as_float(s0_[1]));                                                       \
            a_[2][c] = mask_bit(rowm_, 8 + c, __uint_as_float(s1_[0]));                                                       \
            a_[3][c] = mask_bit(rowm_, 12 + c, __uint_as_float(s1_[1]));                                                      \
        }                                                                                                                     \
        if (REAL) { _Pragma("unroll") for (int j = 0; j < 4; j++) bsum += (a_[j][0] + a_[j][1]) + (a_[j][2] + a_[j][3]); }    \
        wg_slice16(a_, as[par]);                                                                                              \
        WgSl bs_;                                                                                                             \
        wg_slice16(raw[set].b, bs_);                                                                                          \
        f32x4 *dst_ = lds_res + (par) * BUF_F4 + mt * 6 * 64 + lane;                                                          \
        _Pragma("unroll") for (int s = 0; s < 2; s++) {                                                                       \
            dst_[(0 * 2 + s) * 64] = bs_.h[s].f;                                                                              \
            dst_[(1 * 2 + s) * 64] = bs_.m[s].f;                                                                              \
            dst_[(2 * 2 + s) * 64] = bs_.l[s].f;                                                                              \
        }                                                                                                                     \
    } while (0)
#define RG_MMA(par)                                                                                                 \
    do {                                                                                                            \
        const f32x4 *src_ = lds_res + (par) * BUF_F4 + lane;                                                        \
        _Pragma("unroll") for (int np = 0; np < IT; np += 2) {                                                      \
            Frag bh_[2][2], bm_[2][2], bl_[2][2];                                                                   \
            _Pragma("unroll") for (int t = 0; t < 2; t++) _Pragma("unroll") for (int s = 0; s < 2; s++) {           \
                bh_[t][s].f = src_[((np + t) * 6 + 0 * 2 + s) * 64];                                                \
                bm_[t][s].f = src_[((np + t) * 6 + 1 * 2 + s) * 64];                                                \
                bl_[t][s].f = src_[((np + t) * 6 + 2 * 2 + s) * 64];                                                \
            }                                                                                                       \
            _Pragma("unroll") for (int s = 0; s < 2; s++) {                                                         \
                _Pragma("unroll") for (int t = 0; t < 2; t++) acc[np + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as[par].l[s].h, bh_[t][s].h, acc[np + t], 0, 0, 0); \
                _Pragma("unroll") for (int t = 0; t < 2; t++) acc[np + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as[par].m[s].h, bm_[t][s].h, acc[np + t], 0, 0, 0); \
                _Pragma("unroll") for (int t = 0; t < 2; t++) acc[np + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as[par].h[s].h, bl_[t][s].h, acc[np + t], 0, 0, 0); \
                _Pragma("unroll") for (int t = 0; t < 2; t++) acc[np + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as[par].m[s].h, bh_[t][s].h, acc[np + t], 0, 0, 0); \
                _Pragma("unroll") for (int t = 0; t < 2; t++) acc[np + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as[par].h[s].h, bm_[t][s].h, acc[np + t], 0, 0, 0); \
                _Pragma("unroll") for (int t = 0; t < 2; t++) acc[np + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as[par].h[s].h, bh_[t][s].h, acc[np + t], 0, 0, 0); \
            }                                                                                                       \
        }                                                                                                           \
    } while (0)
#define RG_STEP(J)                                                                          \
    do {                                                                                    \
        __syncthreads();                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        RG_LD((J) % NS, k0 + (J) + NS);                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        if (k0 + (J) < n_my) RG_MMA((J) & 1);                                               \
        RG_SPLIT(((J) + 1) % NS, ((J) + 1) & 1, (k0 + (J) + 1 < n_my));                     \
        __builtin_amdgcn_sched_barrier(0);                                                  \
    } while (0)
    if (n_my > 0) {
#pragma unroll
        for (int q = 0; q < NS; q++) RG_LD(q, q);
        __builtin_amdgcn_sched_barrier(0);
        RG_SPLIT(0, 0, true);
        // (the guarded step only.  Measured and dropped, same box: an unguarded main loop as in wgrad_regs_b3_kernel -- 73 spilled
        // registers; the same with the MFMA block fenced from the slicing and the T5 fragments in LDS -- 6 spills, +0.05 ms per launch)
        for (int64_t k0 = 0; k0 < n_my; k0 += 6) {        // lcm(NS, 2) steps per trip
            RG_STEP(0);
            RG_STEP(1);
            RG_STEP(2);
            RG_STEP(3);
            RG_STEP(4);
            RG_STEP(5);
        }
    }
#undef RG_STEP
#undef RG_MMA
#undef RG_SPLIT
#undef RG_LD
#undef RG_T
    float *dw = dw_part + (int64_t)chunk * 128 * 128;
#pragma unroll
    for (int n = 0; n < IT; n++)
#pragma unroll
        for (int r = 0; r < 16; r++) dw[(int64_t)(32 * mt + acc_row(r, g)) * 128 + 32 * n + i] = acc[n][r];
    bsum += __shfl_xor(bsum, 32);
    if (g == 0) db_part[(int64_t)chunk * 128 + 32 * mt + i] = bsum;
}

template <int IT, bool B3 = false>
__global__ __launch_bounds__(256, 1) void wgrad_kernel(const float *__restrict__ acts, const float *__restrict__ dpre,
                                                       int64_t acts_tile_floats, int64_t dpre_tile_floats, int act_off,
                                                       int dpre_off, int out_pad, float *__restrict__ dw_part,
                                                       float *__restrict__ db_part, int64_t n_tiles, int n_chunks, int in_live,
                                                       int out_live) {
    if (B3)
        wgrad_body_b3<IT>(acts, dpre, acts_tile_floats, dpre_tile_floats, act_off, dpre_off, out_pad, dw_part, db_part, n_tiles,
                          n_chunks, (int)blockIdx.x, in_live, out_live);
    else
        wgrad_body<IT>(acts, dpre, acts_tile_floats, dpre_tile_floats, act_off, dpre_off, out_pad, dw_part, db_part, n_tiles,
                       n_chunks, (int)blockIdx.x, in_live, out_live);
}

// All layers of a net group in ONE launch: block b belongs to the layer whose block range holds it.  Used for small
// batches, where one launch per layer (12 for the warp nets) is mostly ramp-up, drain and launch gaps; big batches keep
// the per-layer launches (see mh_mlp_wgrad).
#define WG_MAX_LAYERS 16
struct WgAll {
    int32_t n;
    int32_t first_block[WG_MAX_LAYERS + 1];
    int32_t act_off[WG_MAX_LAYERS], dpre_off[WG_MAX_LAYERS], out_pad[WG_MAX_LAYERS], in_tiles[WG_MAX_LAYERS];
    int32_t in_live[WG_MAX_LAYERS], out_live[WG_MAX_LAYERS];      // rows of the input / dPre tile that carry values (wg_row)
    int32_t chunks[WG_MAX_LAYERS];
    int64_t dw_off[WG_MAX_LAYERS], db_off[WG_MAX_LAYERS];
};

template <bool B3>
__global__ __launch_bounds__(256, 1) void wgrad_all_kernel(const float *__restrict__ acts, const float *__restrict__ dpre,
                                                           int64_t acts_tile_floats, int64_t dpre_tile_floats,
                                                           float *__restrict__ ws, WgAll d, int64_t n_tiles) {
    int l = 0;
    while (l < d.n - 1 && (int)blockIdx.x >= d.first_block[l + 1]) l++;
    const int chunk = (int)blockIdx.x - d.first_block[l];
    if ((int)(threadIdx.x >> 6) >= d.out_pad[l] / 32) return;      // layers with fewer than four 32-row output tiles
    float *dw = ws + d.dw_off[l], *db = ws + d.db_off[l];
#define WG_CASE(IT)                                                                                                          \
    if (B3)                                                                                                                  \
        wgrad_body_b3<IT>(acts, dpre, acts_tile_floats, dpre_tile_floats, d.act_off[l], d.dpre_off[l], d.out_pad[l], dw, db,    \
                          n_tiles, d.chunks[l], chunk, d.in_live[l], d.out_live[l]);                                         \
    else                                                                                                                     \
        wgrad_body<IT>(acts, dpre, acts_tile_floats, dpre_tile_floats, d.act_off[l], d.dpre_off[l], d.out_pad[l], dw, db,       \
                       n_tiles, d.chunks[l], chunk, d.in_live[l], d.out_live[l])
    switch (d.in_tiles[l]) {
        case 1: WG_CASE(1); break;
        case 2: WG_CASE(2); break;
        case 3: WG_CASE(3); break;
        default: WG_CASE(4); break;
    }
#undef WG_CASE
}

// The 128-row layers (128 x 128 and 128 x 64) of a small batch in ONE launch of the slice-once body (wgrad_regs_b3_body: two
// workgroups per CU, every operand value sliced once per workgroup) instead of wgrad_body_b3, whose four waves each slice all
// activation tiles: the four warp weight-gradient calls of a training step 0.79 -> 0.68 ms (same box).  The narrow output layers stay with
// wgrad_all_kernel (one wave per 32-row tile, no LDS).
__global__ __launch_bounds__(256, WG_REG_WAVES) void wgrad_all_regs_b3_kernel(const float *__restrict__ acts,
                                                                                const float *__restrict__ dpre,
                                                                                int64_t acts_tile_floats, int64_t dpre_tile_floats,
                                                                                float *__restrict__ ws, WgAll d, int64_t n_tiles) {
    int l = 0;
    while (l < d.n - 1 && (int)blockIdx.x >= d.first_block[l + 1]) l++;
    const int chunk = (int)blockIdx.x - d.first_block[l];
    float *dw = ws + d.dw_off[l], *db = ws + d.db_off[l];
    if (d.in_tiles[l] == 4)
        wgrad_regs_b3_body<4>(acts, dpre, acts_tile_floats, dpre_tile_floats, d.act_off[l], d.dpre_off[l], dw, db, n_tiles, d.chunks[l], chunk, d.in_live[l]);
    else
        wgrad_regs_b3_body<2>(acts, dpre, acts_tile_floats, dpre_tile_floats, d.act_off[l], d.dpre_off[l], dw, db, n_tiles, d.chunks[l], chunk, d.in_live[l]);
}

// sum the per-chunk partials of every layer in one launch
struct WgReduce {
    int32_t n;
    int32_t accumulate;                    // 0: out = sum of the partials; 1: out += sum (a later call of a shared accumulator)
    int32_t chunks[2 * WG_MAX_LAYERS];
    int32_t len[2 * WG_MAX_LAYERS];
    int64_t part_off[2 * WG_MAX_LAYERS];
    int64_t out_off[2 * WG_MAX_LAYERS];
    int64_t first[2 * WG_MAX_LAYERS + 1];  // prefix of len: element ranges of the launch
};

// 64 consecutive output elements per workgroup (coalesced 256 B rows of the partials); the chunk axis is split over
// the 4 waves and each lane keeps 4 independent partial sums, so 16 loads per lane are in flight -- a single serial
// loop over the 256-1024 chunks (one thread per element) ran at 0.15-0.4 TB/s and cost 0.7 ms per step.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ part, float *__restrict__ out, WgReduce d) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * 64 + lane;
    float acc = 0.f;
    int s = 0;
    int64_t j = 0;
    const bool live = e < d.first[d.n];
    if (live) {
        while (e >= d.first[s + 1]) s++;
        j = e - d.first[s];
        const float *p = part + d.part_off[s] + j;
        const int64_t len = d.len[s];
        const int chunks = d.chunks[s];
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int c = wave;
        for (; c + 12 < chunks; c += 16) {
            a0 += p[(int64_t)c * len];
            a1 += p[(int64_t)(c + 4) * len];
            a2 += p[(int64_t)(c + 8) * len];
            a3 += p[(int64_t)(c + 12) * len];
        }
        for (; c < chunks; c += 4) a0 += p[(int64_t)c * len];
        acc = (a0 + a1) + (a2 + a3);
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && live) {
        const float sum = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        float *o = out + d.out_off[s] + j;
        *o = d.accumulate ? *o + sum : sum;
    }
}

// =====================================================================================
// C ABI
// =====================================================================================
static inline int64_t n_tiles_for(int64_t M) { return ((M + BLOCK_PTS - 1) / BLOCK_PTS) * 4; }

extern "C" int64_t mh_warp_acts_floats(int64_t M) { return n_tiles_for(M) * (int64_t)(WARP_ACT_ROWS * TILE); }
extern "C" int64_t mh_warp_dpre_floats(int64_t M) { return n_tiles_for(M) * (int64_t)(WARP_DPRE_ROWS * TILE); }
extern "C" int64_t mh_warp_wpack_floats(void) { return WARP_NET_WPACK; }
extern "C" int64_t mh_warp_wpackT_floats(void) { return WARP_NET_WPACKT; }
extern "C" int64_t mh_field_acts_floats(int64_t M) { return n_tiles_for(M) * (int64_t)(FIELD_ACT_ROWS * TILE); }
extern "C" int64_t mh_field_wpack_floats(void) { return FIELD_WPACK; }
extern "C" int64_t mh_field_wpackT_floats(void) { return FIELD_WPACKT; }
extern "C" int64_t mh_mlp_tiles(int64_t M) { return n_tiles_for(M); }

extern "C" int mh_warp_fwd(const float *x, const int32_t *slot, const float *bias0_d, const float *bias0_t,
                           const float *wpack_d, const float *wpack_t, const float *bias_d, const float *bias_t,
                           int32_t n_bands, float *out_deform, float *out_topo, float *acts, int64_t M, void *stream) {
    if (M == 0) return MH_OK;
    if (M < 0 || !x || !bias0_d || !bias0_t || !wpack_d || !wpack_t || !bias_d || !bias_t || !out_deform || !out_topo ||
        n_bands < 0 || n_bands > 6)
        return MH_ERR_ARG;
    const int64_t blocks = (M + BLOCK_PTS - 1) / BLOCK_PTS;
    if (blocks > 0x7fffffffLL) return MH_ERR_ARG;
#ifdef MH_PHASE_TRACE
    // trace build only: MH_TRACE_DYNLDS=<bytes> of unused dynamic LDS limits the workgroups per CU (occupancy experiments)
    const char *dl = getenv("MH_TRACE_DYNLDS");
    const size_t dyn_lds = dl ? (size_t)atoi(dl) : 0;
    if (dyn_lds) hipFuncSetAttribute((const void *)warp_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_lds);
#else
    const size_t dyn_lds = 0;
#endif
    hipLaunchKernelGGL(warp_fwd_kernel, dim3((unsigned)blocks), dim3(256), dyn_lds, mh_stream(stream), x, slot, bias0_d,
                       bias0_t, wpack_d, wpack_t, bias_d, bias_t, (int)n_bands, out_deform, out_topo, acts, M);
    MH_CHECK_LAUNCH();
    return MH_OK;
}

extern "C" int mh_warp_bwd_data(const float *x, const float *g_deform, const float *g_topo, const float *wpackT_d,
                                const float *wpackT_t, int32_t n_bands, const float *acts, float *dpre, float *g_x,
                                int64_t M, void *stream) {
    if (M == 0) return MH_OK;
    if (M < 0 || !x || !wpackT_d || !wpackT_t || !acts || !dpre || n_bands < 0 || n_bands > 6) return MH_ERR_ARG;
    const int64_t blocks = (M + BLOCK_PTS - 1) / BLOCK_PTS;
    if (blocks > 0x7fffffffLL) return MH_ERR_ARG;
    hipLaunchKernelGGL(warp_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, mh_stream(stream), x, g_deform, g_topo,
                       wpackT_d, wpackT_t, (int)n_bands, acts, dpre, g_x, M);
    MH_CHECK_LAUNCH();
    return MH_OK;
}

// persistent field kernels: one 8-wave block per CU (the resident weights take 94-98 KB of its LDS)
static inline unsigned field_blocks(int64_t n_tiles) {
    const int64_t need = (n_tiles + FIELD_THREADS / 64 - 1) / (FIELD_THREADS / 64);
    const int64_t cus = mh_cu_count();
    return (unsigned)(need < cus ? need : cus);
}

// the resident-weight kernels need more than the default 64 KB of dynamic LDS: opt in once per kernel, not per launch
static int field_lds_opt_in() {
    static MhOncePerDevice done;
    const int dev = mh_device();
    if (done.need(dev)) {
        if (hipFuncSetAttribute((const void *)field_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(FIELD_WPACK * sizeof(float))) != hipSuccess)
            return MH_ERR_LAUNCH;
        done.mark(dev);
    }
    return MH_OK;
}

extern "C" int mh_field_fwd(const float *xc, const float *feat_s, const float *feat_c, const float *topo,
                            const float *wpack, const float *bias, const float *beta, int32_t n_bands, int32_t with_color,
                            float *sdf, float *sigma, float *albedo, float *acts, int64_t M, void *stream) {
    if (M == 0) return MH_OK;
    if (M < 0 || !xc || !feat_s || !wpack || !bias || !sdf || n_bands < 0 || n_bands > 6 || !beta) return MH_ERR_ARG;
    if (with_color && (!feat_c || !albedo)) return MH_ERR_ARG;
    const int64_t n_tiles = n_tiles_for(M);  // dead tail tiles are processed too: wgrad reads every scratch tile
    const size_t lds = (size_t)FIELD_WPACK * sizeof(float);
    if (field_lds_opt_in() != MH_OK) return MH_ERR_LAUNCH;
    hipLaunchKernelGGL(field_fwd_kernel, dim3(field_blocks(n_tiles)), dim3(FIELD_THREADS), lds, mh_stream(stream), xc, feat_s,
                       feat_c, topo, wpack, bias, beta, (int)n_bands, (int)with_color, sdf, sigma, albedo, acts, M, n_tiles);
    MH_CHECK_LAUNCH();
    return MH_OK;
}

#define WG_PER_LAYER_TILES 16384      // from this many 32-point tiles on: one launch per layer, large-batch kernels
#ifndef WG_MERGED_REGS_TILES
#define WG_MERGED_REGS_TILES 256      // from this many tiles on (below the line above): the merged launch uses the slice-once body
#endif
static inline int wg_chunks(int out_pad, int64_t n_tiles, int n_layers) {
    // 4 waves per CU per launch = exactly one per SIMD (see wgrad_kernel) whatever the number of output tiles
    // large batches (the per-layer launches, from 16 384 tiles on): WG_REG_WAVES workgroups per CU (wgrad_regs_b3_kernel)
    int64_t c = (int64_t)(4 * ((n_tiles >= WG_PER_LAYER_TILES && out_pad == 128) ? WG_REG_WAVES : 1) * mh_cu_count()) / (out_pad / 32);
    if (n_tiles < WG_PER_LAYER_TILES) {
        // small batches (one merged launch for all layers: 12 x chunks workgroups): every chunk writes a full partial dW that the
        // reduction reads back -- at 700 tiles (the 22 000-point calls of a training step) 256 chunks per layer are 164 MB of
        // partials against 250 MB of operands.  At least 8 tiles per chunk, but never fewer than 32 chunks per layer.
        const int64_t cap = n_tiles / 8 > 32 ? n_tiles / 8 : 32;
        if (c > cap) c = cap;
        // ... and the merged launch needs no more than ~4 workgroups per CU over ALL its layers: at 4 400 tiles 12 x 256 chunks
        // wrote 168 MB of partials (44 us of reduction) where 12 x 85 fill the chip as well
        const int64_t fill = (4 * (int64_t)mh_cu_count() + n_layers - 1) / (n_layers > 0 ? n_layers : 1);
        if (c > fill && fill >= 32) c = fill;
    }
    if (c > n_tiles) c = n_tiles;
    return (int)(c < 1 ? 1 : c);
}

extern "C" int64_t mh_mlp_wgrad_workspace_floats(int32_t n_layers, const int32_t *in_feats_host,
                                                 const int32_t *out_feats_host, int64_t n_tiles) {
    int64_t tot = 0;
    for (int l = 0; l < n_layers; l++)
        tot += (int64_t)wg_chunks(out_feats_host[l], n_tiles, n_layers) * ((int64_t)in_feats_host[l] * out_feats_host[l] + out_feats_host[l]);
    return tot;
}

// layers whose dPre rows the launch regenerates instead of reading (wgrad_regen_b3_kernel): the warp nets' layer 4
struct WgRegenHost {
    bool on[WG_MAX_LAYERS];
    const float *g5[WG_MAX_LAYERS];       // the net's incoming gradient [M, nout] (NULL: none, dPre4 = 0)
    int nout[WG_MAX_LAYERS];
    const f32x4 *w3T[WG_MAX_LAYERS];      // the net's transposed slices (T5 first)
    int mask_off[WG_MAX_LAYERS];          // floats from the tile's start to H5's ReLU sign words of the net
    int64_t M;
};

static int wgrad_impl(const float *acts, const float *dpre, int64_t acts_tile_floats, int64_t dpre_tile_floats,
                      int32_t n_layers, const int32_t *act_off_host, const int32_t *dpre_off_host,
                      const int32_t *in_feats_host, const int32_t *out_feats_host, const int32_t *in_live_host,
                      const int32_t *out_live_host, float *workspace, float *dw_raw, float *db_raw, int64_t n_tiles, void *stream,
                      bool b3, const WgRegenHost *regen = nullptr) {
    if (n_tiles == 0 || n_layers == 0) return MH_OK;
    if (!acts || !dpre || !act_off_host || !dpre_off_host || !in_feats_host || !out_feats_host || !workspace || !dw_raw ||
        !db_raw || n_layers < 0 || n_layers > WG_MAX_LAYERS || n_tiles < 0)
        return MH_ERR_ARG;
    for (int l = 0; l < n_layers; l++) {
        const int in = in_feats_host[l], out = out_feats_host[l];
        if (in <= 0 || out <= 0 || (in % 32) || (out % 32) || in > 128 || out > 128) return MH_ERR_ARG;
        if (in_live_host && (in_live_host[l] < 1 || in_live_host[l] > in)) return MH_ERR_ARG;
        if (out_live_host && (out_live_host[l] < 1 || out_live_host[l] > out)) return MH_ERR_ARG;
        // the slice-once kernels take every dPre row of their 128-row layers
        if (out_live_host && out == 128 && out_live_host[l] != 128) return MH_ERR_ARG;
    }
    // workspace: [dW partials of layer 0 | 1 | ...][db partials of layer 0 | 1 | ...]; outputs: dw_raw | db_raw
    WgReduce rd;
    rd.n = 2 * n_layers;
    rd.accumulate = 0;
    int64_t woff = 0, dw_out = 0, db_out = 0;
    int64_t dw_poff[WG_MAX_LAYERS], db_poff[WG_MAX_LAYERS];
    for (int l = 0; l < n_layers; l++) {
        dw_poff[l] = woff;
        woff += (int64_t)wg_chunks(out_feats_host[l], n_tiles, n_layers) * in_feats_host[l] * out_feats_host[l];
    }
    for (int l = 0; l < n_layers; l++) {
        db_poff[l] = woff;
        woff += (int64_t)wg_chunks(out_feats_host[l], n_tiles, n_layers) * out_feats_host[l];
    }
    int64_t dw_total = 0;
    for (int l = 0; l < n_layers; l++) dw_total += (int64_t)in_feats_host[l] * out_feats_host[l];
    if (db_raw != dw_raw + dw_total) return MH_ERR_ARG;  // dw_raw and db_raw must be one contiguous buffer (checked BEFORE any launch)
    rd.first[0] = 0;
    // One launch per layer for big batches, ONE launch for all layers for small ones (measured on MI355X, same box):
    // at 2 M points the merged launch is slower (warp nets 6.6 vs 6.0 ms: workgroups of different layers stream
    // different 2 GB regions at once), at the 2 k-140 k-point calls of a training step it is faster (1.11 vs 1.28 ms
    // per step: the twelve launches are mostly ramp-up and drain).
    const bool per_layer = n_tiles >= WG_PER_LAYER_TILES;
    WgAll all;
    all.n = n_layers;
    all.first_block[0] = 0;
    for (int l = 0; l < n_layers; l++) {
        const int in = in_feats_host[l], out = out_feats_host[l];
        const int chunks = wg_chunks(out, n_tiles, n_layers);
        all.act_off[l] = act_off_host[l];
        all.dpre_off[l] = dpre_off_host[l];
        all.out_pad[l] = out;
        all.in_tiles[l] = in / 32;
        const int in_live = in_live_host ? in_live_host[l] : in, out_live = out_live_host ? out_live_host[l] : out;
        all.in_live[l] = in_live;
        all.out_live[l] = out_live;
        all.chunks[l] = chunks;
        all.dw_off[l] = dw_poff[l];
        all.db_off[l] = db_poff[l];
        all.first_block[l + 1] = all.first_block[l] + chunks;
        if (per_layer && b3 && out == 128 && in == 128 && regen && regen->on[l]) {
            hipLaunchKernelGGL(wgrad_regen_b3_kernel, dim3((unsigned)chunks), dim3(256), 2 * 4 * 6 * 1024, mh_stream(stream), acts,
                               regen->g5[l], regen->nout[l], regen->M, regen->w3T[l], acts_tile_floats, (int)act_off_host[l],
                               regen->mask_off[l], workspace + dw_poff[l], workspace + db_poff[l], n_tiles, chunks);
            MH_CHECK_LAUNCH();
        } else if (per_layer && b3 && out == 128 && (in == 128 || in == 64)) {
            if (in == 128)
                hipLaunchKernelGGL(wgrad_regs_b3_kernel<4>, dim3((unsigned)chunks), dim3(256), 2 * 4 * 6 * 1024, mh_stream(stream),
                                   acts, dpre, acts_tile_floats, dpre_tile_floats, (int)act_off_host[l], (int)dpre_off_host[l],
                                   workspace + dw_poff[l], workspace + db_poff[l], n_tiles, chunks, in_live);
            else
                hipLaunchKernelGGL(wgrad_regs_b3_kernel<2>, dim3((unsigned)chunks), dim3(256), 2 * 2 * 6 * 1024, mh_stream(stream),
                                   acts, dpre, acts_tile_floats, dpre_tile_floats, (int)act_off_host[l], (int)dpre_off_host[l],
                                   workspace + dw_poff[l], workspace + db_poff[l], n_tiles, chunks, in_live);
            MH_CHECK_LAUNCH();
        } else if (per_layer) {
            const dim3 grid((unsigned)chunks), block((unsigned)(out / 32) * 64);
#define WG_LAUNCH(IT)                                                                                         \
    if (b3)                                                                                                   \
        hipLaunchKernelGGL((wgrad_kernel<IT, true>), grid, block, 0, mh_stream(stream), acts, dpre, acts_tile_floats, \
                           dpre_tile_floats, (int)act_off_host[l], (int)dpre_off_host[l], out, workspace + dw_poff[l], \
                           workspace + db_poff[l], n_tiles, chunks, in_live, out_live);                       \
    else                                                                                                      \
        hipLaunchKernelGGL((wgrad_kernel<IT, false>), grid, block, 0, mh_stream(stream), acts, dpre, acts_tile_floats, \
                           dpre_tile_floats, (int)act_off_host[l], (int)dpre_off_host[l], out, workspace + dw_poff[l], \
                           workspace + db_poff[l], n_tiles, chunks, in_live, out_live)
            switch (in / 32) {
                case 1: WG_LAUNCH(1); break;
                case 2: WG_LAUNCH(2); break;
                case 3: WG_LAUNCH(3); break;
                default: WG_LAUNCH(4); break;
            }
#undef WG_LAUNCH
            MH_CHECK_LAUNCH();
        }
        // reduction segments: dW of layer l, then (second half) db of layer l; both outputs live in ONE buffer:
        // out = dw_raw for e < dw_total, db_raw is addressed relative to dw_raw via out_off (caller passes them contiguous)
        rd.chunks[l] = chunks;
        rd.len[l] = in * out;
        rd.part_off[l] = dw_poff[l];
        rd.out_off[l] = dw_out;
        dw_out += (int64_t)in * out;
        rd.chunks[n_layers + l] = chunks;
        rd.len[n_layers + l] = out;
        rd.part_off[n_layers + l] = db_poff[l];
        rd.out_off[n_layers + l] = dw_total + db_out;
        db_out += out;
    }
    if (!per_layer && b3 && n_tiles >= WG_MERGED_REGS_TILES) {
        // two merged launches: the 128-row layers on the slice-once body, the rest (the narrow output layers) as before
        WgAll wide, rest;
        wide.n = rest.n = 0;
        wide.first_block[0] = rest.first_block[0] = 0;
        for (int l = 0; l < n_layers; l++) {
            WgAll &t = (all.out_pad[l] == 128 && (all.in_tiles[l] == 4 || all.in_tiles[l] == 2)) ? wide : rest;
            const int k = t.n++;
            t.act_off[k] = all.act_off[l];
            t.dpre_off[k] = all.dpre_off[l];
            t.out_pad[k] = all.out_pad[l];
            t.in_tiles[k] = all.in_tiles[l];
            t.in_live[k] = all.in_live[l];
            t.out_live[k] = all.out_live[l];
            t.chunks[k] = all.chunks[l];
            t.dw_off[k] = all.dw_off[l];
            t.db_off[k] = all.db_off[l];
            t.first_block[k + 1] = t.first_block[k] + all.chunks[l];
        }
        if (wide.n) {
            hipLaunchKernelGGL(wgrad_all_regs_b3_kernel, dim3((unsigned)wide.first_block[wide.n]), dim3(256), 2 * 4 * 6 * 1024,
                               mh_stream(stream), acts, dpre, acts_tile_floats, dpre_tile_floats, workspace, wide, n_tiles);
            MH_CHECK_LAUNCH();
        }
        if (rest.n) {
            hipLaunchKernelGGL(wgrad_all_kernel<true>, dim3((unsigned)rest.first_block[rest.n]), dim3(256), 0, mh_stream(stream), acts, dpre,
                               acts_tile_floats, dpre_tile_floats, workspace, rest, n_tiles);
            MH_CHECK_LAUNCH();
        }
    } else if (!per_layer) {
        if (b3)
            hipLaunchKernelGGL(wgrad_all_kernel<true>, dim3((unsigned)all.first_block[n_layers]), dim3(256), 0, mh_stream(stream),
                               acts, dpre, acts_tile_floats, dpre_tile_floats, workspace, all, n_tiles);
        else
            hipLaunchKernelGGL(wgrad_all_kernel<false>, dim3((unsigned)all.first_block[n_layers]), dim3(256), 0, mh_stream(stream),
                               acts, dpre, acts_tile_floats, dpre_tile_floats, workspace, all, n_tiles);
        MH_CHECK_LAUNCH();
    }
    for (int s = 0; s < rd.n; s++) rd.first[s + 1] = rd.first[s] + rd.len[s];
    const int64_t total = rd.first[rd.n];
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, mh_stream(stream), workspace,
                       dw_raw, rd);
    MH_CHECK_LAUNCH();
    return MH_OK;
}

extern "C" int mh_mlp_wgrad(const float *acts, const float *dpre, int64_t acts_tile_floats, int64_t dpre_tile_floats,
                            int32_t n_layers, const int32_t *act_off_host, const int32_t *dpre_off_host,
                            const int32_t *in_feats_host, const int32_t *out_feats_host, const int32_t *in_live_host,
                            const int32_t *out_live_host, float *workspace, float *dw_raw, float *db_raw, int64_t n_tiles,
                            void *stream) {
    return wgrad_impl(acts, dpre, acts_tile_floats, dpre_tile_floats, n_layers, act_off_host, dpre_off_host, in_feats_host,
                      out_feats_host, in_live_host, out_live_host, workspace, dw_raw, db_raw, n_tiles, stream, false);
}

extern "C" int mh_mlp_wgrad_b3(const float *acts, const float *dpre, int64_t acts_tile_floats, int64_t dpre_tile_floats,
                               int32_t n_layers, const int32_t *act_off_host, const int32_t *dpre_off_host,
                               const int32_t *in_feats_host, const int32_t *out_feats_host, const int32_t *in_live_host,
                               const int32_t *out_live_host, float *workspace, float *dw_raw, float *db_raw, int64_t n_tiles,
                               void *stream) {
    return wgrad_impl(acts, dpre, acts_tile_floats, dpre_tile_floats, n_layers, act_off_host, dpre_off_host, in_feats_host,
                      out_feats_host, in_live_host, out_live_host, workspace, dw_raw, db_raw, n_tiles, stream, true);
}

// ---- the warp nets' weight gradients with their geometry on this side of the ABI ----------------------------------------------------
// (the 12 layers of deform_net + topo_net in the tiles of mh_warp_fwd / mh_warp_bwd_data; morpheus_amd/ops.py used to spell the
// offsets out).  regen_dpre4: the two layer-4 launches regenerate dPre4 (wgrad_regen_b3_kernel) -- the caller ran
// mh_warp_bwd_data_b3 with skip_dpre4 = 1, and both decisions come from mh_warp_regen_dpre4(M).
struct WarpWg {
    int32_t act_off[12], dpre_off[12], in[12], out[12], in_live[12], out_live[12];
};
static WarpWg warp_wg_geometry() {
    WarpWg w;
    for (int net = 0; net < 2; net++)
        for (int l = 0; l < 6; l++) {
            const int k = net * 6 + l;
            w.act_off[k] = l == 0 ? 0 : (64 + net * 640 + (l - 1) * 128) * TILE;
            w.dpre_off[k] = (net * 672 + l * 128) * TILE;
            w.in[k] = l == 0 ? 64 : 128;
            w.out[k] = l == 5 ? 32 : 128;
            w.in_live[k] = l == 0 ? 40 : 128;
            w.out_live[k] = l == 5 ? (net ? 2 : 3) : 128;
        }
    return w;
}

// dPre4 is regenerated by the large-batch (one launch per layer) weight-gradient path only
extern "C" int32_t mh_warp_regen_dpre4(int64_t M) { return M > 0 && n_tiles_for(M) >= WG_PER_LAYER_TILES ? 1 : 0; }

extern "C" int64_t mh_warp_wgrad_workspace_floats(int64_t M) {
    const WarpWg w = warp_wg_geometry();
    return mh_mlp_wgrad_workspace_floats(12, w.in, w.out, n_tiles_for(M));
}

extern "C" int mh_warp_wgrad_b3(const float *acts, const float *dpre, const float *g_deform, const float *g_topo, const void *w3T_d,
                                const void *w3T_t, int32_t regen_dpre4, float *workspace, float *dw_raw, float *db_raw, int64_t M,
                                void *stream) {
    if (M == 0) return MH_OK;
    if (M < 0 || !w3T_d || !w3T_t) return MH_ERR_ARG;
    if (regen_dpre4 && !mh_warp_regen_dpre4(M)) return MH_ERR_ARG;      // (the small-batch launches read dPre4)
    const WarpWg w = warp_wg_geometry();
    WgRegenHost rg;
    for (int k = 0; k < WG_MAX_LAYERS; k++) rg.on[k] = false;
    rg.M = M;
    for (int net = 0; net < 2; net++) {
        const int k = net * 6 + 4;
        rg.on[k] = regen_dpre4 != 0;
        rg.g5[k] = net ? g_topo : g_deform;
        rg.nout[k] = net ? 2 : 3;
        rg.w3T[k] = reinterpret_cast<const f32x4 *>(net ? w3T_t : w3T_d);
        rg.mask_off[k] = WARP_HID_ROWS * TILE + (net * 5 + 4) * 64 * 2;
    }
    return wgrad_impl(acts, dpre, (int64_t)WARP_ACT_ROWS * TILE, (int64_t)WARP_DPRE_ROWS * TILE, 12, w.act_off, w.dpre_off, w.in, w.out,
                      w.in_live, w.out_live, workspace, dw_raw, db_raw, n_tiles_for(M), stream, true, regen_dpre4 ? &rg : nullptr);
}

// ---- fused field backward (backward-data + weight gradients, see field_fused_*_kernel) ---------------------------------
static const int FUSED_IN[6] = {96, 64, 64, 64, 64, 64};    // raw-gradient layer order: s0, s1, s2, c0, c1, c2
static const int FUSED_OUT[6] = {64, 64, 64, 64, 64, 32};

static inline int fused_blocks(int64_t n_tiles) {
    const int64_t need = (n_tiles + FUSED_THREADS / 64 - 1) / (FUSED_THREADS / 64);
    const int64_t cus = mh_cu_count();
    return (int)(need < cus ? need : cus);
}

extern "C" int64_t mh_field_bwd_fused_workspace_floats(int64_t M) {
    const int64_t chunks = (int64_t)fused_blocks(n_tiles_for(M));     // one partial per workgroup
    int64_t per = 0;
    for (int l = 0; l < 6; l++) per += (int64_t)FUSED_IN[l] * FUSED_OUT[l] + FUSED_OUT[l];
    return chunks * (per + 1);                                        // + the d(beta) partial of each workgroup
}
extern "C" int64_t mh_field_dgeo_floats(int64_t M) { return n_tiles_for(M) * 64 * 16; }

static int field_bwd_fused_impl(const float *xc, const float *sdf, const float *albedo, const float *g_sdf,
                                const float *g_sigma, const float *g_albedo, const float *wpackT, const float *beta,
                                int32_t n_bands, int32_t with_color, const float *acts, float *dgeo_scratch,
                                float *workspace, float *raw, int32_t accumulate, float *g_xc, float *g_feat_s, float *g_feat_c,
                                float *g_topo, uint32_t *gmax_bits, int64_t M, void *stream, bool b3) {
    if (M == 0) return MH_OK;
    if (M < 0 || !xc || !sdf || !wpackT || !acts || !workspace || !raw || n_bands < 0 || n_bands > 6 || !beta) return MH_ERR_ARG;
    if (with_color && (!albedo || !dgeo_scratch)) return MH_ERR_ARG;
    static MhOncePerDevice opted;
    const size_t lds_c = (size_t)(FUSED_TC2(b3) + FUSED_TC1(b3) + FUSED_TC0(b3)) * 16 + (size_t)(4 * SCR_FLOATS) * sizeof(float);
    const size_t lds_s = (size_t)(FUSED_TS2(b3) + FUSED_TS1(b3) + FUSED_TS0(b3)) * 16 + (size_t)(4 * SCR_FLOATS) * sizeof(float);
    const int dev = mh_device();
    if (opted.need(dev)) {
        const int big_c = (int)((size_t)(FUSED_TC2(1) + FUSED_TC1(1) + FUSED_TC0(1)) * 16 + (size_t)(4 * SCR_FLOATS) * sizeof(float));
        const int big_s = (int)((size_t)(FUSED_TS2(1) + FUSED_TS1(1) + FUSED_TS0(1)) * 16 + (size_t)(4 * SCR_FLOATS) * sizeof(float));
        if (hipFuncSetAttribute((const void *)field_fused_color_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, big_c) != hipSuccess ||
            hipFuncSetAttribute((const void *)field_fused_color_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, big_c) != hipSuccess ||
            hipFuncSetAttribute((const void *)field_fused_sdf_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big_s) != hipSuccess ||
            hipFuncSetAttribute((const void *)field_fused_sdf_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big_s) != hipSuccess ||
            hipFuncSetAttribute((const void *)field_fused_sdf_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big_s) != hipSuccess ||
            hipFuncSetAttribute((const void *)field_fused_sdf_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big_s) != hipSuccess)
            return MH_ERR_LAUNCH;
        opted.mark(dev);
    }
    const int64_t n_tiles = n_tiles_for(M);
    const int blocks = fused_blocks(n_tiles);
    const int64_t chunks = blocks;                  // one partial per workgroup (its four waves add up through LDS)
    // workspace: [dW partials s0 | s1 | s2 | c0 | c1 | c2][db partials s0 | ... | c2], each [chunks][...]
    int64_t dw_off[6], db_off[6], off = 0, dw_total = 0, db_total = 0;
    for (int l = 0; l < 6; l++) {
        dw_off[l] = off;
        off += chunks * FUSED_IN[l] * FUSED_OUT[l];
        dw_total += (int64_t)FUSED_IN[l] * FUSED_OUT[l];
    }
    for (int l = 0; l < 6; l++) {
        db_off[l] = off;
        off += chunks * FUSED_OUT[l];
        db_total += FUSED_OUT[l];
    }
    FusedPart pc, ps;
    for (int k = 0; k < 3; k++) {
        ps.dw[k] = dw_off[k];
        ps.db[k] = db_off[k];
        pc.dw[k] = dw_off[3 + k];
        pc.db[k] = db_off[3 + k];
    }
    ps.gb = off;                                    // [chunks] d(beta) partials behind the bias partials
    pc.gb = off;
    ps.dw_s2 = pc.dw_s2 = dw_off[2];                // the sdf net's last layer: out tile 0 (geo rows) from the colour launch,
    ps.db_s2 = pc.db_s2 = db_off[2];                // out tile 1 (the sdf row) from the sdf launch
    hipStream_t st = mh_stream(stream);
#define FUSED_LAUNCH_COLOR(B3_)                                                                                             \
    hipLaunchKernelGGL(field_fused_color_kernel<B3_>, dim3((unsigned)blocks), dim3(FUSED_THREADS), lds_c, st, albedo, g_albedo, wpackT, \
                       acts, dgeo_scratch, g_feat_c, workspace, pc, gmax_bits, M, n_tiles)
#define FUSED_LAUNCH_SDF(WC, B3_, DGEO)                                                                                      \
    hipLaunchKernelGGL((field_fused_sdf_kernel<WC, B3_>), dim3((unsigned)blocks), dim3(FUSED_THREADS), lds_s, st, xc, sdf, g_sdf, \
                       g_sigma, wpackT, beta, (int)n_bands, acts, (const float *)(DGEO), g_xc, g_feat_s, g_topo,             \
                       workspace, ps, gmax_bits, M, n_tiles)
    if (with_color) {
        if (b3) FUSED_LAUNCH_COLOR(true); else FUSED_LAUNCH_COLOR(false);
        MH_CHECK_LAUNCH();
        if (b3) FUSED_LAUNCH_SDF(true, true, dgeo_scratch); else FUSED_LAUNCH_SDF(true, false, dgeo_scratch);
    } else {
        if (b3) FUSED_LAUNCH_SDF(false, true, nullptr); else FUSED_LAUNCH_SDF(false, false, nullptr);
    }
#undef FUSED_LAUNCH_COLOR
#undef FUSED_LAUNCH_SDF
    MH_CHECK_LAUNCH();
    // reduce the per-workgroup partials into raw = [dW s0..c2 | db s0..c2] (mh_mlp_wgrad's output format).  On the sdf-only pass the
    // colour net's segments are reduced over ZERO chunks, i.e. written as 0 by the same launch (no separate memsets)
    const int n_l = with_color ? 6 : 3;
    WgReduce rd;
    rd.n = 13;
    rd.accumulate = accumulate ? 1 : 0;
    int64_t dw_out = 0, db_out = 0;
    rd.first[0] = 0;
    for (int l = 0; l < 6; l++) {
        const int32_t ch = l < n_l ? (int32_t)chunks : 0;
        rd.chunks[l] = ch;
        rd.len[l] = FUSED_IN[l] * FUSED_OUT[l];
        rd.part_off[l] = dw_off[l];
        rd.out_off[l] = dw_out;
        rd.chunks[6 + l] = ch;
        rd.len[6 + l] = FUSED_OUT[l];
        rd.part_off[6 + l] = db_off[l];
        rd.out_off[6 + l] = dw_total + db_out;
        dw_out += (int64_t)FUSED_IN[l] * FUSED_OUT[l];
        db_out += FUSED_OUT[l];
    }
    rd.chunks[12] = (int32_t)chunks;                // d(beta): one element behind the 24 928 weight / bias gradients
    rd.len[12] = 1;
    rd.part_off[12] = ps.gb;
    rd.out_off[12] = dw_total + db_total;
    for (int sgm = 0; sgm < rd.n; sgm++) rd.first[sgm + 1] = rd.first[sgm] + rd.len[sgm];
    const int64_t total = rd.first[rd.n];
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st, workspace, raw, rd);
    MH_CHECK_LAUNCH();
    return MH_OK;
}

extern "C" int mh_field_bwd_fused(const float *xc, const float *sdf, const float *albedo, const float *g_sdf,
                                  const float *g_sigma, const float *g_albedo, const float *wpackT, const float *beta,
                                  int32_t n_bands, int32_t with_color, const float *acts, float *dgeo_scratch,
                                  float *workspace, float *raw, int32_t accumulate, float *g_xc, float *g_feat_s,
                                  float *g_feat_c, float *g_topo, uint32_t *gmax_bits, int64_t M, void *stream) {
    return field_bwd_fused_impl(xc, sdf, albedo, g_sdf, g_sigma, g_albedo, wpackT, beta, n_bands, with_color, acts, dgeo_scratch,
                                workspace, raw, accumulate, g_xc, g_feat_s, g_feat_c, g_topo, gmax_bits, M, stream, false);
}

// The same pass with exact fp32 products from three bf16 slices on the bf16 matrix pipe: w3T = the sliced TRANSPOSED pack of the
// six field layers (packing.py field_joint_packer().b3T_layers: TC2 | TC1 | TC0 | TS2 | TS1 | TS0).  Same arguments otherwise.
extern "C" int64_t mh_field_w3T_bytes(void) {
    return (int64_t)(FUSED_TC2(1) + FUSED_TC1(1) + FUSED_TC0(1) + FUSED_TS2(1) + FUSED_TS1(1) + FUSED_TS0(1)) * 16;
}
extern "C" int mh_field_bwd_fused_b3(const float *xc, const float *sdf, const float *albedo, const float *g_sdf,
                                     const float *g_sigma, const float *g_albedo, const void *w3T, const float *beta,
                                     int32_t n_bands, int32_t with_color, const float *acts, float *dgeo_scratch,
                                     float *workspace, float *raw, int32_t accumulate, float *g_xc, float *g_feat_s,
                                     float *g_feat_c, float *g_topo, uint32_t *gmax_bits, int64_t M, void *stream) {
    return field_bwd_fused_impl(xc, sdf, albedo, g_sdf, g_sigma, g_albedo, reinterpret_cast<const float *>(w3T), beta, n_bands,
                                with_color, acts, dgeo_scratch, workspace, raw, accumulate, g_xc, g_feat_s, g_feat_c, g_topo,
                                gmax_bits, M, stream, true);
}

extern "C" int mh_abi_version(void) { return MH_ABI_VERSION; }
extern "C" const char *mh_status_string(int status) {
    switch (status) {
        case MH_OK: return "ok";
        case MH_ERR_ARG: return "invalid argument (null pointer, bad size or unsupported configuration)";
        case MH_ERR_LAUNCH: return "kernel launch failed (hipGetLastError)";
        default: return "unknown status";
    }
}
