// The warp networks (deform_net + topo_net, models/model.py:412-437) with products from TWO fp16 slices per operand: the
// OPT-IN arithmetic mode "h2" (ops.set_mlp_mode / MORPHEUS_MLP=h2).  Its operands carry 22 of fp32's 24 significand bits at
// block scales, so it is NOT fp32-faithful and bench.py reports it beside the headline only; the default is mlp_b3.hip.
//
// Same idea as mlp_b3.hip -- fp32 operands cut into narrow slices, slice products exact on the 16-bit matrix pipe, fp32
// accumulation -- at half the matrix work: fp16 carries 11 significand bits, so
//      x . 2^k = h + l + r,   h = fp16(x . 2^k),  l = fp16(x . 2^k - h),  |r| <= 2^-22 |x . 2^k|
// and a product W.x is THREE slice products, Wh.xh + Wh.xl + Wl.xh, through v_mfma_f32_32x32x16_f16 (what is dropped, Wl.xl
// and the residuals r, is <= ~2^-22 of a product and unbiased; measured on layer-shaped data the representation error is
// 7e-8 relative, a third of the accumulation error a plain fp32 GEMM has: tools/h2_error_model.py;
// tests/test_gpu_ops.py::test_warp_sliced_arithmetic_against_float64 measures it against float64 on six operand distributions:
// forward values and d/dx at the fp32 kernels' own error, weight gradients 3-8x theirs -- DESIGN.md section 4).
//
// fp16 has five exponent bits, so the slices only work at the right scale; every operand carries a power-of-two scale that
// puts its largest magnitude in [2^14, 2^15) -- an exact operation in both directions:
//   * weights: one exponent per layer, from the layer's largest |w| (h2_amax_kernel -> a table behind the net's slices);
//   * activations / gradients: one exponent PER POINT, from the largest magnitude of the point's feature vector (the two
//     lanes that hold a point's column exchange their maxima).  An element then keeps all 22 bits while it is within 2^-17
//     of its vector's maximum, and an absolute error of 2^-39 of that maximum below.
//   * the weight-gradient kernel's operands (mlp.hip: wgrad_regs_h2_kernel): one exponent per parked TENSOR -- it contracts
//     over points -- from the maxima these kernels record while parking (amax table, mlp_dev.h: h2_note_amax).
// Accumulators run in the scaled domain, 2^(kw + kx) W x; a hidden layer's epilogue forms y = relu(acc . 2^-(kw + kx) + b) in one
// fma (the bias never enters the scaled domain; layer 0, whose bias row is per frame slot, pre-scales it into the
// accumulators instead), parks y as fp32 and cuts y . 2^kx' into the next operand.  Infinities and NaNs propagate as NaN, a
// zero vector clamps its exponent, the clamps keep every 2^-(kw + kx) a normal float.
//
// Everything else is mlp_b3.hip's scheme: register-resident chain (accumulator registers 8s'..8s'+7 of output tile t are the
// B operand of k16 step 2t + s' of the next layer), parked tiles and ReLU sign masks bit-for-bit in mlp.hip's layout (the
// weight-gradient kernels read them as fp32), weight slices [plane h|l][out tile][k16 step][lane][8 fp16] staged by LDS-DMA.
// A 128 x 128 layer is 64 KB of slices: TWO layers fit in LDS, so either layer l+1 is fetched while layer l computes and a
// layer costs one workgroup barrier (8-wave workgroup), or two independent 4-wave workgroups share a CU (see the kernels).
#include "mlp_dev.h"
#include <stdlib.h>

#define H2_L0_F4 1536                                    // 2 planes x 4 tiles x 3 k16 steps x 64 lanes
#define H2_LH_F4 4096                                    // 128 x 128: two k-half blocks [khalf][plane][tile][k16 step 0..3][lane]
#define H2_KH_F4 2048
#define H2_L5_F4 1024                                    // one padded output tile, 8 k16 steps
#define H2_TAB_F4 (H2_L0_F4 + 4 * H2_LH_F4 + H2_L5_F4)   // the net's scale table (largest |w| per layer, fp32 bits) sits here
#define H2_NET_F4 (H2_TAB_F4 + 512)
#define H2_T5_F4 1024                                    // 2 planes x 4 tiles x 2 k16 steps x 64 lanes
#define H2_T0_F4 2048                                    // 2 planes x 2 tiles x 8 k16 steps x 64 lanes
#define H2_TABT_F4 (H2_T5_F4 + 4 * H2_LH_F4 + H2_T0_F4)
#define H2_NETT_F4 (H2_TABT_F4 + 512)
#define H2_BUF_F4 (H2_LH_F4 + 64)                        // one staged layer + its bias row
#define H2_LDS_BYTES (2 * H2_BUF_F4 * 16)                // 133 120
extern __shared__ f32x4 lds_h2[];

// amax table (H2_AMAX_COPIES x 32 words, zeroed by the caller before the forward): largest magnitude of every parked row block,
// for the weight-gradient kernel's per-tensor scales (mlp_dev.h: h2_note_amax).  Word 0: encoding rows; 1 + 5 net + j: output of
// layer j (0..4); 16 + 6 net + l: dPre_l
#define H2_AMAX_H(net, j) (1 + 5 * (net) + (j))
#define H2_AMAX_D(net, l) (16 + 6 * (net) + (l))

#ifdef MH_PHASE_TRACE
// phase trace for tools/phase_trace_h2.py (never compiled into the product library): wave 0 of every 64th workgroup stamps
// s_memtime at the phase boundaries of the forward kernel's hidden layers of net 0 (8 slots per layer), s_memrealtime in 62/63
__device__ long long mh_h2_trace[256 * 64];
#define H2_STAMP(slot)                                                                          \
    do {                                                                                        \
        if ((threadIdx.x == 0) && (blockIdx.x % 64 == 0) && (blockIdx.x / 64 < 256))            \
            mh_h2_trace[(blockIdx.x / 64) * 64 + (slot)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
#define H2_STAMP_REAL(slot)                                                                     \
    do {                                                                                        \
        if ((threadIdx.x == 0) && (blockIdx.x % 64 == 0) && (blockIdx.x / 64 < 256))            \
            mh_h2_trace[(blockIdx.x / 64) * 64 + (slot)] = (long long)__builtin_amdgcn_s_memrealtime(); \
    } while (0)
extern "C" int mh_h2_trace_read(long long *dst_host) {
    return hipMemcpyFromSymbol(dst_host, HIP_SYMBOL(mh_h2_trace), sizeof(long long) * 256 * 64) == hipSuccess ? 0 : 2;
}
#else
#define H2_STAMP(slot) do { } while (0)
#define H2_STAMP_REAL(slot) do { } while (0)
#endif

template <int N_F4, int NW>
__device__ __forceinline__ void h2_stage(int buf, const f32x4 *__restrict__ src) {
    constexpr int NTHR = NW * 64;
    static_assert(N_F4 % NTHR == 0, "whole rounds of the block");
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < N_F4 / NTHR; k++)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + k * NTHR + threadIdx.x),
                                         (__attribute__((address_space(3))) void *)(lds_h2 + buf * H2_BUF_F4 + k * NTHR + wave * 64),
                                         16, 0, 0);
}
// the layer's bias row (<= 128 floats) rides along, into one of the buffer's two 32-float4 slots (layers alternate slots: with
// ONE weight buffer the next layer's fetch is in flight while this layer's epilogue still reads its bias)
__device__ __forceinline__ void h2_stage_bias(int buf, int slot, const float *__restrict__ bias, int n_f4) {
    if ((int)threadIdx.x < n_f4)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const f32x4 *>(bias) + threadIdx.x),
                                         (__attribute__((address_space(3))) void *)(lds_h2 + buf * H2_BUF_F4 + H2_LH_F4 + slot * 32 +
                                                                                    (threadIdx.x >> 6) * 64),
                                         16, 0, 0);
}
// my DMA pieces (and parking stores) are done; behind the barrier everyone's are visible and everyone has left the other buffer
__device__ __forceinline__ void h2_wait() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// acc[t] += W[t] . b : three slice products per k16 step, two output tiles in rotation, small terms first
template <int KS, int MT, bool ZERO = false>
__device__ __forceinline__ void h2_layer(const f32x4 *__restrict__ w, const FragH (&bh)[8], const FragH (&bl)[8], f32x16 (&acc)[MT],
                                         int lane) {
    constexpr int PL = MT * KS * 64;
    constexpr int NT = MT >= 2 ? 2 : 1;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; s++) {
#pragma unroll
        for (int mp = 0; mp < MT; mp += NT) {
            FragH ah[NT], al[NT];
#pragma unroll
            for (int t = 0; t < NT; t++) {
                ah[t].f = w[0 * PL + ((mp + t) * KS + s) * 64 + lane];
                al[t].f = w[1 * PL + ((mp + t) * KS + s) * 64 + lane];
            }
#pragma unroll
            for (int t = 0; t < NT; t++)
                acc[mp + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t].h, bh[s].h, (ZERO && s == 0) ? zero : acc[mp + t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[mp + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t].h, bl[s].h, acc[mp + t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[mp + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t].h, bh[s].h, acc[mp + t], 0, 0, 0);
        }
    }
}

// a 128 x 128 layer from its two k-half blocks; the accumulators start from the inline constant 0 (the bias joins in the
// epilogue's fma, backward layers have none)
__device__ __forceinline__ void h2_hidden(const f32x4 *__restrict__ w, const FragH (&bh)[8], const FragH (&bl)[8], f32x16 (&acc)[4],
                                          int lane) {
    constexpr int PLH = 4 * 4 * 64;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; s++) {
        const f32x4 *wk = w + (s >> 2) * H2_KH_F4 + (s & 3) * 64 + lane;
#pragma unroll
        for (int mp = 0; mp < 4; mp += 2) {
            FragH ah[2], al[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                ah[t].f = wk[0 * PLH + (mp + t) * 256];
                al[t].f = wk[1 * PLH + (mp + t) * 256];
            }
#pragma unroll
            for (int t = 0; t < 2; t++) acc[mp + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t].h, bh[s].h, s == 0 ? zero : acc[mp + t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 2; t++) acc[mp + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t].h, bl[s].h, acc[mp + t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 2; t++) acc[mp + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t].h, bh[s].h, acc[mp + t], 0, 0, 0);
        }
    }
}

// layer-0 epilogue (the per-slot bias row went into the accumulators).  acc = 2^ks (W x + b).  ReLU; y = acc . 2^-ks is parked (feature-major) and gives the sign mask; the
// point's next shift d comes from the largest acc; acc . 2^d = y . 2^(ks + d) is cut into the next layer's B operand slices.
// Returns the point's new exponent kx = ks + d (the next layer adds its weight exponent).
__device__ __forceinline__ int h2_epilogue(f32x16 (&acc)[4], int ks, float *__restrict__ ht, uint2 *__restrict__ mk, int pt, int h,
                                           FragH (&bh)[8], FragH (&bl)[8], uint32_t *__restrict__ amax, int slot) {
    int mi = 0;
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            acc[t][r] = relu_i(acc[t][r]);
            mi = max(mi, __float_as_int(acc[t][r]));
        }
    uint32_t mt[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        float y[16];
#pragma unroll
        for (int r = 0; r < 16; r++) y[r] = __builtin_ldexpf(acc[t][r], -ks);
        if (ht) {
#pragma unroll
            for (int r = 0; r < 16; r++) PARK_STORE(y[r], &ht[(32 * t + acc_row(r, h)) * TILE + pt]);
        }
        uint32_t m = 0;
#pragma unroll
        for (int r = 15; r >= 0; r--) m = push_nz(m, y[r]);
        mt[t] = m;
    }
    if (mk) *mk = make_uint2(mt[0] | (mt[1] << 16), mt[2] | (mt[3] << 16));
    if (amax) h2_note_amax(amax, slot, __float_as_int(__builtin_ldexpf(__int_as_float(mi), -ks)));
    const int d = h2_point_shift(mi, H2_FWD_CLAMP - ks);
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
            for (int e2 = 0; e2 < 4; e2++)
                split_h(__builtin_ldexpf(acc[t][8 * s2 + 2 * e2], d), __builtin_ldexpf(acc[t][8 * s2 + 2 * e2 + 1], d),
                        bh[2 * t + s2].u[e2], bl[2 * t + s2].u[e2]);
    return d + ks;
}

// hidden-layer epilogue.  acc = 2^ks W x, the bias is NOT inside: y = relu(acc . 2^-ks + b) is one fma (the product is an exact
// power-of-two scaling, the sum rounds once) and one integer max; y is parked and gives the sign mask; the point's next
// exponent kx comes from the largest y; y . 2^kx is cut into the next layer's B operand slices.  Returns kx.
__device__ __forceinline__ int h2_epilogue_hidden(f32x16 (&acc)[4], int ks, const f32x4 *__restrict__ brow, float *__restrict__ ht,
                                                  uint2 *__restrict__ mk, int pt, int h, FragH (&bh)[8], FragH (&bl)[8],
                                                  uint32_t *__restrict__ amax, int slot, int stamp = -1) {
    const float sc = __builtin_ldexpf(1.0f, -ks);
    int mi = 0;
    uint32_t mt[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++) {
            const f32x4 b = brow[8 * t + 2 * r4 + h];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const float y = relu_i(__builtin_fmaf(acc[t][4 * r4 + c], sc, b[c]));
                acc[t][4 * r4 + c] = y;
                mi = max(mi, __float_as_int(y));
            }
        }
        if (ht) {
#pragma unroll
            for (int r = 0; r < 16; r++) PARK_STORE(acc[t][r], &ht[(32 * t + acc_row(r, h)) * TILE + pt]);
        }
        uint32_t m = 0;
#pragma unroll
        for (int r = 15; r >= 0; r--) m = push_nz(m, acc[t][r]);
        mt[t] = m;
    }
    if (mk) *mk = make_uint2(mt[0] | (mt[1] << 16), mt[2] | (mt[3] << 16));
    if (stamp >= 0) {
        asm volatile("" ::"v"(mt[0]), "v"(mt[3]), "v"(mi));   // trace build only: the stamp waits for the values above
        H2_STAMP(stamp);
    }
    if (amax) h2_note_amax(amax, slot, mi);
    const int kx = h2_point_shift(mi, H2_FWD_CLAMP);
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
            for (int e2 = 0; e2 < 4; e2++)
                split_h(__builtin_ldexpf(acc[t][8 * s2 + 2 * e2], kx), __builtin_ldexpf(acc[t][8 * s2 + 2 * e2 + 1], kx),
                        bh[2 * t + s2].u[e2], bl[2 * t + s2].u[e2]);
    if (stamp >= 0) {
        asm volatile("" ::"v"(bl[7].u[3]), "v"(bl[0].u[0]), "v"(bh[3].u[1]));
        H2_STAMP(stamp + 1);
    }
    return kx;
}

// NW = 8: one 256-point workgroup per CU, two LDS buffers (layer l+1 is fetched while layer l computes), one barrier per
// layer.  NW = 4: TWO independent 128-point workgroups per CU with one buffer each (2 x 65 KB); the next layer's fetch has
// to wait for the layer's last LDS read and then hides under the epilogue -- but the SIMD's two waves now belong to different
// workgroups, share no barrier and drift apart: one's MFMA stretch runs under the other's epilogue.
template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void warp_fwd_h2_kernel(
    const float *__restrict__ x, const int32_t *__restrict__ slot, const float *__restrict__ bias0_d,
    const float *__restrict__ bias0_t, const f32x4 *__restrict__ w2_d, const f32x4 *__restrict__ w2_t,
    const float *__restrict__ bias_d, const float *__restrict__ bias_t, int n_bands, float *__restrict__ out_deform,
    float *__restrict__ out_topo, float *__restrict__ acts, uint32_t *__restrict__ amax, int64_t M, int64_t n_tiles) {
    constexpr bool TWO = NW == 8;                         // two LDS buffers
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pt = lane & 31, h = lane >> 5;
    const int64_t tile_id = (int64_t)blockIdx.x * NW + wave;
    const int64_t p = tile_id * TILE + pt;
    const int64_t pc = p < M ? p : M - 1;
    float xv[3] = {x[pc * 3 + 0], x[pc * 3 + 1], x[pc * 3 + 2]};
    const int sl = slot ? slot[pc] : 0;
    // the scratch holds whole 128-point blocks (mh_mlp_tiles); a 256-point workgroup's tail tiles beyond it park nothing
    float *tile = (acts && tile_id < n_tiles) ? acts + tile_id * (int64_t)(WARP_ACT_ROWS * TILE) : nullptr;

    int cb = 0;                                           // LDS buffer of the layer about to run
    h2_stage<H2_L0_F4, NW>(cb, w2_d);
    float bin0[24];
    enc_bin(xv, h, n_bands, bin0);
#pragma unroll
    for (int k = 20; k < 24; k++) bin0[k] = 0.f;
    if (tile) {
#pragma unroll
        for (int k = 0; k < 32; k++) tile[(2 * k + h) * TILE + pt] = k < 20 ? bin0[k] : 0.f;  // k-step ordered, rows 40..63 pad
    }
    uint2 *mk = tile ? reinterpret_cast<uint2 *>(tile + WARP_HID_ROWS * TILE) : nullptr;
    int m0 = 0;
#pragma unroll
    for (int k = 0; k < 20; k++) m0 = max(m0, __float_as_int(bin0[k]) & 0x7fffffff);
    if (amax) h2_note_amax(amax, 0, m0);
    const int kx0 = h2_point_shift(m0, H2_FWD_CLAMP);

    for (int net = 0; net < 2; net++) {
        const f32x4 *wp = net ? w2_t : w2_d;
        const uint32_t *tab = reinterpret_cast<const uint32_t *>(wp + H2_TAB_F4);
        const float *bs = net ? bias_t : bias_d;
        const float *b0 = (net ? bias0_t : bias0_d) + (int64_t)sl * 128;
        float *ht = tile ? tile + (64 + net * 640) * TILE : nullptr;
        f32x16 acc[4];
        FragH bh[8], bl[8];
        // the fetch of the layer after stage `l` (0..5), into buffer `buf`
        auto fetch_after = [&](int l, int buf) {
            if (l < 4) {
                h2_stage<H2_LH_F4, NW>(buf, wp + H2_L0_F4 + l * H2_LH_F4);
                h2_stage_bias(buf, (l + 1) & 1, bs + l * 128, 32);
            } else if (l == 4) {
                h2_stage<H2_L5_F4, NW>(buf, wp + H2_L0_F4 + 4 * H2_LH_F4);
                h2_stage_bias(buf, 1, bs + 4 * 128, 8);     // b5 is one 32-row tile
            } else if (net == 0) {
                h2_stage<H2_L0_F4, NW>(buf, w2_t);
            }
        };
        // layer 0: 40 (+8 zero) -> 128, bias row chosen by the point's frame slot
#pragma unroll
        for (int s = 0; s < 3; s++)
#pragma unroll
            for (int e2 = 0; e2 < 4; e2++)
                split_h(__builtin_ldexpf(bin0[8 * s + 2 * e2], kx0), __builtin_ldexpf(bin0[8 * s + 2 * e2 + 1], kx0), bh[s].u[e2],
                        bl[s].u[e2]);
        int ks = h2_wexp(tab[0]) + kx0;
        acc_bias<4>(acc, b0, h);
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[t][r] = __builtin_ldexpf(acc[t][r], ks);
        h2_wait();
        if (TWO) fetch_after(0, cb ^ 1);
        h2_layer<3, 4>(lds_h2 + cb * H2_BUF_F4, bh, bl, acc, lane);
        if (TWO) {
            cb ^= 1;
        } else {
            __syncthreads();
            fetch_after(0, cb);
        }
        int kx = h2_epilogue(acc, ks, ht, mk ? mk + (net * 5 + 0) * 64 + lane : nullptr, pt, h, bh, bl, amax, H2_AMAX_H(net, 0));
        // layers 1..4: 128 -> 128
        for (int l = 1; l <= 4; l++) {
            if (net == 0) H2_STAMP((l - 1) * 8 + 0);
            if (net == 0 && l == 1) H2_STAMP_REAL(62);
            ks = h2_wexp(tab[l]) + kx;
            h2_wait();
            if (net == 0) H2_STAMP((l - 1) * 8 + 1);
            if (TWO) fetch_after(l, cb ^ 1);
            if (net == 0) H2_STAMP((l - 1) * 8 + 2);
            h2_hidden(lds_h2 + cb * H2_BUF_F4, bh, bl, acc, lane);
            if (net == 0) H2_STAMP((l - 1) * 8 + 3);
            const f32x4 *brow = lds_h2 + cb * H2_BUF_F4 + H2_LH_F4 + (l & 1) * 32;
            if (TWO) {
                cb ^= 1;
            } else {
                __syncthreads();
                fetch_after(l, cb);
            }
            if (net == 0) H2_STAMP((l - 1) * 8 + 4);
            kx = h2_epilogue_hidden(acc, ks, brow, ht ? ht + l * 128 * TILE : nullptr, mk ? mk + (net * 5 + l) * 64 + lane : nullptr, pt, h,
                                    bh, bl, amax, H2_AMAX_H(net, l), net == 0 ? (l - 1) * 8 + 5 : -1);
            if (net == 0 && l == 4) H2_STAMP_REAL(63);
        }
        // layer 5: 128 -> 3 | 2 (one padded tile)
        ks = h2_wexp(tab[5]) + kx;
        f32x16 o[1];
        h2_wait();
        if (TWO) fetch_after(5, cb ^ 1);
        h2_layer<8, 1, true>(lds_h2 + cb * H2_BUF_F4, bh, bl, o, lane);
        const f32x4 b5 = lds_h2[cb * H2_BUF_F4 + H2_LH_F4 + 32 + h];   // rows 0..3 of the padded tile (slot 1)
        if (TWO) {
            cb ^= 1;
        } else {
            __syncthreads();
            fetch_after(5, cb);
        }
        const float sc = __builtin_ldexpf(1.0f, -ks);
        if (h == 0 && p < M) {
            if (net == 0) {
                out_deform[p * 3 + 0] = __builtin_fmaf(o[0][0], sc, b5[0]);
                out_deform[p * 3 + 1] = __builtin_fmaf(o[0][1], sc, b5[1]);
                out_deform[p * 3 + 2] = __builtin_fmaf(o[0][2], sc, b5[2]);
            } else {
                out_topo[p * 2 + 0] = __builtin_fmaf(o[0][0], sc, b5[0]);
                out_topo[p * 2 + 1] = __builtin_fmaf(o[0][1], sc, b5[1]);
            }
        }
    }
}

// ---- backward-data -------------------------------------------------------------------------------------------------
// Same chain, transposed packs (T5, T4..T1, T0), ReLU derivative from the sign masks the forward parked; parks dPre tiles in
// mlp.hip's layout for mh_mlp_wgrad.  acc = 2^ks W^T dPre: masked, y = acc . 2^-ks parked, acc . 2^d sliced.
__device__ __forceinline__ int h2_epilogue_bwd(f32x16 (&acc)[4], int ks, uint2 m, float *__restrict__ dt, int pt, int h, FragH (&bh)[8],
                                               FragH (&bl)[8], uint32_t *__restrict__ amax, int slot) {
    int mi = 0;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const uint32_t mw = (t < 2 ? m.x : m.y) >> (16 * (t & 1));
#pragma unroll
        for (int r = 0; r < 16; r++) {
            acc[t][r] = mask_bit(mw, r, acc[t][r]);
            mi = max(mi, __float_as_int(acc[t][r]) & 0x7fffffff);
        }
    }
    if (dt) {
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) PARK_STORE(__builtin_ldexpf(acc[t][r], -ks), &dt[(32 * t + acc_row(r, h)) * TILE + pt]);
    }
    if (amax) h2_note_amax(amax, slot, __float_as_int(__builtin_ldexpf(__int_as_float(mi), -ks)));
    const int d = h2_point_shift(mi, H2_BWD_CLAMP - ks);
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
            for (int e2 = 0; e2 < 4; e2++)
                split_h(__builtin_ldexpf(acc[t][8 * s2 + 2 * e2], d), __builtin_ldexpf(acc[t][8 * s2 + 2 * e2 + 1], d),
                        bh[2 * t + s2].u[e2], bl[2 * t + s2].u[e2]);
    return d + ks;
}

template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void warp_bwd_h2_kernel(const float *__restrict__ x, const float *__restrict__ g_deform,
                                                                 const float *__restrict__ g_topo, const f32x4 *__restrict__ w2T_d,
                                                                 const f32x4 *__restrict__ w2T_t, int n_bands,
                                                                 const float *__restrict__ acts, float *__restrict__ dpre,
                                                                 float *__restrict__ g_x, uint32_t *__restrict__ amax, int64_t M,
                                                                 int64_t n_tiles) {
    constexpr bool TWO = NW == 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pt = lane & 31, h = lane >> 5;
    const int64_t tile_id = (int64_t)blockIdx.x * NW + wave;
    const int64_t p = tile_id * TILE + pt;
    const bool live = p < M;
    // the scratch holds whole 128-point blocks: a wave beyond it (tail of the last 256-point workgroup) runs the chain on
    // the last real tile's masks and stores nothing
    const bool have = tile_id < n_tiles;
    const float *atile = acts + (have ? tile_id : n_tiles - 1) * (int64_t)(WARP_ACT_ROWS * TILE);
    float *dtile = have ? dpre + tile_id * (int64_t)(WARP_DPRE_ROWS * TILE) : nullptr;
    float gx[3] = {0.f, 0.f, 0.f};

    int cb = 0;
    h2_stage<H2_T5_F4, NW>(cb, w2T_d);
    for (int net = 0; net < 2; net++) {
        const f32x4 *wt = net ? w2T_t : w2T_d;
        const uint32_t *tab = reinterpret_cast<const uint32_t *>(wt + H2_TABT_F4);
        const float *g = net ? g_topo : g_deform;
        const int nout = net ? 2 : 3;
        float *dt = dtile ? dtile + net * 672 * TILE : nullptr;
        const uint2 *mk = reinterpret_cast<const uint2 *>(atile + WARP_HID_ROWS * TILE) + net * 5 * 64 + lane;
        uint2 msk[5];
#pragma unroll
        for (int l = 0; l < 5; l++) msk[l] = mk[l * 64];
        // the fetch of the stage after chain position j (0 = T5, 1..4 = T4..T1, 5 = T0), into buffer `buf`
        auto fetch_after = [&](int j, int buf) {
            if (j < 4)
                h2_stage<H2_LH_F4, NW>(buf, wt + H2_T5_F4 + j * H2_LH_F4);
            else if (j == 4 && g_x)
                h2_stage<H2_T0_F4, NW>(buf, wt + H2_T5_F4 + 4 * H2_LH_F4);
            else if (net == 0)
                h2_stage<H2_T5_F4, NW>(buf, w2T_t);
        };
        // dPre5: rows 0..nout-1 carry the incoming gradient (no activation on the last layer)
        float d5[16];
#pragma unroll
        for (int r = 0; r < 16; r++) d5[r] = 0.f;
        if (g && live && h == 0) {
            d5[0] = g[p * nout + 0];
            d5[1] = g[p * nout + 1];
            if (nout == 3) d5[2] = g[p * nout + 2];
        }
        if (dt) store_acc_rows<1>(dt + 640 * TILE, d5, pt, h);
        int m5 = max(max(__float_as_int(d5[0]) & 0x7fffffff, __float_as_int(d5[1]) & 0x7fffffff), __float_as_int(d5[2]) & 0x7fffffff);
        int kx = h2_point_shift(m5, H2_BWD_CLAMP);
        FragH bh[8], bl[8];
#pragma unroll
        for (int s = 0; s < 2; s++)
#pragma unroll
            for (int e2 = 0; e2 < 4; e2++)
                split_h(__builtin_ldexpf(d5[8 * s + 2 * e2], kx), __builtin_ldexpf(d5[8 * s + 2 * e2 + 1], kx), bh[s].u[e2], bl[s].u[e2]);
        // dH5 = W5^T dPre5
        int ks = h2_wexp(tab[0]) + kx;
        f32x16 acc[4];
        h2_wait();
        if (TWO) fetch_after(0, cb ^ 1);
        h2_layer<2, 4, true>(lds_h2 + cb * H2_BUF_F4, bh, bl, acc, lane);
        if (TWO) {
            cb ^= 1;
        } else {
            __syncthreads();
            fetch_after(0, cb);
        }
        kx = h2_epilogue_bwd(acc, ks, msk[4], dt ? dt + 4 * 128 * TILE : nullptr, pt, h, bh, bl, amax, H2_AMAX_D(net, 4));
        for (int l = 4; l >= 1; l--) {
            // dH_l = W_l^T dPre_l, then dPre_{l-1} = dH_l masked by H_l's ReLU bits
            ks = h2_wexp(tab[5 - l]) + kx;
            h2_wait();
            if (TWO) fetch_after(5 - l, cb ^ 1);
            h2_hidden(lds_h2 + cb * H2_BUF_F4, bh, bl, acc, lane);
            if (TWO) {
                cb ^= 1;
            } else {
                __syncthreads();
                fetch_after(5 - l, cb);
            }
            kx = h2_epilogue_bwd(acc, ks, msk[l - 1], dt ? dt + (l - 1) * 128 * TILE : nullptr, pt, h, bh, bl, amax, H2_AMAX_D(net, l - 1));
        }
        if (g_x) {
            // d(enc features) = W0^T dPre0; rows ordered (kk = 16t + r, h = lane>>5).  Skipped when nobody asks for d/dx
            ks = h2_wexp(tab[5]) + kx;
            f32x16 e[2];
            h2_wait();
            if (TWO) fetch_after(5, cb ^ 1);
            h2_layer<8, 2, true>(lds_h2 + cb * H2_BUF_F4, bh, bl, e, lane);
            if (TWO) {
                cb ^= 1;
            } else {
                __syncthreads();
                fetch_after(5, cb);
            }
            float dsc[18];
            enc_deriv_parked(atile, pt, h, dsc);
#pragma unroll
            for (int k = 0; k < 18; k++) {
                const float de = __builtin_ldexpf(k < 16 ? e[0][k] : e[1][k - 16], -ks);
                gx[k % 3] += de * dsc[k];
            }
            // kk 18: (x0 | x1), kk 19: (x2 | -)
            if (h == 0) {
                gx[0] += __builtin_ldexpf(e[1][2], -ks);
                gx[2] += __builtin_ldexpf(e[1][3], -ks);
            } else {
                gx[1] += __builtin_ldexpf(e[1][2], -ks);
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

// ---- canonical field forward (sdf_net + Laplace density, color_net; models/model.py:273-307) --------------------------------
// mlp_b3.hip's field_fwd_b3_kernel with two fp16 slices: a persistent 8-wave workgroup per CU keeps every layer's [h | l] planes
// in LDS (96 KB + the scale table), a wave owns a 32-point tile through both nets with no barrier, and parks the tile in
// exactly the fp32 kernel's layout (mh_field_bwd_fused consumes it).
//   sliced pack (packing.py, field_joint_packer().h2_blocks), float4 offsets: S0 0 (K = 80: 5 k16 steps, 2 tiles), S1 1536,
//   S2 2560, C0 3584, C1 4608 (K = 64: 4 steps, 2 tiles), C2 5632 (1 tile), the six layers' scale table 6144; 6656 in all
#define FH2_S0 0
#define FH2_S1 1536
#define FH2_S2 2560
#define FH2_C0 3584
#define FH2_C1 4608
#define FH2_C2 5632
#define FH2_TAB 6144
#define FH2_F4 6656
#define FH2_THREADS 512

__device__ __forceinline__ float laplace_sigma_h2(float s, float beta) {
    // density.py:22-31, cancellation-free form (mlp_dev.h: laplace_unit)
    return (1.0f / beta) * laplace_unit(s, beta);
}

// 64 pre-activations of a lane (two accumulator tiles, acc = 2^ks W x): y = relu(acc . 2^-ks + b), park feature-major, sign mask,
// the point's next exponent, slices of k16 steps 0..3.  bias: the layer's row in accumulator order (global, L1-resident)
__device__ __forceinline__ int fh2_epilogue(f32x16 (&acc)[2], int ks, const float *__restrict__ bias, float *__restrict__ ht,
                                            uint32_t *__restrict__ mk, int pt, int h, FragH (&bh)[8], FragH (&bl)[8]) {
    const float sc = __builtin_ldexpf(1.0f, -ks);
    int mi = 0;
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++) {
            const f32x4 b = *reinterpret_cast<const f32x4 *>(bias + 32 * t + 8 * r4 + 4 * h);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const float y = relu_i(__builtin_fmaf(acc[t][4 * r4 + c], sc, b[c]));
                acc[t][4 * r4 + c] = y;
                mi = max(mi, __float_as_int(y));
            }
        }
    uint32_t m = 0;
#pragma unroll
    for (int t = 1; t >= 0; t--)
#pragma unroll
        for (int r = 15; r >= 0; r--) m = push_nz(m, acc[t][r]);
    if (ht) {
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) PARK_STORE(acc[t][r], &ht[(32 * t + acc_row(r, h)) * TILE + pt]);
    }
    if (mk) *mk = m;
    const int kx = h2_point_shift(mi, H2_FWD_CLAMP);
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
            for (int e2 = 0; e2 < 4; e2++)
                split_h(__builtin_ldexpf(acc[t][8 * s2 + 2 * e2], kx), __builtin_ldexpf(acc[t][8 * s2 + 2 * e2 + 1], kx),
                        bh[2 * t + s2].u[e2], bl[2 * t + s2].u[e2]);
    return kx;
}

__global__ __launch_bounds__(FH2_THREADS, 2) void field_fwd_h2_kernel(
    const float *__restrict__ xc, const float *__restrict__ feat_s, const float *__restrict__ feat_c, const float *__restrict__ topo,
    const f32x4 *__restrict__ w2, const float *__restrict__ bias, const float *__restrict__ beta_p, int n_bands, int with_color,
    float *__restrict__ sdf, float *__restrict__ sigma, float *__restrict__ albedo, float *__restrict__ acts, int64_t M,
    int64_t n_tiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pt = lane & 31, h = lane >> 5;
    {
        const int n = with_color ? FH2_F4 : FH2_C0;      // the sdf-only pass (finite-difference taps) needs S0..S2 only
        for (int i = threadIdx.x; i < n; i += FH2_THREADS) lds_h2[i] = w2[i];
    }
    // weight exponents of the six layers (uniform: scalar loads)
    const uint32_t *tab = reinterpret_cast<const uint32_t *>(w2 + FH2_TAB);
    const int kws0 = h2_wexp(tab[0]), kws1 = h2_wexp(tab[1]), kws2 = h2_wexp(tab[2]);
    const int kwc0 = h2_wexp(tab[3]), kwc1 = h2_wexp(tab[4]), kwc2 = h2_wexp(tab[5]);
    __syncthreads();
    for (int64_t tile_id = (int64_t)blockIdx.x * (FH2_THREADS / 64) + wave; tile_id < n_tiles;
         tile_id += (int64_t)gridDim.x * (FH2_THREADS / 64)) {
        const int64_t p = tile_id * TILE + pt;
        const bool live = p < M;
        const int64_t pc = live ? p : M - 1;
        float xv[3] = {xc[pc * 3 + 0], xc[pc * 3 + 1], xc[pc * 3 + 2]};
        float *tile = acts ? acts + tile_id * (int64_t)(FIELD_ACT_ROWS * TILE) : nullptr;
        uint32_t *mk = tile ? reinterpret_cast<uint32_t *>(tile + FIELD_HID_ROWS * TILE) + lane : nullptr;
        FragH bh[8], bl[8];
        f32x16 acc[2];
        int kx;
        {
            // sdf L0 input, k-step ordered: 20 encoding steps | 16 hash-feature steps (level pairs) | topo | zero pad
            float bin0[40];
            enc_bin(xv, h, n_bands, bin0);
            const f32x4 *fs = reinterpret_cast<const f32x4 *>(feat_s + pc * 32 + 16 * h);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const f32x4 v = fs[q];
#pragma unroll
                for (int c = 0; c < 4; c++) bin0[20 + 4 * q + c] = v[c];
            }
            bin0[36] = topo ? topo[pc * 2 + h] : 0.f;
            bin0[37] = bin0[38] = bin0[39] = 0.f;
            if (tile) {
#pragma unroll
                for (int k = 0; k < 48; k++) PARK_STORE(k < 40 ? bin0[k] : 0.f, &tile[(2 * k + h) * TILE + pt]);   // rows 80..95 pad
            }
            int m0 = 0;
#pragma unroll
            for (int k = 0; k < 37; k++) m0 = max(m0, __float_as_int(bin0[k]) & 0x7fffffff);
            kx = h2_point_shift(m0, H2_FWD_CLAMP);
#pragma unroll
            for (int s = 0; s < 5; s++)
#pragma unroll
                for (int e2 = 0; e2 < 4; e2++)
                    split_h(__builtin_ldexpf(bin0[8 * s + 2 * e2], kx), __builtin_ldexpf(bin0[8 * s + 2 * e2 + 1], kx), bh[s].u[e2],
                            bl[s].u[e2]);
        }
        // sdf L0: 73 -> 64
        int ks = kws0 + kx;
        h2_layer<5, 2, true>(lds_h2 + FH2_S0, bh, bl, acc, lane);
        kx = fh2_epilogue(acc, ks, bias, tile ? tile + 96 * TILE : nullptr, mk ? mk + 0 * 64 : nullptr, pt, h, bh, bl);
        // sdf L1: 64 -> 64
        ks = kws1 + kx;
        h2_layer<4, 2, true>(lds_h2 + FH2_S1, bh, bl, acc, lane);
        kx = fh2_epilogue(acc, ks, bias + 64, tile ? tile + 160 * TILE : nullptr, mk ? mk + 1 * 64 : nullptr, pt, h, bh, bl);
        // sdf L2: 64 -> [geo(32) | sdf], no activation
        ks = kws2 + kx;
        const float sc2 = __builtin_ldexpf(1.0f, -ks);
        if (!with_color) {
            // sdf-only pass: the geo tile feeds nothing -- evaluate the tile holding the sdf row only (plane stride of the
            // two-tile pack: 2 * 4 * 64)
            f32x16 a1 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const f32x4 *w = lds_h2 + FH2_S2 + 4 * 64;      // tile 1 of each plane
#pragma unroll
            for (int s = 0; s < 4; s++) {
                FragH ah, al;
                ah.f = w[0 * 512 + s * 64 + lane];
                al.f = w[1 * 512 + s * 64 + lane];
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al.h, bh[s].h, a1, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.h, bl[s].h, a1, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.h, bh[s].h, a1, 0, 0, 0);
            }
            if (h == 0 && live) {
                const float sv = __builtin_fmaf(a1[0], sc2, bias[128 + 32]);
                sdf[p] = sv;
                if (sigma) sigma[p] = laplace_sigma_h2(sv, *beta_p);
            }
            continue;
        }
        h2_layer<4, 2, true>(lds_h2 + FH2_S2, bh, bl, acc, lane);
        if (h == 0 && live) {
            const float sv = __builtin_fmaf(acc[1][0], sc2, bias[128 + 32]);
            sdf[p] = sv;
            if (sigma) sigma[p] = laplace_sigma_h2(sv, *beta_p);
        }
        // color L0: [hash_c(32) | geo(32)] -> 64
        {
            float binc[32];
            const f32x4 *fc = reinterpret_cast<const f32x4 *>(feat_c + pc * 32 + 16 * h);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const f32x4 v = fc[q];
#pragma unroll
                for (int c = 0; c < 4; c++) binc[4 * q + c] = v[c];
            }
#pragma unroll
            for (int r4 = 0; r4 < 4; r4++) {
                const f32x4 b = *reinterpret_cast<const f32x4 *>(bias + 128 + 8 * r4 + 4 * h);
#pragma unroll
                for (int c = 0; c < 4; c++) binc[16 + 4 * r4 + c] = __builtin_fmaf(acc[0][4 * r4 + c], sc2, b[c]);
            }
            if (tile) {
#pragma unroll
                for (int k = 0; k < 32; k++) PARK_STORE(binc[k], &tile[(224 + 2 * k + h) * TILE + pt]);
            }
            int mc = 0;
#pragma unroll
            for (int k = 0; k < 32; k++) mc = max(mc, __float_as_int(binc[k]) & 0x7fffffff);
            kx = h2_point_shift(mc, H2_FWD_CLAMP);
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int e2 = 0; e2 < 4; e2++)
                    split_h(__builtin_ldexpf(binc[8 * s + 2 * e2], kx), __builtin_ldexpf(binc[8 * s + 2 * e2 + 1], kx), bh[s].u[e2],
                            bl[s].u[e2]);
        }
        ks = kwc0 + kx;
        h2_layer<4, 2, true>(lds_h2 + FH2_C0, bh, bl, acc, lane);
        kx = fh2_epilogue(acc, ks, bias + 192, tile ? tile + 288 * TILE : nullptr, mk ? mk + 2 * 64 : nullptr, pt, h, bh, bl);
        // color L1
        ks = kwc1 + kx;
        h2_layer<4, 2, true>(lds_h2 + FH2_C1, bh, bl, acc, lane);
        kx = fh2_epilogue(acc, ks, bias + 256, tile ? tile + 352 * TILE : nullptr, mk ? mk + 3 * 64 : nullptr, pt, h, bh, bl);
        // color L2: 64 -> 3, sigmoid
        ks = kwc2 + kx;
        f32x16 o[1];
        h2_layer<4, 1, true>(lds_h2 + FH2_C2, bh, bl, o, lane);
        if (h == 0 && live) {
            const float sc = __builtin_ldexpf(1.0f, -ks);
#pragma unroll
            for (int c = 0; c < 3; c++) albedo[p * 3 + c] = 1.0f / (1.0f + expf(-__builtin_fmaf(o[0][c], sc, bias[320 + c])));
        }
    }
}

// ---- weight slices --------------------------------------------------------------------------------------------------
// One pass finds each layer's largest |w| (fp32 bits into the net's table), the next cuts the layers -- gathered in fragment
// order by the caller (packing.py: fwd3 / bwd3 maps, shared with mlp_b3.hip) -- into [h | l] fp16 planes at the layer's scale.
#define H2_MAX_BLOCKS 32
struct H2Groups {
    int src_off[H2_MAX_BLOCKS];   // floats
    int n[H2_MAX_BLOCKS];         // floats of the whole layer (all its blocks are consecutive in src)
    int tab[H2_MAX_BLOCKS];       // index (32-bit words) of the layer's table entry in dst
};
struct H2Blocks {
    int n_blocks;
    int src_off[H2_MAX_BLOCKS];   // floats
    int n8[H2_MAX_BLOCKS];        // groups of 8 floats (= float4 of fp16 per plane)
    int dst_off[H2_MAX_BLOCKS];   // float4 units
    int g_end[H2_MAX_BLOCKS];     // running end of the blocks' group ranges
    int tab[H2_MAX_BLOCKS];       // the block's layer's table entry
};

__global__ __launch_bounds__(256) void h2_amax_kernel(const float *__restrict__ src, uint32_t *__restrict__ dst, H2Groups G) {
    __shared__ int part[4];
    const int g = blockIdx.x;
    const float *s = src + G.src_off[g];
    int m = 0;
    for (int i = threadIdx.x; i < G.n[g]; i += 256) m = max(m, __float_as_int(s[i]) & 0x7fffffff);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = max(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) dst[G.tab[g]] = (uint32_t)max(max(part[0], part[1]), max(part[2], part[3]));
}

__global__ void h2_slice_kernel(const float *__restrict__ src, f32x4 *__restrict__ dst, H2Blocks L) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= L.g_end[L.n_blocks - 1]) return;
    int l = 0;
    while (g >= L.g_end[l]) l++;
    const int i = g - (l ? L.g_end[l - 1] : 0);
    const int k = h2_wexp(reinterpret_cast<const uint32_t *>(dst)[L.tab[l]]);
    const f32x4 a = *reinterpret_cast<const f32x4 *>(src + L.src_off[l] + 8 * i);
    const f32x4 b = *reinterpret_cast<const f32x4 *>(src + L.src_off[l] + 8 * i + 4);
    FragH fh, fl;
    split_h(__builtin_ldexpf(a[0], k), __builtin_ldexpf(a[1], k), fh.u[0], fl.u[0]);
    split_h(__builtin_ldexpf(a[2], k), __builtin_ldexpf(a[3], k), fh.u[1], fl.u[1]);
    split_h(__builtin_ldexpf(b[0], k), __builtin_ldexpf(b[1], k), fh.u[2], fl.u[2]);
    split_h(__builtin_ldexpf(b[2], k), __builtin_ldexpf(b[3], k), fh.u[3], fl.u[3]);
    f32x4 *d = dst + L.dst_off[l] + i;
    d[0] = fh.f;
    d[L.n8[l]] = fl.f;
}

// src: the layers' weights gathered in fragment order, block after block; block b: n[b] floats at src_off[b] -> planes at
// float4 offset dst_off_f4[b] of dst; layer[b] = the block's layer (0..n_layers-1, blocks of a layer consecutive);
// table_word[layer] = 32-bit word index in dst that receives the layer's largest |w|.
extern "C" int mh_h2_slice(const float *src, void *dst, int32_t n_blocks, const int32_t *src_off_host, const int32_t *n_host,
                           const int32_t *dst_off_f4_host, const int32_t *layer_host, int32_t n_layers,
                           const int32_t *table_word_host, void *stream) {
    if (n_blocks == 0) return MH_OK;
    if (!src || !dst || n_blocks < 0 || n_blocks > H2_MAX_BLOCKS || n_layers <= 0 || n_layers > n_blocks || !src_off_host ||
        !n_host || !dst_off_f4_host || !layer_host || !table_word_host)
        return MH_ERR_ARG;
    H2Blocks L;
    H2Groups G;
    L.n_blocks = n_blocks;
    int end = 0, prev = -1;
    for (int b = 0; b < n_blocks; b++) {
        const int ly = layer_host[b];
        if (n_host[b] <= 0 || n_host[b] % 8 || src_off_host[b] % 4 || src_off_host[b] < 0 || dst_off_f4_host[b] < 0 || ly < 0 ||
            ly >= n_layers || (ly != prev && ly != prev + 1) || table_word_host[ly] < 0)
            return MH_ERR_ARG;
        if (ly != prev) {
            G.src_off[ly] = src_off_host[b];
            G.n[ly] = 0;
            G.tab[ly] = table_word_host[ly];
        } else if (src_off_host[b] != G.src_off[ly] + G.n[ly]) {
            return MH_ERR_ARG;   // a layer's blocks must be consecutive in src
        }
        G.n[ly] += n_host[b];
        prev = ly;
        L.src_off[b] = src_off_host[b];
        L.n8[b] = n_host[b] / 8;
        L.dst_off[b] = dst_off_f4_host[b];
        L.tab[b] = table_word_host[ly];
        end += n_host[b] / 8;
        L.g_end[b] = end;
    }
    if (prev != n_layers - 1) return MH_ERR_ARG;
    hipLaunchKernelGGL(h2_amax_kernel, dim3(n_layers), dim3(256), 0, mh_stream(stream), src, reinterpret_cast<uint32_t *>(dst), G);
    hipLaunchKernelGGL(h2_slice_kernel, dim3((end + 255) / 256), dim3(256), 0, mh_stream(stream), src, reinterpret_cast<f32x4 *>(dst), L);
    MH_CHECK_LAUNCH();
    return MH_OK;
}

extern "C" int64_t mh_warp_w2_bytes(void) { return (int64_t)H2_NET_F4 * 16; }
extern "C" int64_t mh_warp_w2T_bytes(void) { return (int64_t)H2_NETT_F4 * 16; }

// workgroup shape: the forward runs two independent 4-wave workgroups per CU with one LDS buffer each, backward-data one 8-wave
// workgroup per CU with two buffers.  Measured at 2 M points: forward 2.59 (8) / 2.53 ms (4), backward-data 2.39 (8) / 2.58 ms (4).
#define H2_FWD_WAVES 4
#define H2_BWD_WAVES 8
static int h2_lds_opt_in() {
    static MhOncePerDevice done;
    const int dev = mh_device();
    if (done.need(dev)) {
        if (hipFuncSetAttribute((const void *)warp_bwd_h2_kernel<H2_BWD_WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, H2_LDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute((const void *)warp_fwd_h2_kernel<H2_FWD_WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, H2_BUF_F4 * 16) != hipSuccess)
            return MH_ERR_LAUNCH;
        done.mark(dev);
    }
    return MH_OK;
}

extern "C" int mh_warp_fwd_h2(const float *x, const int32_t *slot, const float *bias0_d, const float *bias0_t, const void *w2_d,
                              const void *w2_t, const float *bias_d, const float *bias_t, int32_t n_bands, float *out_deform,
                              float *out_topo, float *acts, uint32_t *amax, int64_t M, void *stream) {
    if (M == 0) return MH_OK;
    if (M < 0 || !x || !bias0_d || !bias0_t || !w2_d || !w2_t || !bias_d || !bias_t || !out_deform || !out_topo || n_bands < 0 ||
        n_bands > 6)
        return MH_ERR_ARG;
    const int64_t blocks = (M + H2_FWD_WAVES * TILE - 1) / (H2_FWD_WAVES * TILE);
    if (blocks > 0x7fffffffLL) return MH_ERR_ARG;
    if (h2_lds_opt_in() != MH_OK) return MH_ERR_LAUNCH;
    const f32x4 *wd = reinterpret_cast<const f32x4 *>(w2_d), *wt = reinterpret_cast<const f32x4 *>(w2_t);
    hipLaunchKernelGGL(warp_fwd_h2_kernel<H2_FWD_WAVES>, dim3((unsigned)blocks), dim3(64 * H2_FWD_WAVES), H2_BUF_F4 * 16, mh_stream(stream),
                       x, slot, bias0_d, bias0_t, wd, wt, bias_d, bias_t, (int)n_bands, out_deform, out_topo, acts, amax, M, mh_mlp_tiles(M));
    MH_CHECK_LAUNCH();
    return MH_OK;
}

extern "C" int mh_warp_bwd_data_h2(const float *x, const float *g_deform, const float *g_topo, const void *w2T_d, const void *w2T_t,
                                   int32_t n_bands, const float *acts, float *dpre, float *g_x, uint32_t *amax, int64_t M,
                                   void *stream) {
    if (M == 0) return MH_OK;
    if (M < 0 || !x || !w2T_d || !w2T_t || !acts || !dpre || n_bands < 0 || n_bands > 6) return MH_ERR_ARG;
    const int64_t blocks = (M + H2_BWD_WAVES * TILE - 1) / (H2_BWD_WAVES * TILE);
    if (blocks > 0x7fffffffLL) return MH_ERR_ARG;
    if (h2_lds_opt_in() != MH_OK) return MH_ERR_LAUNCH;
    const f32x4 *wd = reinterpret_cast<const f32x4 *>(w2T_d), *wt = reinterpret_cast<const f32x4 *>(w2T_t);
    hipLaunchKernelGGL(warp_bwd_h2_kernel<H2_BWD_WAVES>, dim3((unsigned)blocks), dim3(64 * H2_BWD_WAVES), H2_LDS_BYTES, mh_stream(stream),
                       x, g_deform, g_topo, wd, wt, (int)n_bands, acts, dpre, g_x, amax, M, mh_mlp_tiles(M));
    MH_CHECK_LAUNCH();
    return MH_OK;
}

extern "C" int64_t mh_field_w2_bytes(void) { return (int64_t)FH2_F4 * 16; }

extern "C" int mh_field_fwd_h2(const float *xc, const float *feat_s, const float *feat_c, const float *topo, const void *w2,
                               const float *bias, const float *beta, int32_t n_bands, int32_t with_color, float *sdf, float *sigma,
                               float *albedo, float *acts, int64_t M, void *stream) {
    if (M == 0) return MH_OK;
    if (M < 0 || !xc || !feat_s || !w2 || !bias || !sdf || n_bands < 0 || n_bands > 6 || !beta) return MH_ERR_ARG;
    if (with_color && (!feat_c || !albedo)) return MH_ERR_ARG;
    const int64_t n_tiles = mh_mlp_tiles(M);   // dead tail tiles are processed too: the backward reads every scratch tile
    static MhOncePerDevice ok;
    const int dev = mh_device();
    if (ok.need(dev)) {
        if (hipFuncSetAttribute((const void *)field_fwd_h2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FH2_F4 * 16) != hipSuccess)
            return MH_ERR_LAUNCH;
        ok.mark(dev);
    }
    const int64_t need = (n_tiles + FH2_THREADS / 64 - 1) / (FH2_THREADS / 64);
    const int64_t cus = mh_cu_count();
    hipLaunchKernelGGL(field_fwd_h2_kernel, dim3((unsigned)(need < cus ? need : cus)), dim3(FH2_THREADS), FH2_F4 * 16,
                       mh_stream(stream), xc, feat_s, feat_c, topo, reinterpret_cast<const f32x4 *>(w2), bias, beta, (int)n_bands,
                       (int)with_color, sdf, sigma, albedo, acts, M, n_tiles);
    MH_CHECK_LAUNCH();
    return MH_OK;
}

extern "C" int64_t mh_h2_amax_words(void) { return (int64_t)H2_AMAX_COPIES * 32; }
