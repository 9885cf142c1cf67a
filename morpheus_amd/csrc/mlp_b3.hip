// The warp networks (deform_net + topo_net, models/model.py:412-437) with EXACT fp32 products issued on the bf16 matrix pipe.
//
// gfx950's fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at the vector-ALU rate, 1/16 of the bf16 matrix rate, and blocks the
// SIMD's VALU issue while it runs.  Here every fp32 operand is cut into three bf16 slices (round to nearest,
// mlp_dev.h: split2),
//      x = hi + mid + lo   exactly   (8 + 8 + 8 significand bits),
// and a product W.x is the six cross terms whose weight is >= 2^-16 of the leading one,
//      Wh.xh + Wh.xm + Wm.xh + Wh.xl + Wm.xm + Wl.xh,
// each an exact bf16 x bf16 product accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  What is dropped (Wm.xl, Wl.xm, Wl.xl)
// is below 2^-24 of the product, less than one fp32 rounding: every operand keeps all 24 significand bits (the fp32-faithful
// default mode "b3"), for 6/16 of the matrix cycles.  tests/test_gpu_ops.py::test_warp_sliced_arithmetic_against_float64 measures
// it against float64 on six operand distributions: forward values and d/dx at the native fp32 kernels' own error; weight
// gradients -- 10^3..10^6-term sums through the bf16 pipe's internal adder, which does not round to nearest -- 3-8x theirs
// (2e-6..8e-6 rel-L2), two orders below the 1e-4 contract.
//
// Range: the split needs |x| below the bf16 maximum (3.39e38; fp32 reaches 3.40e38) -- above it hi rounds to infinity and the
// residual is NaN -- and operands below ~1e-33 lose their low slices to the bf16 subnormal range; neither occurs in a network
// whose activations are O(1).  Infinities and NaNs propagate as NaN.
//
// Layout facts the kernels rely on (32x32x16 bf16): lane (i = lane & 31, g = lane >> 5) supplies A[m = i][k = 8g..8g+7] and
// B[k = 8g..8g+7][n = i] as 8 packed bf16 (4 VGPRs); D is the same 32x32 fp32 accumulator layout as the fp32 MFMA,
// D[row = (r&3) + 8(r>>2) + 4g][col = i].  So accumulator registers 8s'..8s'+7 of output tile t, sliced and packed pairwise,
// ARE the B operand of k16-step s = 2t + s' of the next layer: the register-resident chain of mlp.hip carries over, with the
// k-permutation baked into the weight slices on the host (packing.py: frag_index_b3).  The parked tiles (activations
// feature-major [F][32] + ReLU masks) are bit-for-bit in the layout of mlp.hip's kernels, which keep serving backward.
//
// Weight slices of a layer: [plane hi|mid|lo][out tile][k16 step][lane][8 bf16], 96 KB for a 128 x 128 layer, staged by
// LDS-DMA once per 256-point workgroup (8 waves: two per SIMD share one copy, 160 KB of LDS do not hold two).
#include "mlp_dev.h"
#include <stdlib.h>

#define B3_THREADS 512
#define B3_BLOCK_PTS 256
#define B3_L0_F4 2560                                  // 3 planes x 4 tiles x 3 k16 steps x 64 lanes = 2304, padded to whole DMA rounds
#define B3_LH_F4 6144                                  // 128 x 128
#define B3_KH_F4 3072                                  // one k-half of it: 3 planes x 4 tiles x 4 k16 steps x 64 lanes (48 KB)
#define B3_L5_F4 1536                                  // one padded output tile
#define B3_NET_F4 (B3_L0_F4 + 4 * B3_LH_F4 + B3_L5_F4)  // 28 672 float4 = 448 KB per net
#define B3_LDS_BYTES (B3_LH_F4 * 16 + 1024)            // one layer's slices + the layer's bias row (forward)

extern __shared__ f32x4 lds_b3[];
#ifndef B3_Q4_FILL
#define B3_Q4_FILL 6
#endif
// Parking stores a wave issues per epilogue EIGHTH (8 accumulator registers of one tile: one global_store_dword each) and per
// epilogue HALF (two tiles = four eighths).  The vmcnt(KEEP) waits below are only right if at least KEEP such stores sit between
// the LDS-DMA they wait for and the wait itself: KEEP is DERIVED from these counts, never chosen (round-5 advisor finding: a -D
// override could exceed the store count and the kernel would read stale weight slices).
#define B3_PARK_STORES_EIGHTH 8
#define B3_PARK_STORES_HALF (4 * B3_PARK_STORES_EIGHTH)
#define B3_KEEP_BWD B3_PARK_STORES_HALF
static_assert(B3_KEEP_BWD <= B3_PARK_STORES_HALF && B3_PARK_STORES_HALF <= 63, "vmcnt(KEEP) must not exceed the stores behind the DMA");

#ifdef MH_PHASE_TRACE
// phase trace for tools/phase_trace_b3.py (never compiled into the product library): wave 0 of every 32nd workgroup stamps
// s_memtime at the phase boundaries of warp_fwd_b3_kernel's hidden layers of net 0 (8 slots per layer), s_memrealtime in 62/63
__device__ long long mh_b3_trace[256 * 64];
// rows: workgroup blockIdx.x / 64 (every 64th), wave 0 -> even row, wave 4 -> odd row (waves w and w + 4 share SIMD w)
#define B3_TRACE_ON (((threadIdx.x & 255) == 0) && (blockIdx.x % 64 == 0) && (blockIdx.x / 64 < 128))
#define B3_TRACE_ROW ((blockIdx.x / 64) * 2 + (threadIdx.x >> 8))
#define B3_STAMP(slot)                                                                          \
    do {                                                                                        \
        if (B3_TRACE_ON) mh_b3_trace[B3_TRACE_ROW * 64 + (slot)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
#define B3_STAMP_REAL(slot)                                                                     \
    do {                                                                                        \
        if (B3_TRACE_ON) mh_b3_trace[B3_TRACE_ROW * 64 + (slot)] = (long long)__builtin_amdgcn_s_memrealtime(); \
    } while (0)
extern "C" int mh_b3_trace_read(long long *dst_host) {
    return hipMemcpyFromSymbol(dst_host, HIP_SYMBOL(mh_b3_trace), sizeof(long long) * 256 * 64) == hipSuccess ? 0 : 2;
}
#else
#define B3_STAMP(slot) do { } while (0)
#define B3_STAMP_REAL(slot) do { } while (0)
#endif

template <int N_F4, int NTHR = B3_THREADS>
__device__ __forceinline__ void b3_stage_issue(const f32x4 *__restrict__ src) {
    static_assert(N_F4 % NTHR == 0, "whole rounds of the block");
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < N_F4 / NTHR; k++)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + k * NTHR + threadIdx.x),
                                         (__attribute__((address_space(3))) void *)(lds_b3 + k * NTHR + wave * 64), 16, 0, 0);
}
__device__ __forceinline__ void b3_stage_wait() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}
// The same wait when the wave has issued AT LEAST `YOUNGER` vector-memory operations (parking stores) after the DMA it waits for:
// vmcnt counts in issue order, so "all but the YOUNGER most recent" covers the DMA and everything older without waiting for the
// acknowledgement of stores issued a few hundred cycles ago (gfx9 has no separate store counter: vmcnt(0) waits for them too).
// The caller fences the DMA issue with b3_dma_fence() so that no store can be scheduled ahead of it; anything the compiler adds
// behind the fence (a spill) only makes the wait cover more.
template <int YOUNGER>
__device__ __forceinline__ void b3_stage_wait_keep() {
    static_assert(YOUNGER >= 0 && YOUNGER <= 63, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER) : "memory");
    __syncthreads();
}
__device__ __forceinline__ void b3_dma_fence() { asm volatile("" ::: "memory"); }

// acc[t] += W[t] . b : six slice products per k16 step, two output tiles in rotation (consecutive MFMAs never chain on one
// accumulator), small terms first
template <int KS, int MT, bool ZERO = false>
__device__ __forceinline__ void b3_layer(const f32x4 *__restrict__ w, const Frag (&bh)[8], const Frag (&bm)[8], const Frag (&bl)[8],
                                         f32x16 (&acc)[MT], int lane) {
    constexpr int PL = MT * KS * 64;
    constexpr int NT = MT >= 2 ? 2 : 1;
    // ZERO: the accumulators are not pre-initialised, the first MFMA of each takes the inline constant 0 as C (backward
    // layers have no bias)
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; s++) {
#pragma unroll
        for (int mp = 0; mp < MT; mp += NT) {
            Frag ah[NT], am[NT], al[NT];
#pragma unroll
            for (int t = 0; t < NT; t++) {
                ah[t].f = w[0 * PL + ((mp + t) * KS + s) * 64 + lane];
                am[t].f = w[1 * PL + ((mp + t) * KS + s) * 64 + lane];
                al[t].f = w[2 * PL + ((mp + t) * KS + s) * 64 + lane];
            }
#pragma unroll
            for (int t = 0; t < NT; t++)
                acc[mp + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[t].h, bh[s].h, (ZERO && s == 0) ? zero : acc[mp + t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[mp + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[t].h, bm[s].h, acc[mp + t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[mp + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t].h, bl[s].h, acc[mp + t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[mp + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[t].h, bh[s].h, acc[mp + t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[mp + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t].h, bm[s].h, acc[mp + t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[mp + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t].h, bh[s].h, acc[mp + t], 0, 0, 0);
        }
    }
}

// mlp_dev.h's push_gt0 (and, until this was found, relu1) is inline asm, and hipcc's hazard recognizer does not see an inline-asm READ of a register
// an in-flight bf16 MFMA is still writing: the first v_max after the layer's last MFMA would return the stale accumulator
// (found the hard way: the compiler sinks the layer's last MFMAs past the barrier and the DMA issue, right up to the first
// asm statement; the fp32 MFMA of mlp.hip runs in the vector ALU's own order and is immune).  So the epilogue opens with the
// wait states an 8-pass MFMA result needs (11; 16 given), fenced so that no MFMA moves below and no asm above them.
__device__ __forceinline__ void mfma_results_settle() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15");
    __builtin_amdgcn_sched_barrier(0);
}

// one quarter of a 128 x 128 layer: output tiles T0, T0+1 over k16 steps S0 .. S0+3 (accumulators already hold bias or the
// earlier steps)
template <int T0, int S0>
__device__ __forceinline__ void b3_quarter(const f32x4 *__restrict__ w, const Frag (&bh)[8], const Frag (&bm)[8], const Frag (&bl)[8],
                                           f32x16 (&acc)[4], int lane) {
    // a 128 x 128 layer is two k-half blocks [khalf][plane][out tile][k16 step 0..3][lane] of B3_KH_F4 float4 (packing.py)
    constexpr int PLH = 4 * 4 * 64;
#pragma unroll
    for (int s = S0; s < S0 + 4; s++) {
        Frag ah[2], am[2], al[2];
        const f32x4 *wk = w + (S0 / 4) * B3_KH_F4 + ((s & 3)) * 64 + lane;
#pragma unroll
        for (int t = 0; t < 2; t++) {
            ah[t].f = wk[0 * PLH + (T0 + t) * 256];
            am[t].f = wk[1 * PLH + (T0 + t) * 256];
            al[t].f = wk[2 * PLH + (T0 + t) * 256];
        }
#pragma unroll
        for (int t = 0; t < 2; t++) acc[T0 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[t].h, bh[s].h, acc[T0 + t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; t++) acc[T0 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[t].h, bm[s].h, acc[T0 + t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; t++) acc[T0 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t].h, bl[s].h, acc[T0 + t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; t++) acc[T0 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[t].h, bh[s].h, acc[T0 + t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; t++) acc[T0 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t].h, bm[s].h, acc[T0 + t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; t++) acc[T0 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t].h, bh[s].h, acc[T0 + t], 0, 0, 0);
    }
}

// one k16 step of output tiles T0, T0+1 (12 MFMAs)
template <int T0>
__device__ __forceinline__ void b3_quarter_step(const f32x4 *__restrict__ w, const Frag (&bh)[8], const Frag (&bm)[8],
                                                const Frag (&bl)[8], f32x16 (&acc)[4], int lane, int s) {
    constexpr int PLH = 4 * 4 * 64;
    Frag ah[2], am[2], al[2];
    const f32x4 *wk = w + (s >> 2) * B3_KH_F4 + (s & 3) * 64 + lane;
#pragma unroll
    for (int t = 0; t < 2; t++) {
        ah[t].f = wk[0 * PLH + (T0 + t) * 256];
        am[t].f = wk[1 * PLH + (T0 + t) * 256];
        al[t].f = wk[2 * PLH + (T0 + t) * 256];
    }
#pragma unroll
    for (int t = 0; t < 2; t++) acc[T0 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[t].h, bh[s].h, acc[T0 + t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 2; t++) acc[T0 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[t].h, bm[s].h, acc[T0 + t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 2; t++) acc[T0 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t].h, bl[s].h, acc[T0 + t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 2; t++) acc[T0 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[t].h, bh[s].h, acc[T0 + t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 2; t++) acc[T0 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t].h, bm[s].h, acc[T0 + t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 2; t++) acc[T0 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t].h, bh[s].h, acc[T0 + t], 0, 0, 0);
}

// registers 8 s2 .. 8 s2 + 7 of output tile t: ReLU in place, park, their 8 mask bits, the slices of k16 step 2t + s2
template <bool PARK>
__device__ __forceinline__ void b3_epilogue_eighth(f32x16 (&acc)[4], float *__restrict__ ht, uint32_t (&mt)[4], int pt, int h,
                                                   Frag (&bh)[8], Frag (&bm)[8], Frag (&bl)[8], int t, int s2) {
#pragma unroll
    for (int r = 8 * s2; r < 8 * s2 + 8; r++) acc[t][r] = relu_i(acc[t][r]);
    if (PARK) {
#pragma unroll
        for (int r = 8 * s2; r < 8 * s2 + B3_PARK_STORES_EIGHTH; r++) PARK_STORE(acc[t][r], &ht[(32 * t + acc_row(r, h)) * TILE + pt]);
    }
    uint32_t m = 0;
#pragma unroll
    for (int r = 8 * s2 + 7; r >= 8 * s2; r--) m = push_nz(m, acc[t][r]);
    mt[t] |= m << (8 * s2);
#pragma unroll
    for (int e2 = 0; e2 < 4; e2++)
        split2(acc[t][8 * s2 + 2 * e2], acc[t][8 * s2 + 2 * e2 + 1], bh[2 * t + s2].u[e2], bm[2 * t + s2].u[e2], bl[2 * t + s2].u[e2]);
    // (the caller pins the slices at the end of its quarter: b3_pin_slices)
}

// the epilogue of output tiles T0, T0+1 only: ReLU, park, sign-mask halves, slices of k16 steps 2 T0 .. 2 T0 + 3
template <int T0, bool PARK>
__device__ __forceinline__ void b3_epilogue_half(f32x16 (&acc)[4], float *__restrict__ ht, uint32_t (&mt)[4], int pt, int h,
                                                 Frag (&bh)[8], Frag (&bm)[8], Frag (&bl)[8]) {
#pragma unroll
    for (int t = T0; t < T0 + 2; t++) {
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = relu_i(acc[t][r]);
        if (PARK) {
#pragma unroll
            for (int r = 0; r < 16; r++) PARK_STORE(acc[t][r], &ht[(32 * t + acc_row(r, h)) * TILE + pt]);
        }
        uint32_t m = 0;
#pragma unroll
        for (int r = 15; r >= 0; r--) m = push_nz(m, acc[t][r]);
        mt[t] = m;
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
            for (int e2 = 0; e2 < 4; e2++)
                split2(acc[t][8 * s2 + 2 * e2], acc[t][8 * s2 + 2 * e2 + 1], bh[2 * t + s2].u[e2], bm[2 * t + s2].u[e2],
                       bl[2 * t + s2].u[e2]);
    }
}

// ---- forward, weight staging pipelined over two k-half slots (round 5) ------------------------------------------------------
// Round 4's kernel staged ONE layer's slices (96 KB) at a time: the next layer's DMA could only be issued once every wave had left
// the layer, and the two-wave phase trace (tools/phase_trace_b3_pair.py, profiles/r05_phase_trace_warp_fwd_pair_before.txt) shows
// what that costs: between the mid-layer barrier and the next layer's first MFMA -- DMA issue, the exposed epilogue of tiles 2, 3,
// the DMA's flight -- NEITHER wave of a SIMD issues an MFMA for 5.7 k of a layer's 20.2 k ticks (inside the rest the matrix pipe
// is saturated: 38 ticks per MFMA).  Here the same 96 KB are two 48 KB SLOTS, a slot = one k-half of a 128 x 128 layer (the packs
// are [k-half][plane][tile][k16 step][lane] already), and a hidden layer is
//      Q1 (tiles 0,1 x k-half 0) || E23 of the PREVIOUS layer     Q2 (tiles 2,3 x k0)      --M--
//      Q3 (tiles 0,1 x k-half 1)                                  Q4 (tiles 2,3 x k1) || E01   --E--
//   * k-half 0 lives in slot 1, k-half 1 in slot 0 (L0 in slot 0, L5 in slot 1).  At M every wave has left slot 1: it takes k0 of
//     the NEXT layer (L5 after layer 4); at E every wave has left slot 0: it takes k1 of the next layer (the other net's L0 after
//     layer 4).  A block is read half a layer after it was issued; the wait at M / E is `vmcnt(KEEP)`: all but the wave's KEEP most
//     recent vector-memory operations -- the parking stores of the quarter just finished -- so it covers the DMA issued at the
//     previous point without waiting for stores a few hundred cycles old (gfx9 counts both in vmcnt).
//   * E01 (ReLU, park, mask bits, slices of k16 steps 0..3 -- dead since Q2) runs under Q4 as before; E23, whose slices feed k16
//     steps 4..7 (first read in Q3), runs under Q1 of the NEXT layer: no epilogue is exposed except layer 0's tiles 0, 1.
//   * inside Q1 / Q4 the two instruction streams are interleaved by scheduling groups (one MFMA, B3_Q_FILL epilogue instructions):
//     tools/micro/mfma_valu_gap.hip -- 5-6 single-issue instructions ride free in the shadow of the wave's own bf16 MFMA.
// Every accumulator sees the same sequence of slice products as before: outputs, parked tiles and mask words are bit-identical.
// PARK: activations / masks are parked (training).  A wave beyond the scratch's last tile (tail of the last workgroup: the scratch
// holds whole 128-point blocks) re-runs that LAST tile and stores the same bytes again, so that no parking store sits behind a
// branch -- a branch ends the scheduling region the epilogue instructions are meant to share with the MFMAs.
#define B3_SLOT_F4 B3_KH_F4                             // 3072 float4 = 48 KB; L0 (2560) and L5 (1536) fit one slot
#define B3_BIASROW_F4 (2 * B3_SLOT_F4)                  // two bias rows of 32 float4 behind the slots (row = layer & 1)
#ifndef B3_Q_FILL
#define B3_Q_FILL 6
#endif

// wave-uniform source offset / slot -> LDS-DMA of N_F4 float4.  The offset passes through an empty asm statement: the address
// arithmetic is loop-invariant for most blocks, and hipcc otherwise hoists it out of the layer loop and spills it
template <int N_F4, int NTHR>
__device__ __forceinline__ void b3_slot_issue(const f32x4 *__restrict__ base, int off_f4, int slot) {
    static_assert(N_F4 % NTHR == 0, "whole rounds of the block");
    asm volatile("" : "+s"(off_f4));
    const f32x4 *src = base + off_f4 + threadIdx.x;
    f32x4 *dst = lds_b3 + slot * B3_SLOT_F4 + (threadIdx.x >> 6) * 64;
#pragma unroll
    for (int k = 0; k < N_F4 / NTHR; k++)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + k * NTHR),
                                         (__attribute__((address_space(3))) void *)(dst + k * NTHR), 16, 0, 0);
}
__device__ __forceinline__ void b3_bias_issue(const float *__restrict__ bias, int off_floats, int n_f4, int row) {
    asm volatile("" : "+s"(off_floats));
    if ((int)threadIdx.x < n_f4)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const f32x4 *>(bias + off_floats) + threadIdx.x),
                                         (__attribute__((address_space(3))) void *)(lds_b3 + B3_BIASROW_F4 + row * 32), 16, 0, 0);
}
// accumulators of output tiles T0, T0+1 <- the layer's bias row
template <int T0, int NT, int MT>
__device__ __forceinline__ void b3_acc_bias_row(f32x16 (&acc)[MT], int h, int row) {
    const f32x4 *b = lds_b3 + B3_BIASROW_F4 + row * 32;
#pragma unroll
    for (int t = T0; t < T0 + NT; t++)
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++) {
            const f32x4 v = b[8 * t + 2 * r4 + h];
#pragma unroll
            for (int c = 0; c < 4; c++) acc[t][4 * r4 + c] = v[c];
        }
}
// M / E point: this wave's LDS reads have returned, its DMA portions older than its KEEP most recent VMEM operations have landed;
// then the workgroup barrier.  Not __syncthreads(): with an LDS-DMA in flight its release fence becomes vmcnt(0).
template <int KEEP>
__device__ __forceinline__ void b3_point() {
    static_assert(KEEP >= 0 && KEEP <= 63, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(KEEP) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// one k16 step `s` (B slices bh[s] ...) of output tiles T0, T0+1 against the k-half block `wk` (12 MFMAs)
template <int T0>
__device__ __forceinline__ void b3_step(const f32x4 *__restrict__ wk, const Frag (&bh)[8], const Frag (&bm)[8], const Frag (&bl)[8],
                                        f32x16 (&acc)[4], int lane, int s) {
    constexpr int PLH = 4 * 4 * 64;
    Frag ah[2], am[2], al[2];
    const f32x4 *w = wk + (s & 3) * 64 + lane;
#pragma unroll
    for (int t = 0; t < 2; t++) {
        ah[t].f = w[0 * PLH + (T0 + t) * 256];
        am[t].f = w[1 * PLH + (T0 + t) * 256];
        al[t].f = w[2 * PLH + (T0 + t) * 256];
    }
#pragma unroll
    for (int t = 0; t < 2; t++) acc[T0 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[t].h, bh[s].h, acc[T0 + t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 2; t++) acc[T0 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[t].h, bm[s].h, acc[T0 + t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 2; t++) acc[T0 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t].h, bl[s].h, acc[T0 + t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 2; t++) acc[T0 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[t].h, bh[s].h, acc[T0 + t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 2; t++) acc[T0 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t].h, bm[s].h, acc[T0 + t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 2; t++) acc[T0 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t].h, bh[s].h, acc[T0 + t], 0, 0, 0);
}
// one k16 step of the single-tile output layer (pack [plane][k16 step 0..7][lane]); the six products chain on one accumulator in
// b3_layer<8, 1>'s order
__device__ __forceinline__ void b3_l5_step(const f32x4 *__restrict__ w, const Frag (&bh)[8], const Frag (&bm)[8], const Frag (&bl)[8],
                                           f32x16 &o, int lane, int s) {
    Frag ah, am, al;
    ah.f = w[0 * 512 + s * 64 + lane];
    am.f = w[1 * 512 + s * 64 + lane];
    al.f = w[2 * 512 + s * 64 + lane];
    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al.h, bh[s].h, o, 0, 0, 0);
    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am.h, bm[s].h, o, 0, 0, 0);
    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.h, bl[s].h, o, 0, 0, 0);
    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am.h, bh[s].h, o, 0, 0, 0);
    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.h, bm[s].h, o, 0, 0, 0);
    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.h, bh[s].h, o, 0, 0, 0);
}
// The slices an epilogue makes are only USED after the next barrier, and hipcc's sinking pass would move the ~270 instructions that
// make them down to that use, out of the MFMA stretch they are meant to fill: pin them at the END of the quarter (one point, behind
// every MFMA and LDS read of the quarter -- a pin between the steps would fence the next step's LDS reads: asm volatile orders memory)
template <int S0>
__device__ __forceinline__ void b3_pin_slices(Frag (&bh)[8], Frag (&bm)[8], Frag (&bl)[8]) {
#pragma unroll
    for (int s = S0; s < S0 + 4; s++)
        asm volatile("" : "+v"(bh[s].u[0]), "+v"(bh[s].u[1]), "+v"(bh[s].u[2]), "+v"(bh[s].u[3]), "+v"(bm[s].u[0]), "+v"(bm[s].u[1]),
                     "+v"(bm[s].u[2]), "+v"(bm[s].u[3]), "+v"(bl[s].u[0]), "+v"(bl[s].u[1]), "+v"(bl[s].u[2]), "+v"(bl[s].u[3]));
}
// one MFMA, then FILL other VALU instructions, N times: the order the scheduler is asked for inside a quarter that carries an epilogue
template <int N, int FILL>
__device__ __forceinline__ void b3_interleave() {
#pragma unroll
    for (int g = 0; g < N; g++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, FILL, 0);
    }
}

template <int NW, bool PARK>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 1 : 2) void warp_fwd_b3_kernel(
    const float *__restrict__ x, const int32_t *__restrict__ slot, const float *__restrict__ bias0_d,
    const float *__restrict__ bias0_t, const f32x4 *__restrict__ w3_d, const f32x4 *__restrict__ w3_t,
    const float *__restrict__ bias_d, const float *__restrict__ bias_t, int n_bands, float *__restrict__ out_deform,
    float *__restrict__ out_topo, float *__restrict__ acts, int64_t M, int64_t n_tiles) {
    constexpr int NT = NW * 64;
    constexpr int KEEP = PARK ? B3_PARK_STORES_HALF : 0;      // parking stores a wave issues in a quarter that carries an epilogue (4 eighths)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pt = lane & 31, h = lane >> 5;
    const int64_t tile_raw = (int64_t)blockIdx.x * NW + wave;
    const int64_t tile_id = tile_raw < n_tiles ? tile_raw : n_tiles - 1;
    const int64_t p = tile_id * TILE + pt;
    const int64_t pc = p < M ? p : M - 1;
    float xv[3] = {x[pc * 3 + 0], x[pc * 3 + 1], x[pc * 3 + 2]};
    const int sl = slot ? slot[pc] : 0;
    float *tile = PARK ? acts + tile_id * (int64_t)(WARP_ACT_ROWS * TILE) : nullptr;
    const f32x4 *const slot0 = lds_b3, *const slot1 = lds_b3 + B3_SLOT_F4;

    // blocks L0d -> slot 0, L1d.k0 -> slot 1, L1d's bias row -> row 1
    b3_slot_issue<B3_L0_F4, NT>(w3_d, 0, 0);
    b3_slot_issue<B3_KH_F4, NT>(w3_d, B3_L0_F4, 1);
    b3_bias_issue(bias_d, 0, 32, 1);
    float bin0[24];
    enc_bin(xv, h, n_bands, bin0);
#pragma unroll
    for (int k = 20; k < 24; k++) bin0[k] = 0.f;
    if (PARK) {
#if defined(MH_PARK_PAD_ROWS) || defined(MH_PARK_H0_PAD)   // A/B (tools/gpu/r6_pad_ab.sh): the round-5 form, pad rows written (and, with MORPHEUS_WGRAD_LIVE=0, read)
#pragma unroll
        for (int k = 0; k < 32; k++) tile[(2 * k + h) * TILE + pt] = k < 20 ? bin0[k] : 0.f;
#else
#pragma unroll
        for (int k = 0; k < 20; k++) tile[(2 * k + h) * TILE + pt] = bin0[k];  // k-step ordered; the pad rows 40..63 are never read (wg_row)
#endif
    }
    uint2 *mk = PARK ? reinterpret_cast<uint2 *>(tile + WARP_HID_ROWS * TILE) : nullptr;

    for (int net = 0; net < 2; net++) {
        const f32x4 *wn = net ? w3_t : w3_d;
        const float *bs = net ? bias_t : bias_d;
        const float *b0 = (net ? bias0_t : bias0_d) + (int64_t)sl * 128;
        float *ht = PARK ? tile + (64 + net * 640) * TILE : nullptr;
        f32x16 acc[4];
        Frag bh[8], bm[8], bl[8];
        uint32_t mt[4] = {0u, 0u, 0u, 0u};
        // layer 0: 40 (+8 zero) -> 128, bias row chosen by the point's frame slot.  (The encoding's 24 values stay live across the first
        // net -- the compiler keeps them in scratch, written and read back once per net, outside the layer loop; re-evaluating them for
        // the second net instead was tried: 45 -> 60 spilled registers, hipcc hoists other things in their place.)
#pragma unroll
        for (int s = 0; s < 3; s++)
#pragma unroll
            for (int e2 = 0; e2 < 4; e2++) split2(bin0[8 * s + 2 * e2], bin0[8 * s + 2 * e2 + 1], bh[s].u[e2], bm[s].u[e2], bl[s].u[e2]);
        acc_bias<4>(acc, b0, h);
        b3_point<0>();                                     // L0, L1.k0 and L1's bias row have landed (everything has)
        b3_layer<3, 4>(slot0, bh, bm, bl, acc, lane);
        b3_point<0>();                                     // every wave has left slot 0
        b3_slot_issue<B3_KH_F4, NT>(wn, B3_L0_F4 + B3_KH_F4, 0);           // L1.k1 -> slot 0
        b3_dma_fence();
        // layer 0's tiles 0, 1 here (a 72-MFMA layer has nothing to hide them under); its tiles 2, 3 under layer 1's Q1
        b3_epilogue_half<0, PARK>(acc, ht, mt, pt, h, bh, bm, bl);
        __builtin_amdgcn_sched_barrier(0);
        // layers 1..4: 128 -> 128
        for (int l = 1; l <= 4; l++) {
            const int row = l & 1;
            float *hp = ht + (l - 1) * 128 * TILE, *hl = ht + l * 128 * TILE;
            if (net == 0) B3_STAMP((l - 1) * 8 + 0);
            if (net == 0 && l == 1) B3_STAMP_REAL(62);
            // ---- Q1 (slot 1: k-half 0) under the PREVIOUS layer's tiles 2, 3 epilogue
            b3_acc_bias_row<0, 2, 4>(acc, h, row);
            mt[2] = mt[3] = 0u;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                b3_step<0>(slot1, bh, bm, bl, acc, lane, c);
                b3_epilogue_eighth<PARK>(acc, hp, mt, pt, h, bh, bm, bl, 2 + (c >> 1), c & 1);
            }
            b3_pin_slices<4>(bh, bm, bl);
            asm volatile("" : "+v"(mt[2]), "+v"(mt[3]));
            b3_interleave<48, B3_Q_FILL>();
            __builtin_amdgcn_sched_barrier(0);
            if (PARK) mk[(net * 5 + l - 1) * 64 + lane] = make_uint2(mt[0] | (mt[1] << 16), mt[2] | (mt[3] << 16));
            if (net == 0) B3_STAMP((l - 1) * 8 + 1);
            // ---- Q2
            b3_acc_bias_row<2, 2, 4>(acc, h, row);
#pragma unroll
            for (int c = 0; c < 4; c++) b3_step<2>(slot1, bh, bm, bl, acc, lane, c);
            __builtin_amdgcn_sched_barrier(0);
            if (net == 0) B3_STAMP((l - 1) * 8 + 2);
            // ---- M: slot 1 is free; k-half 1 (issued at the previous E / after L0) has landed
            b3_point<KEEP>();
            if (l < 4)
                b3_slot_issue<B3_KH_F4, NT>(wn, B3_L0_F4 + 2 * l * B3_KH_F4, 1);          // k0 of layer l + 1
            else
                b3_slot_issue<B3_L5_F4, NT>(wn, B3_L0_F4 + 8 * B3_KH_F4, 1);              // L5
            b3_bias_issue(bs, l * 128, l < 4 ? 32 : 8, (l + 1) & 1);                       // its bias row (b5 is one 32-row tile)
            b3_dma_fence();
            if (net == 0) B3_STAMP((l - 1) * 8 + 3);
            // ---- Q3 (slot 0: k-half 1)
#pragma unroll
            for (int c = 4; c < 8; c++) b3_step<0>(slot0, bh, bm, bl, acc, lane, c);
            __builtin_amdgcn_sched_barrier(0);
            if (net == 0) B3_STAMP((l - 1) * 8 + 4);
            // ---- Q4 under this layer's tiles 0, 1 epilogue
            mt[0] = mt[1] = 0u;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                b3_step<2>(slot0, bh, bm, bl, acc, lane, 4 + c);
                b3_epilogue_eighth<PARK>(acc, hl, mt, pt, h, bh, bm, bl, c >> 1, c & 1);
            }
            b3_pin_slices<0>(bh, bm, bl);
            asm volatile("" : "+v"(mt[0]), "+v"(mt[1]));
            b3_interleave<48, B3_Q_FILL>();
            __builtin_amdgcn_sched_barrier(0);
            if (net == 0) B3_STAMP((l - 1) * 8 + 5);
            // ---- E: slot 0 is free; the block issued at M has landed
            b3_point<KEEP>();
            if (l < 4)
                b3_slot_issue<B3_KH_F4, NT>(wn, B3_L0_F4 + (2 * l + 1) * B3_KH_F4, 0);    // k1 of layer l + 1
            else if (net == 0)
                b3_slot_issue<B3_L0_F4, NT>(w3_t, 0, 0);                                  // the other net's L0
            b3_dma_fence();
            if (net == 0) B3_STAMP((l - 1) * 8 + 6);
            if (net == 0 && l == 4) B3_STAMP_REAL(63);
        }
        // layer 5: 128 -> 3 | 2 (one padded tile, slot 1); k16 steps 0..3 under layer 4's tiles 2, 3 epilogue
        f32x16 o[1];
        b3_acc_bias_row<0, 1, 1>(o, h, 1);
        mt[2] = mt[3] = 0u;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            b3_l5_step(slot1, bh, bm, bl, o[0], lane, c);
            b3_epilogue_eighth<PARK>(acc, ht + 4 * 128 * TILE, mt, pt, h, bh, bm, bl, 2 + (c >> 1), c & 1);
        }
        b3_pin_slices<4>(bh, bm, bl);
        asm volatile("" : "+v"(mt[2]), "+v"(mt[3]));
        b3_interleave<24, 12>();
        __builtin_amdgcn_sched_barrier(0);
        if (PARK) mk[(net * 5 + 4) * 64 + lane] = make_uint2(mt[0] | (mt[1] << 16), mt[2] | (mt[3] << 16));
#pragma unroll
        for (int s = 4; s < 8; s++) b3_l5_step(slot1, bh, bm, bl, o[0], lane, s);
        b3_point<0>();                                     // slot 1 is free (net 0: the other net's L0 has landed)
        if (net == 0) {
            b3_slot_issue<B3_KH_F4, NT>(w3_t, B3_L0_F4, 1);                                // L1t.k0
            b3_bias_issue(bias_t, 0, 32, 1);
        }
        if (h == 0 && p < M && tile_raw < n_tiles) {
            if (net == 0) {
                out_deform[p * 3 + 0] = o[0][0];
                out_deform[p * 3 + 1] = o[0][1];
                out_deform[p * 3 + 2] = o[0][2];
            } else {
                out_topo[p * 2 + 0] = o[0][0];
                out_topo[p * 2 + 1] = o[0][1];
            }
        }
    }
}

// ---- backward-data -------------------------------------------------------------------------------------------------
// Same chain, transposed packs (T5, T4..T1, T0), ReLU derivative from the sign masks the forward parked; parks dPre tiles in
// mlp.hip's layout for mh_mlp_wgrad.
#define B3_T5_F4 1536                                   // 3 planes x 4 tiles x 2 k16 steps x 64 lanes
#define B3_T0_F4 3072                                   // 3 planes x 2 tiles x 8 k16 steps x 64 lanes
#define B3_NETT_F4 (B3_T5_F4 + 4 * B3_LH_F4 + B3_T0_F4)  // 29 184 float4 per net

// masked accumulators -> parked dPre tile (PARK) + the next transposed layer's B operands
template <bool PARK = true>
__device__ __forceinline__ void b3_epilogue_bwd(const f32x16 (&acc)[4], uint2 m, float *__restrict__ dt, int pt, int h, Frag (&bh)[8],
                                                Frag (&bm)[8], Frag (&bl)[8]) {
    mfma_results_settle();
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const uint32_t mw = (t < 2 ? m.x : m.y) >> (16 * (t & 1));
        float y[16];
#pragma unroll
        for (int r = 0; r < 16; r++) y[r] = mask_bit(mw, r, acc[t][r]);
        if (PARK) {
#pragma unroll
            for (int r = 0; r < 16; r++) PARK_STORE(y[r], &dt[(32 * t + acc_row(r, h)) * TILE + pt]);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
            for (int e2 = 0; e2 < 4; e2++)
                split2(y[8 * s2 + 2 * e2], y[8 * s2 + 2 * e2 + 1], bh[2 * t + s2].u[e2], bm[2 * t + s2].u[e2], bl[2 * t + s2].u[e2]);
    }
}

// backward flavour of b3_epilogue_eighth: registers 8 s2 .. 8 s2 + 7 of tile t masked by their ReLU bits -> parked dPre, slices
__device__ __forceinline__ void b3_epilogue_bwd_eighth(f32x16 (&acc)[4], uint2 m, float *__restrict__ dt, int pt, int h, Frag (&bh)[8],
                                                       Frag (&bm)[8], Frag (&bl)[8], int t, int s2) {
    const uint32_t mw = (t < 2 ? m.x : m.y) >> (16 * (t & 1));
#pragma unroll
    for (int r = 8 * s2; r < 8 * s2 + 8; r++) acc[t][r] = mask_bit(mw, r, acc[t][r]);
#pragma unroll
    for (int r = 8 * s2; r < 8 * s2 + B3_PARK_STORES_EIGHTH; r++) PARK_STORE(acc[t][r], &dt[(32 * t + acc_row(r, h)) * TILE + pt]);
#pragma unroll
    for (int e2 = 0; e2 < 4; e2++)
        split2(acc[t][8 * s2 + 2 * e2], acc[t][8 * s2 + 2 * e2 + 1], bh[2 * t + s2].u[e2], bm[2 * t + s2].u[e2], bl[2 * t + s2].u[e2]);
#ifdef B3_Q4_CHUNKS
    Frag &fh = bh[2 * t + s2], &fm = bm[2 * t + s2], &fl = bl[2 * t + s2];
    asm volatile("" : "+v"(fh.u[0]), "+v"(fh.u[1]), "+v"(fh.u[2]), "+v"(fh.u[3]), "+v"(fm.u[0]), "+v"(fm.u[1]), "+v"(fm.u[2]),
                 "+v"(fm.u[3]), "+v"(fl.u[0]), "+v"(fl.u[1]), "+v"(fl.u[2]), "+v"(fl.u[3]));
#endif
}

// PARK4 = false (round 6): dPre4 is not parked -- the layer-4 weight-gradient launch regenerates it (mlp.hip: wgrad_regen_b3_kernel)
template <int NW, bool PARK4 = true>
__global__ __launch_bounds__(NW * 64, 2) void warp_bwd_b3_kernel(const float *__restrict__ x, const float *__restrict__ g_deform,
                                                                 const float *__restrict__ g_topo, const f32x4 *__restrict__ w3T_d,
                                                                 const f32x4 *__restrict__ w3T_t, int n_bands,
                                                                 const float *__restrict__ acts, float *__restrict__ dpre,
                                                                 float *__restrict__ g_x, int64_t M, int64_t n_tiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pt = lane & 31, h = lane >> 5;
    const int64_t tile_raw = (int64_t)blockIdx.x * NW + wave;
    // the scratch holds whole 128-point blocks: a wave beyond it (tail of the last 256-point workgroup) re-runs the LAST tile and
    // stores the same bytes again (no parking store behind a branch: see warp_fwd_b3_kernel)
    const bool have = tile_raw < n_tiles;
    const int64_t tile_id = have ? tile_raw : n_tiles - 1;
    const int64_t p = tile_id * TILE + pt;
    const bool live = p < M;
    const float *atile = acts + tile_id * (int64_t)(WARP_ACT_ROWS * TILE);
    float *dtile = dpre + tile_id * (int64_t)(WARP_DPRE_ROWS * TILE);
    float gx[3] = {0.f, 0.f, 0.f};

    b3_stage_issue<B3_T5_F4, NW * 64>(w3T_d);
    for (int net = 0; net < 2; net++) {
        const f32x4 *wt = net ? w3T_t : w3T_d;
        const float *g = net ? g_topo : g_deform;
        const int nout = net ? 2 : 3;
        float *dt = dtile + net * 672 * TILE;
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
        // dPre5's live rows only (0..nout-1 <= 3: registers 0..3 of the lower half; mh_mlp_wgrad reads no row behind them, wg_row)
#ifdef MH_PARK_PAD_ROWS
        store_acc_rows<1>(dt + 640 * TILE, d5, pt, h);
#else
        if (h == 0) {
#pragma unroll
            for (int r = 0; r < 4; r++) dt[(640 + r) * TILE + pt] = d5[r];
        }
#endif
        Frag bh[8], bm[8], bl[8];
#pragma unroll
        for (int s = 0; s < 2; s++)
#pragma unroll
            for (int e2 = 0; e2 < 4; e2++) split2(d5[8 * s + 2 * e2], d5[8 * s + 2 * e2 + 1], bh[s].u[e2], bm[s].u[e2], bl[s].u[e2]);
        // dH5 = W5^T dPre5
        f32x16 acc[4];
        b3_stage_wait();
        b3_layer<2, 4, true>(lds_b3, bh, bm, bl, acc, lane);
        __syncthreads();
        wt += B3_T5_F4;
        b3_stage_issue<B3_LH_F4, NW * 64>(wt);
        b3_dma_fence();
        // dPre_4 from T5's output (the short first stage: nothing to hide it under)
        b3_epilogue_bwd<PARK4>(acc, msk[4], dt + 4 * 128 * TILE, pt, h, bh, bm, bl);
        for (int l = 4; l >= 1; l--) {
            // dH_l = W_l^T dPre_l in quarters (see the forward kernel): tiles 0, 1 are complete after the third quarter and
            // their half of dPre_{l-1} (mask by H_l's ReLU bits, park, slice) runs under the fourth quarter's MFMAs
            if (!PARK4 && l == 4)
                b3_stage_wait();                             // no parking stores behind this DMA: vmcnt(KEEP) would not cover it
            else
                b3_stage_wait_keep<B3_KEEP_BWD>();           // behind the DMA: the 32 parking stores of tiles 2, 3
            acc_zero<4>(acc);
            b3_quarter<0, 0>(lds_b3, bh, bm, bl, acc, lane);
            b3_quarter<2, 0>(lds_b3, bh, bm, bl, acc, lane);
            b3_quarter<0, 4>(lds_b3, bh, bm, bl, acc, lane);
            __builtin_amdgcn_sched_barrier(0);
#ifdef B3_Q4_CHUNKS
#pragma unroll
            for (int c = 0; c < 4; c++) {
                b3_quarter_step<2>(lds_b3, bh, bm, bl, acc, lane, 4 + c);
                b3_epilogue_bwd_eighth(acc, msk[l - 1], dt + (l - 1) * 128 * TILE, pt, h, bh, bm, bl, c >> 1, c & 1);
                __builtin_amdgcn_sched_barrier(0);
            }
#else
#pragma unroll
            for (int c = 0; c < 4; c++) {
                b3_quarter_step<2>(lds_b3, bh, bm, bl, acc, lane, 4 + c);
                b3_epilogue_bwd_eighth(acc, msk[l - 1], dt + (l - 1) * 128 * TILE, pt, h, bh, bm, bl, c >> 1, c & 1);
            }
            b3_pin_slices<0>(bh, bm, bl);
#pragma unroll
            for (int g_ = 0; g_ < 48; g_++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, B3_Q4_FILL, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#endif
            __syncthreads();
            wt += B3_LH_F4;
            if (l > 1)
                b3_stage_issue<B3_LH_F4, NW * 64>(wt);
            else if (g_x)
                b3_stage_issue<B3_T0_F4, NW * 64>(wt);
            else if (net == 0)
                b3_stage_issue<B3_T5_F4, NW * 64>(w3T_t);
            b3_dma_fence();
#pragma unroll
            for (int c = 0; c < 4; c++)
                b3_epilogue_bwd_eighth(acc, msk[l - 1], dt + (l - 1) * 128 * TILE, pt, h, bh, bm, bl, 2 + (c >> 1), c & 1);
        }
        if (g_x) {
            // d(enc features) = W0^T dPre0; rows ordered (kk = 16t + r, h = lane>>5).  Skipped when nobody asks for d/dx
            f32x16 e[2];
            b3_stage_wait_keep<B3_KEEP_BWD>();
            b3_layer<8, 2, true>(lds_b3, bh, bm, bl, e, lane);
            __syncthreads();
            if (net == 0) b3_stage_issue<B3_T5_F4, NW * 64>(w3T_t);
            mfma_results_settle();
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
#pragma unroll
    for (int d = 0; d < 3; d++) gx[d] += __shfl_xor(gx[d], 32);
    if (g_x && live && have && h == 0) {
        g_x[p * 3 + 0] = gx[0];
        g_x[p * 3 + 1] = gx[1];
        g_x[p * 3 + 2] = gx[2];
    }
}

// ---- canonical field forward (sdf_net + Laplace density, color_net; models/model.py:273-307) on the bf16 pipe ------------
// Same structure as mlp.hip's field_fwd_kernel: a persistent 8-wave workgroup per CU keeps EVERY layer's weight operand in
// LDS (here the three bf16 planes of all six layers: 144 KB), a wave owns a 32-point tile through both nets with no barrier
// at all, and parks the tile in exactly the fp32 kernel's layout (mh_field_bwd_fused consumes it).
//   sliced pack (packing.py, field_joint_packer().b3_layers), float4 offsets: S0 0 (K = 80: 5 k16 steps, 2 tiles), S1 2048,
//   S2 3584, C0 5120, C1 6656 (K = 64: 4 steps, 2 tiles), C2 8192 (1 tile); 9216 float4 in all
#define FB3_S0 0
#define FB3_S1 2048
#define FB3_S2 3584
#define FB3_C0 5120
#define FB3_C1 6656
#define FB3_C2 8192
#define FB3_F4 9216
#define FB3_THREADS 512

__device__ __forceinline__ float laplace_sigma_b3(float s, float beta) {
    // density.py:22-31, cancellation-free form (mlp_dev.h: laplace_unit)
    return (1.0f / beta) * laplace_unit(s, beta);
}

// 64 activations of a lane (two accumulator tiles): ReLU, park feature-major, sign mask, slices of k16 steps 0..3
__device__ __forceinline__ void fb3_epilogue(f32x16 (&acc)[2], float *__restrict__ ht, uint32_t *__restrict__ mk, int pt, int h,
                                             Frag (&bh)[8], Frag (&bm)[8], Frag (&bl)[8]) {
    uint32_t m = 0;
#pragma unroll
    for (int t = 1; t >= 0; t--) {
#pragma unroll
        for (int r = 15; r >= 0; r--) {
            acc[t][r] = relu_i(acc[t][r]);
            m = push_nz(m, acc[t][r]);
        }
    }
#pragma unroll
    for (int t = 0; t < 2; t++) {
        if (ht) {
#pragma unroll
            for (int r = 0; r < 16; r++) PARK_STORE(acc[t][r], &ht[(32 * t + acc_row(r, h)) * TILE + pt]);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
            for (int e2 = 0; e2 < 4; e2++)
                split2(acc[t][8 * s2 + 2 * e2], acc[t][8 * s2 + 2 * e2 + 1], bh[2 * t + s2].u[e2], bm[2 * t + s2].u[e2],
                       bl[2 * t + s2].u[e2]);
    }
    if (mk) *mk = m;
}

// the same without the slices: the layer in front of a VALU-evaluated row (sdf-only pass)
__device__ __forceinline__ void fb3_epilogue_plain(f32x16 (&acc)[2], float *__restrict__ ht, uint32_t *__restrict__ mk, int pt, int h) {
    uint32_t m = 0;
#pragma unroll
    for (int t = 1; t >= 0; t--) {
#pragma unroll
        for (int r = 15; r >= 0; r--) {
            acc[t][r] = relu_i(acc[t][r]);
            m = push_nz(m, acc[t][r]);
        }
    }
    if (ht) {
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) PARK_STORE(acc[t][r], &ht[(32 * t + acc_row(r, h)) * TILE + pt]);
    }
    if (mk) *mk = m;
}

__global__ __launch_bounds__(FB3_THREADS, 2) void field_fwd_b3_kernel(
    const float *__restrict__ xc, const float *__restrict__ feat_s, const float *__restrict__ feat_c, const float *__restrict__ topo,
    const f32x4 *__restrict__ w3, const float *__restrict__ bias, const float *__restrict__ beta_p, int n_bands, int with_color,
    float *__restrict__ sdf, float *__restrict__ sigma, float *__restrict__ albedo, float *__restrict__ acts, int64_t M,
    int64_t n_tiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pt = lane & 31, h = lane >> 5;
    {
        const int n = with_color ? FB3_F4 : FB3_C0;      // the sdf-only pass (finite-difference taps) needs S0..S2 only
        for (int i = threadIdx.x; i < n; i += FB3_THREADS) lds_b3[i] = w3[i];
    }
    __syncthreads();
    // sdf-only pass: of the last layer's 33 outputs only the sdf row is wanted -- a 64-term dot product per point, 32 fused
    // multiply-adds per lane on the activations it holds, instead of slicing them (176 instructions) for 24 dependent MFMAs.
    // Its weights as fp32, once per block: hi + mid + lo of the slices IS the fp32 weight; [half h][k-step order] behind the sdf
    // net's blocks (the colour blocks are not staged on this pass).  A-fragment lane 32 h of tile 1 holds row 0's steps.
    float *wsdf = reinterpret_cast<float *>(lds_b3 + FB3_C0);
    if (!with_color) {
        if (threadIdx.x < 64) {
            const int hh = threadIdx.x >> 5, kk = threadIdx.x & 31, s = kk >> 3, e = kk & 7;
            float v[3];
#pragma unroll
            for (int pl = 0; pl < 3; pl++) {
                Frag f;
                f.f = lds_b3[FB3_S2 + 4 * 64 + pl * 512 + s * 64 + 32 * hh];
                const uint32_t w = f.u[e >> 1];
                v[pl] = __uint_as_float((e & 1) ? (w & 0xffff0000u) : (w << 16));
            }
            wsdf[hh * 32 + kk] = (v[0] + v[1]) + v[2];
        }
        __syncthreads();
    }
    for (int64_t tile_id = (int64_t)blockIdx.x * (FB3_THREADS / 64) + wave; tile_id < n_tiles;
         tile_id += (int64_t)gridDim.x * (FB3_THREADS / 64)) {
        const int64_t p = tile_id * TILE + pt;
        const bool live = p < M;
        const int64_t pc = live ? p : M - 1;
        float xv[3] = {xc[pc * 3 + 0], xc[pc * 3 + 1], xc[pc * 3 + 2]};
        float *tile = acts ? acts + tile_id * (int64_t)(FIELD_ACT_ROWS * TILE) : nullptr;
        uint32_t *mk = tile ? reinterpret_cast<uint32_t *>(tile + FIELD_HID_ROWS * TILE) + lane : nullptr;
        Frag bh[8], bm[8], bl[8];
        f32x16 acc[2];
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
                for (int k = 0; k < 40; k++) PARK_STORE(bin0[k], &tile[(2 * k + h) * TILE + pt]);
                // rows 80..95 of the block are padding of the 96-row k extent: neither written here nor read by the fused backward,
                // whose pad lanes read the zero row 79 instead (mlp.hip: layer s0) -- 64 B per point less to park (round 6, same-box
                // A/B profiles/r06_ab_field_park_rows.txt: field forward -3 % at cfg3, -5 % on a whole view's tap passes)
            }
#pragma unroll
            for (int s = 0; s < 5; s++)
#pragma unroll
                for (int e2 = 0; e2 < 4; e2++) split2(bin0[8 * s + 2 * e2], bin0[8 * s + 2 * e2 + 1], bh[s].u[e2], bm[s].u[e2], bl[s].u[e2]);
        }
        // sdf L0: 73 -> 64
        acc_bias<2>(acc, bias, h);
        b3_layer<5, 2>(lds_b3 + FB3_S0, bh, bm, bl, acc, lane);
        fb3_epilogue(acc, tile ? tile + 96 * TILE : nullptr, mk ? mk + 0 * 64 : nullptr, pt, h, bh, bm, bl);
        // sdf L1: 64 -> 64
        acc_bias<2>(acc, bias + 64, h);
        b3_layer<4, 2>(lds_b3 + FB3_S1, bh, bm, bl, acc, lane);
        // sdf L2: 64 -> [geo(32) | sdf], no activation
        if (!with_color) {
            fb3_epilogue_plain(acc, tile ? tile + 160 * TILE : nullptr, mk ? mk + 1 * 64 : nullptr, pt, h);
            const f32x4 *wv = reinterpret_cast<const f32x4 *>(wsdf + 32 * h);      // the same address in every lane of a half
            float sv = 0.f;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const f32x4 w4 = wv[q];
#pragma unroll
                for (int c = 0; c < 4; c++) sv = fmaf(w4[c], acc[q >> 2][4 * (q & 3) + c], sv);
            }
            sv += __shfl_xor(sv, 32);
            sv += bias[128 + 32];
            if (h == 0 && live) {
                sdf[p] = sv;
                if (sigma) sigma[p] = laplace_sigma_b3(sv, *beta_p);
            }
            continue;
        }
        fb3_epilogue(acc, tile ? tile + 160 * TILE : nullptr, mk ? mk + 1 * 64 : nullptr, pt, h, bh, bm, bl);
        acc_bias<2>(acc, bias + 128, h);
        b3_layer<4, 2>(lds_b3 + FB3_S2, bh, bm, bl, acc, lane);
        if (h == 0 && live) {
            const float sv = acc[1][0];
            sdf[p] = sv;
            if (sigma) sigma[p] = laplace_sigma_b3(sv, *beta_p);
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
            for (int r = 0; r < 16; r++) binc[16 + r] = acc[0][r];
            if (tile) {
#pragma unroll
                for (int k = 0; k < 32; k++) PARK_STORE(binc[k], &tile[(224 + 2 * k + h) * TILE + pt]);
            }
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int e2 = 0; e2 < 4; e2++) split2(binc[8 * s + 2 * e2], binc[8 * s + 2 * e2 + 1], bh[s].u[e2], bm[s].u[e2], bl[s].u[e2]);
        }
        acc_bias<2>(acc, bias + 192, h);
        b3_layer<4, 2>(lds_b3 + FB3_C0, bh, bm, bl, acc, lane);
        fb3_epilogue(acc, tile ? tile + 288 * TILE : nullptr, mk ? mk + 2 * 64 : nullptr, pt, h, bh, bm, bl);
        // color L1
        acc_bias<2>(acc, bias + 256, h);
        b3_layer<4, 2>(lds_b3 + FB3_C1, bh, bm, bl, acc, lane);
        fb3_epilogue(acc, tile ? tile + 352 * TILE : nullptr, mk ? mk + 3 * 64 : nullptr, pt, h, bh, bm, bl);
        // color L2: 64 -> 3, sigmoid
        f32x16 o[1];
        acc_bias<1>(o, bias + 320, h);
        b3_layer<4, 1>(lds_b3 + FB3_C2, bh, bm, bl, o, lane);
        if (h == 0 && live) {
#pragma unroll
            for (int c = 0; c < 3; c++) albedo[p * 3 + c] = 1.0f / (1.0f + expf(-o[0][c]));
        }
    }
}

// ---- fp32 fragments (b3 order, one gather of the natural weights on the host side) -> three bf16 planes per layer
#define B3_MAX_LAYERS 32
struct B3Layers {
    int n_layers;
    int src_off[B3_MAX_LAYERS];   // floats
    int n8[B3_MAX_LAYERS];        // groups of 8 floats (= float4 of bf16 per plane)
    int dst_off[B3_MAX_LAYERS];   // float4 units
    int g_end[B3_MAX_LAYERS];     // running end of the layers' group ranges
};

__global__ void b3_slice_kernel(const float *__restrict__ src, f32x4 *__restrict__ dst, B3Layers L) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= L.g_end[L.n_layers - 1]) return;
    int l = 0;
    while (g >= L.g_end[l]) l++;
    const int i = g - (l ? L.g_end[l - 1] : 0);
    const f32x4 a = *reinterpret_cast<const f32x4 *>(src + L.src_off[l] + 8 * i);
    const f32x4 b = *reinterpret_cast<const f32x4 *>(src + L.src_off[l] + 8 * i + 4);
    Frag fh, fm, fl;
    split2(a[0], a[1], fh.u[0], fm.u[0], fl.u[0]);
    split2(a[2], a[3], fh.u[1], fm.u[1], fl.u[1]);
    split2(b[0], b[1], fh.u[2], fm.u[2], fl.u[2]);
    split2(b[2], b[3], fh.u[3], fm.u[3], fl.u[3]);
    f32x4 *d = dst + L.dst_off[l] + i;
    d[0] = fh.f;
    d[L.n8[l]] = fm.f;
    d[2 * L.n8[l]] = fl.f;
}

extern "C" int mh_b3_slice(const float *src, void *dst, int32_t n_layers, const int32_t *src_off_host, const int32_t *n_host,
                           const int32_t *dst_off_f4_host, void *stream) {
    if (n_layers == 0) return MH_OK;
    if (!src || !dst || n_layers < 0 || n_layers > B3_MAX_LAYERS || !src_off_host || !n_host || !dst_off_f4_host) return MH_ERR_ARG;
    B3Layers L;
    L.n_layers = n_layers;
    int end = 0;
    for (int l = 0; l < n_layers; l++) {
        if (n_host[l] <= 0 || n_host[l] % 8 || src_off_host[l] % 4 || src_off_host[l] < 0 || dst_off_f4_host[l] < 0) return MH_ERR_ARG;
        L.src_off[l] = src_off_host[l];
        L.n8[l] = n_host[l] / 8;
        L.dst_off[l] = dst_off_f4_host[l];
        end += n_host[l] / 8;
        L.g_end[l] = end;
    }
    hipLaunchKernelGGL(b3_slice_kernel, dim3((end + 255) / 256), dim3(256), 0, mh_stream(stream), src,
                       reinterpret_cast<f32x4 *>(dst), L);
    MH_CHECK_LAUNCH();
    return MH_OK;
}

extern "C" int64_t mh_warp_w3_bytes(void) { return (int64_t)B3_NET_F4 * 16; }

static int b3_lds_opt_in() {
    static MhOncePerDevice done;
    const int dev = mh_device();
    if (done.need(dev)) {
        if (hipFuncSetAttribute((const void *)warp_fwd_b3_kernel<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, B3_LDS_BYTES) !=
                hipSuccess ||
            hipFuncSetAttribute((const void *)warp_fwd_b3_kernel<8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, B3_LDS_BYTES) !=
                hipSuccess ||
            hipFuncSetAttribute((const void *)warp_fwd_b3_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, B3_LDS_BYTES) !=
                hipSuccess ||
            hipFuncSetAttribute((const void *)warp_bwd_b3_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, B3_LDS_BYTES) !=
                hipSuccess ||
            hipFuncSetAttribute((const void *)(warp_bwd_b3_kernel<8, false>), hipFuncAttributeMaxDynamicSharedMemorySize, B3_LDS_BYTES) !=
                hipSuccess ||
            hipFuncSetAttribute((const void *)warp_fwd_b3_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, B3_LDS_BYTES) !=
                hipSuccess ||
            hipFuncSetAttribute((const void *)warp_bwd_b3_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, B3_LDS_BYTES) !=
                hipSuccess)
            return MH_ERR_LAUNCH;
        done.mark(dev);
    }
    return MH_OK;
}

extern "C" int64_t mh_field_w3_bytes(void) { return (int64_t)FB3_F4 * 16; }

extern "C" int mh_field_fwd_b3(const float *xc, const float *feat_s, const float *feat_c, const float *topo, const void *w3,
                               const float *bias, const float *beta, int32_t n_bands, int32_t with_color, float *sdf, float *sigma,
                               float *albedo, float *acts, int64_t M, void *stream) {
    if (M == 0) return MH_OK;
    if (M < 0 || !xc || !feat_s || !w3 || !bias || !sdf || n_bands < 0 || n_bands > 6 || !beta) return MH_ERR_ARG;
    if (with_color && (!feat_c || !albedo)) return MH_ERR_ARG;
    const int64_t n_tiles = mh_mlp_tiles(M);   // dead tail tiles are processed too: the backward reads every scratch tile
    static MhOncePerDevice ok;
    const int dev = mh_device();
    if (ok.need(dev)) {
        if (hipFuncSetAttribute((const void *)field_fwd_b3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FB3_F4 * 16) != hipSuccess)
            return MH_ERR_LAUNCH;
        ok.mark(dev);
    }
    const int64_t need = (n_tiles + FB3_THREADS / 64 - 1) / (FB3_THREADS / 64);
    const int64_t cus = mh_cu_count();
    hipLaunchKernelGGL(field_fwd_b3_kernel, dim3((unsigned)(need < cus ? need : cus)), dim3(FB3_THREADS), FB3_F4 * 16,
                       mh_stream(stream), xc, feat_s, feat_c, topo, reinterpret_cast<const f32x4 *>(w3), bias, beta, (int)n_bands,
                       (int)with_color, sdf, sigma, albedo, acts, M, n_tiles);
    MH_CHECK_LAUNCH();
    return MH_OK;
}

extern "C" int64_t mh_warp_w3T_bytes(void) { return (int64_t)B3_NETT_F4 * 16; }

// Workgroup shape by batch size.  Large batches: one 8-wave workgroup per 256 points (two waves per SIMD share one LDS copy of a
// layer's slices and fill each other's epilogues).  Small batches are latency-bound -- a call on 2 048 points (the surface-point
// query of a training step) is 8 such workgroups on 256 CUs and takes as long as one on 22 000 points: every SIMD walks two
// tiles through 12 layer steps -- so up to one 128-point workgroup per CU (32 768 points on MI355X) the 4-wave shape runs one
// tile per SIMD and twice the CUs.
static inline bool b3_small_batch(int64_t M) { return M <= (int64_t)BLOCK_PTS * mh_cu_count(); }

extern "C" int mh_warp_bwd_data_b3(const float *x, const float *g_deform, const float *g_topo, const void *w3T_d, const void *w3T_t,
                                   int32_t n_bands, const float *acts, float *dpre, float *g_x, int64_t M, int32_t skip_dpre4,
                                   void *stream) {
    if (M == 0) return MH_OK;
    if (M < 0 || !x || !w3T_d || !w3T_t || !acts || !dpre || n_bands < 0 || n_bands > 6) return MH_ERR_ARG;
    const bool small = b3_small_batch(M);
    if (skip_dpre4 && small) return MH_ERR_ARG;       // (mh_warp_regen_dpre4(M) is never set for such a batch)
    const int64_t blocks = small ? (M + BLOCK_PTS - 1) / BLOCK_PTS : (M + B3_BLOCK_PTS - 1) / B3_BLOCK_PTS;
    if (blocks > 0x7fffffffLL) return MH_ERR_ARG;
    if (b3_lds_opt_in() != MH_OK) return MH_ERR_LAUNCH;
    if (small)
        hipLaunchKernelGGL(warp_bwd_b3_kernel<4>, dim3((unsigned)blocks), dim3(256), B3_LDS_BYTES, mh_stream(stream), x, g_deform,
                           g_topo, reinterpret_cast<const f32x4 *>(w3T_d), reinterpret_cast<const f32x4 *>(w3T_t), (int)n_bands,
                           acts, dpre, g_x, M, mh_mlp_tiles(M));
    else if (skip_dpre4)
        hipLaunchKernelGGL((warp_bwd_b3_kernel<8, false>), dim3((unsigned)blocks), dim3(B3_THREADS), B3_LDS_BYTES, mh_stream(stream), x,
                           g_deform, g_topo, reinterpret_cast<const f32x4 *>(w3T_d), reinterpret_cast<const f32x4 *>(w3T_t),
                           (int)n_bands, acts, dpre, g_x, M, mh_mlp_tiles(M));
    else
        hipLaunchKernelGGL(warp_bwd_b3_kernel<8>, dim3((unsigned)blocks), dim3(B3_THREADS), B3_LDS_BYTES, mh_stream(stream), x,
                           g_deform, g_topo, reinterpret_cast<const f32x4 *>(w3T_d), reinterpret_cast<const f32x4 *>(w3T_t),
                           (int)n_bands, acts, dpre, g_x, M, mh_mlp_tiles(M));
    MH_CHECK_LAUNCH();
    return MH_OK;
}

extern "C" int mh_warp_fwd_b3(const float *x, const int32_t *slot, const float *bias0_d, const float *bias0_t, const void *w3_d,
                              const void *w3_t, const float *bias_d, const float *bias_t, int32_t n_bands, float *out_deform,
                              float *out_topo, float *acts, int64_t M, void *stream) {
    if (M == 0) return MH_OK;
    if (M < 0 || !x || !bias0_d || !bias0_t || !w3_d || !w3_t || !bias_d || !bias_t || !out_deform || !out_topo || n_bands < 0 ||
        n_bands > 6)
        return MH_ERR_ARG;
    const bool small = b3_small_batch(M);
    const int64_t blocks = small ? (M + BLOCK_PTS - 1) / BLOCK_PTS : (M + B3_BLOCK_PTS - 1) / B3_BLOCK_PTS;
    if (blocks > 0x7fffffffLL) return MH_ERR_ARG;
    if (b3_lds_opt_in() != MH_OK) return MH_ERR_LAUNCH;
#define B3_FWD_LAUNCH(NW_, PARK_, THREADS_)                                                                                          \
    hipLaunchKernelGGL((warp_fwd_b3_kernel<NW_, PARK_>), dim3((unsigned)blocks), dim3(THREADS_), B3_LDS_BYTES, mh_stream(stream), x, slot, \
                       bias0_d, bias0_t, reinterpret_cast<const f32x4 *>(w3_d), reinterpret_cast<const f32x4 *>(w3_t), bias_d, bias_t, \
                       (int)n_bands, out_deform, out_topo, acts, M, mh_mlp_tiles(M))
    if (small) {
        if (acts)
            B3_FWD_LAUNCH(4, true, 256);
        else
            B3_FWD_LAUNCH(4, false, 256);
    } else {
        if (acts)
            B3_FWD_LAUNCH(8, true, B3_THREADS);
        else
            B3_FWD_LAUNCH(8, false, B3_THREADS);
    }
#undef B3_FWD_LAUNCH
    MH_CHECK_LAUNCH();
    return MH_OK;
}
