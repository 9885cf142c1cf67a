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
#include <stdlib.h>
#include <atomic>

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

// the rows of a cell's eight corners (corner c: bit 0 -> x + 1, bit 1 -> y + 1, bit 2 -> z + 1): the per-axis terms of
// get_grid_index are formed ONCE for the two y and the two z values -- four 32-bit multiplies (v_mul_lo_u32: quarter rate) per cell
// instead of two or three per corner; the same integers as grid_row() corner by corner
__device__ __forceinline__ void grid_rows8(const uint32_t (&g)[3], const uint32_t (&g1)[3], uint32_t res, uint32_t T, bool dense,
                                           bool pow2, uint32_t (&row)[8]) {
    // branch-free over the lanes of a wave (its 16 level-lanes mix dense and hashed levels, and a divergent branch per corner cost
    // ~100 scalar instructions and 30 taken branches per point): the axis multipliers are selected once, both combinations are
    // formed (two adds / two xors per corner pair) and one select keeps the lane's.  A hashed level whose table size is NOT a
    // power of two (no shipped geometry has one: a hashed level's size is the 2^k cap) takes the modulo in ONE rarely entered
    // branch for all eight corners.
    const uint32_t my = dense ? res : 2654435761u, mz = dense ? res * res : 805459861u;
    const uint32_t ay[2] = {g[1] * my, g1[1] * my}, az[2] = {g[2] * mz, g1[2] * mz};
    // (the select is a bit merge, v_bfi_b32, on a per-lane all-ones / all-zeros word: written as `dense ? a : b` hipcc turns it
    //  back into eight divergent branches)
    const uint32_t mask = pow2 ? T - 1 : 0xffffffffu, sel = dense ? 0xffffffffu : 0u;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const uint32_t cx = (c & 1) ? g1[0] : g[0], y = ay[(c >> 1) & 1], z = az[(c >> 2) & 1];
        row[c] = ((cx + y + z) & sel) | ((cx ^ y ^ z) & mask & ~sel);
    }
    if (__builtin_expect(!dense && !pow2, 0)) {
#pragma unroll
        for (int c = 0; c < 8; c++) row[c] %= T;
    }
}

// n / d, correctly rounded (what the reference's fp32 `/` gives), for a divisor that stays the same over many quotients: hipcc
// expands an IEEE division into ~14 VALU instructions (scale, reciprocal, two refinements, fix-up), and every (point, level) lane
// of the grid kernels did three of them -- a tenth of the brick backward's instruction stream.  With r = RN(1 / d) (ONE division,
// hoisted out of the loops: d = 2 bound is a kernel argument) two residual corrections give the correctly rounded quotient
// (Markstein: q' = RN(q + RN(n - d q) r) is correctly rounded once q is within an ulp and r is the correctly rounded reciprocal);
// checked against exact rational arithmetic on 2.2e5 numerators for seven divisors (tools/micro/README.md).  No scaling: the
// operands here are O(1), far from the subnormal / overflow ranges the compiler's fix-up instructions exist for.
__device__ __forceinline__ float exact_div(float n, float d, float r) {
    const float q0 = n * r;
    const float q1 = fmaf(fmaf(-d, q0, n), r, q0);
    return fmaf(fmaf(-d, q1, n), r, q1);
}

// position inside level: returns false if the point is outside [0,1]^3
__device__ __forceinline__ bool grid_locate(const float (&xv)[3], float bound, float two_bound, uint32_t res, uint32_t g[3],
                                            float f[3]) {
    bool inb = true;
    const float inv = 1.0f / two_bound;                // loop-invariant in every caller
#pragma unroll
    for (int d = 0; d < 3; d++) {
        float u = exact_div(xv[d] + bound, two_bound, inv);  // grid.py:157, fp32 add then IEEE divide
        // (an infinite coordinate makes exact_div's residual inf - inf = NaN where the IEEE quotient is +-inf: the ordered
        //  comparison sends both -- and a NaN coordinate -- out of the box, as `u < 0 || u > 1` does for the reference's +-inf)
        inb = inb && (u >= 0.0f && u <= 1.0f);
        float pos = fminf(fmaxf(fmaf(u, (float)res, -0.5f), 0.0f), (float)(res - 1));
        float fl = floorf(pos);
        g[d] = (uint32_t)fl;
        f[d] = pos - fl;
    }
    return inb;
}

__device__ __forceinline__ bool grid_locate(const float *__restrict__ x, int64_t p, float bound, float two_bound,
                                            uint32_t res, uint32_t g[3], float f[3]) {
    const float xv[3] = {x[p * 3 + 0], x[p * 3 + 1], x[p * 3 + 2]};
    return grid_locate(xv, bound, two_bound, res, g, f);
}

__global__ __launch_bounds__(256) void grid_fwd_kernel(const float *__restrict__ x, const float2 *__restrict__ emb, GridMeta meta,
                                                       float2 *__restrict__ out, int64_t M, int L, int n_levels, float bound,
                                                       float two_bound) {
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
            const uint32_t g1[3] = {min(g[0] + 1, res - 1), min(g[1] + 1, res - 1), min(g[2] + 1, res - 1)};
            uint32_t row[8];
            grid_rows8(g, g1, res, T, dense, pow2, row);
            float2 v[8];
#pragma unroll
            for (int c = 0; c < 8; c++) v[c] = tab[row[c]];
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

// Grouped forward: every G consecutive points are known to lie close together (the six finite-difference taps of one
// sample, models/model.py:367-385, laid out point-major) -- a lane walks the G points of its (group, level) and keeps the
// last cell's eight corner values in registers, so taps that share a cell (94 % of them on average: the tap offset is
// 0.13 of the finest cell) cost no gathers.  Per-point arithmetic is the ungrouped kernel's: results are bit-identical.
template <int G>
__global__ __launch_bounds__(256) void grid_fwd_grouped_kernel(const float *__restrict__ x, const float2 *__restrict__ emb,
                                                               GridMeta meta, float2 *__restrict__ out, int64_t n_groups,
                                                               int L, int n_levels, float bound, float two_bound) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t grp = gid / L;
    const int l = (int)(gid - grp * L);
    if (grp >= n_groups) return;
    if (l >= n_levels) {
#pragma unroll
        for (int i = 0; i < G; i++) out[(grp * G + i) * L + l] = make_float2(0.f, 0.f);
        return;
    }
    const uint32_t res = (uint32_t)meta.res[l];
    const uint32_t T = (uint32_t)(meta.offsets[l + 1] - meta.offsets[l]);
    const bool dense = (uint64_t)res * res * res <= (uint64_t)T;
    const bool pow2 = (T & (T - 1)) == 0;
    const float2 *tab = emb + meta.offsets[l];
    float2 v[8];
    uint32_t cg[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};  // cached cell
#pragma unroll
    for (int i = 0; i < G; i++) {
        const int64_t p = grp * G + i;
        uint32_t g[3];
        float f[3];
        float2 r = make_float2(0.f, 0.f);
        if (grid_locate(x, p, bound, two_bound, res, g, f)) {
            if (g[0] != cg[0] || g[1] != cg[1] || g[2] != cg[2]) {
                const uint32_t g1[3] = {min(g[0] + 1, res - 1), min(g[1] + 1, res - 1), min(g[2] + 1, res - 1)};
                uint32_t row[8];
                grid_rows8(g, g1, res, T, dense, pow2, row);
#pragma unroll
                for (int c = 0; c < 8; c++) v[c] = tab[row[c]];
                cg[0] = g[0], cg[1] = g[1], cg[2] = g[2];
            }
#pragma unroll
            for (int c = 0; c < 8; c++) {
                float w = ((c & 1) ? f[0] : 1.f - f[0]) * ((c & 2) ? f[1] : 1.f - f[1]) * ((c & 4) ? f[2] : 1.f - f[2]);
                r.x = fmaf(w, v[c].x, r.x);
                r.y = fmaf(w, v[c].y, r.y);
            }
        }
        out[p * L + l] = r;
    }
}

// d(loss)/d(pos) of one (point, level): dy_dx (gridencoder.cu:205-247) is res * the sum over the 4 corners of the other two axes
// of w_other * (table[right] - table[left]) per channel (the border clamp ignored on purpose), and the loss gradient contracts the
// two channels.  The contraction goes FIRST here -- s_c = gr . table[corner c], eight dot products -- so the three axis sums run on
// one scalar per corner instead of two channels: 55 VALU instructions instead of 134 for the same sum of the same products
// (associated differently: fp32 rounding-level differences to the channel-wise form, and to the reference's, which sums the
// per-level rows in a torch reduction of unspecified order anyway).
__device__ __forceinline__ void grid_ddx(const float2 (&v)[8], const float (&f)[3], float2 gr, float s, float (&dx)[3]) {
    float sc[8];
#pragma unroll
    for (int c = 0; c < 8; c++) sc[c] = fmaf(gr.y, v[c].y, gr.x * v[c].x);
    const float p[3][2] = {{1.f - f[0], f[0]}, {1.f - f[1], f[1]}, {1.f - f[2], f[2]}};
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const int a = (d + 1) % 3, b = (d + 2) % 3;
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int ba = q & 1, bb = (q >> 1) & 1;
            const int lo_c = (ba << a) | (bb << b), hi_c = lo_c | (1 << d);
            const float w = p[a][ba] * p[b][bb], diff = sc[hi_c] - sc[lo_c];
            acc = q ? fmaf(w, diff, acc) : w * diff;
        }
        dx[d] = s * acc;
    }
}

// sum over the 16 level-lanes of a point (an aligned group of 16 lanes = one DPP row), every lane gets the total.  The butterfly
// v += v[lane ^ 1], ^ 2, ^ 4, ^ 8 written as DPP operands of the adds: quad_perm [1,0,3,2], quad_perm [2,3,0,1], then
// row_half_mirror (lane i <-> 7 - i: after the first two steps a quad's lanes agree, and 7 - i lies in the other quad of the half
// row -- the value lane i ^ 4 holds) and row_mirror (i <-> 15 - i, the other half row).  The same additions in the same order as the
// __shfl_xor form, which hipcc turns into twelve ds_bpermute_b32 per point: LDS-pipe instructions with a round trip each, in the
// middle of the brick backward's dependent chain.
#define DPP_ADD(v, ctrl) ((v) + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), 0xf, 0xf, true)))
__device__ __forceinline__ float sum16(float v) {
    v = DPP_ADD(v, 0xB1);    // quad_perm:[1,0,3,2]
    v = DPP_ADD(v, 0x4E);    // quad_perm:[2,3,0,1]
    v = DPP_ADD(v, 0x141);   // row_half_mirror
    v = DPP_ADD(v, 0x140);   // row_mirror
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
                grid_ddx(v, f, gr, (float)res, dx);
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


// =====================================================================================
// Brick-binned backward.
// Device-scope float atomics on MI355X execute memory-side (the per-XCD L2s are not coherent), at
// ~12 G transactions/s: the reference's formulation -- 16 atomics per (point, level) -- costs 43 ms
// for 2.1 M points.  Here the points are counting-sorted into 16^3 spatial bricks once per step
// (shared by both encoders); one workgroup owns one brick, accumulates every level's vertex
// gradients of its points in LDS (73 KB: sum over levels of (ceil(res/16)+2)^3 vertices x 2 channels
// of int64 fixed point), and only the touched vertices are flushed with global atomics -- the 8
// corner contributions of neighbouring samples collapse on-chip first.  (The on-chip accumulation is
// 64-bit fixed point, see fx_scales below.)
// =====================================================================================
#define BRK 16
#define NBRK (BRK * BRK * BRK)
#define BRK_NODES_MAX 4608   // >= sum_l (ceil(res_l/16)+2)^3 for the 16-level 16..128 pyramid (4558)

struct BrickMeta {
    int32_t n[MH_MAX_LEVELS];        // LDS vertices per axis at level l
    int32_t lds_off[MH_MAX_LEVELS];  // first LDS vertex of level l
};

__device__ __forceinline__ int brick_of(const float *__restrict__ x, int64_t p, float bound, float two_bound) {
    int b[3];
    const float inv = 1.0f / two_bound;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const float u = exact_div(x[p * 3 + d] + bound, two_bound, inv);
        if (!(u >= 0.0f && u <= 1.0f)) return NBRK;  // out of range (or not finite): no gradient (gridencoder.cu:279-284)
        b[d] = min((int)floorf(u * (float)BRK), BRK - 1);
    }
    return (b[2] * BRK + b[1]) * BRK + b[0];
}

__global__ __launch_bounds__(256) void bin_hist_kernel(const float *__restrict__ x, int64_t M, int64_t chunk, float bound,
                                                       float two_bound, int32_t *__restrict__ block_hist) {
    __shared__ int hist[NBRK + 1];
    for (int i = threadIdx.x; i <= NBRK; i += 256) hist[i] = 0;
    __syncthreads();
    const int64_t p0 = (int64_t)blockIdx.x * chunk, p1 = min(p0 + chunk, M);
    for (int64_t p = p0 + threadIdx.x; p < p1; p += 256) atomicAdd(&hist[brick_of(x, p, bound, two_bound)], 1);
    __syncthreads();
    for (int i = threadIdx.x; i <= NBRK; i += 256) block_hist[(int64_t)blockIdx.x * (NBRK + 1) + i] = hist[i];
}

// one thread per brick: exclusive scan of its counts over the G blocks
__global__ __launch_bounds__(256) void bin_colscan_kernel(int32_t *__restrict__ block_hist, int G,
                                                          int32_t *__restrict__ brick_cnt) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b > NBRK) return;
    int run = 0, g = 0;
    // 8 loads in flight per round trip (a plain in-place loop is one dependent L2 round trip per block row)
    for (; g + 8 <= G; g += 8) {
        int c[8];
#pragma unroll
        for (int k = 0; k < 8; k++) c[k] = block_hist[(int64_t)(g + k) * (NBRK + 1) + b];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            block_hist[(int64_t)(g + k) * (NBRK + 1) + b] = run;
            run += c[k];
        }
    }
    for (; g < G; g++) {
        const int c = block_hist[(int64_t)g * (NBRK + 1) + b];
        block_hist[(int64_t)g * (NBRK + 1) + b] = run;
        run += c;
    }
    brick_cnt[b] = run;
}

// single block: exclusive scans over the NBRK+1 brick totals -> brick_start[NBRK+2], and over the
// per-brick work-item counts ceil(cnt/chunk) -> work_start[NBRK+1] (stored behind brick_start)
// points per work item.  Every item flushes its brick's touched vertices -- ~9000 global float atomics whatever its size, a
// fifth of the staged backward at cfg3 (-DBRK_EXP_NOFLUSH: 1.16 -> 0.94 ms) -- so large calls take larger items (cfg3: 1.16 ->
// 1.08 ms, the binned forward 0.36 -> 0.34); the small calls of a training step lose balance with them (real-view step 0.78 ->
// 0.92 ms at 2048, 1.15 at 4096).  The binning writes its choice behind the work-item table; the kernels read it there.
#define BRK_CHUNK_SMALL 1024
#ifndef BRK_CHUNK_LARGE
#define BRK_CHUNK_LARGE 2048
#endif
#define BRK_CHUNK_WORD (2 * NBRK + 5)
__global__ __launch_bounds__(1024) void bin_rowscan_kernel(const int32_t *__restrict__ brick_cnt,
                                                           int32_t *__restrict__ brick_start, int chunk) {
    __shared__ int part[1024];
    __shared__ int partw[1024];
    constexpr int PER = (NBRK + 1 + 1023) / 1024;  // 5
    const int t = threadIdx.x;
    int32_t *work_start = brick_start + NBRK + 2;
    int loc[PER], locw[PER], s = 0, sw = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int i = t * PER + k;
        loc[k] = i <= NBRK ? brick_cnt[i] : 0;
        locw[k] = i < NBRK ? (loc[k] + chunk - 1) / chunk : 0;  // the out-of-box bucket does no work
        s += loc[k];
        sw += locw[k];
    }
    part[t] = s;
    partw[t] = sw;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = t >= o ? part[t - o] : 0;
        const int vw = t >= o ? partw[t - o] : 0;
        __syncthreads();
        part[t] += v;
        partw[t] += vw;
        __syncthreads();
    }
    int run = part[t] - s, runw = partw[t] - sw;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int i = t * PER + k;
        if (i <= NBRK) {
            brick_start[i] = run;
            work_start[i] = runw;
        }
        run += loc[k];
        runw += locw[k];
    }
    if (t == 1023) brick_start[NBRK + 1] = part[1023], brick_start[BRK_CHUNK_WORD] = chunk;
}

__global__ __launch_bounds__(256) void bin_scatter_kernel(const float *__restrict__ x, int64_t M, int64_t chunk, float bound,
                                                          float two_bound, const int32_t *__restrict__ block_off,
                                                          const int32_t *__restrict__ brick_start, int32_t *__restrict__ perm) {
    __shared__ int cur[NBRK + 1];
    for (int i = threadIdx.x; i <= NBRK; i += 256) cur[i] = brick_start[i] + block_off[(int64_t)blockIdx.x * (NBRK + 1) + i];
    __syncthreads();
    const int64_t p0 = (int64_t)blockIdx.x * chunk, p1 = min(p0 + chunk, M);
    for (int64_t p = p0 + threadIdx.x; p < p1; p += 256) {
        const int slot = atomicAdd(&cur[brick_of(x, p, bound, two_bound)], 1);
        perm[slot] = (int32_t)p;
    }
}

// max |grad| over the whole feature-gradient tensor, as the raw bits of a non-negative float
// (monotone in the value, so an integer atomicMax does the reduction)
__global__ __launch_bounds__(256) void absmax_kernel(const float *__restrict__ g, int64_t n, uint32_t *__restrict__ out) {
    uint32_t m = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t n4 = n >> 2;   // n = M * 32 floats: whole float4s (the tail loop below covers any other caller)
    const f32x4 *g4 = reinterpret_cast<const f32x4 *>(g);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const f32x4 v = g4[i];
        m = max(max(m, __float_as_uint(fabsf(v[0]))), __float_as_uint(fabsf(v[1])));
        m = max(max(m, __float_as_uint(fabsf(v[2]))), __float_as_uint(fabsf(v[3])));
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) m = max(m, __float_as_uint(fabsf(g[i])));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// LDS float atomics (ds_add_f32) run ~40x slower than integer ones on gfx950 (177 vs 4.6-6.4 cycles per
// wave instruction, tools/micro/lds_atomics.hip), so vertex gradients are accumulated on-chip in 64-bit
// fixed point: q = (int64) (v * 2^40 / G),  G = power of two >= max|grad|.  A vertex receives at most 2048
// terms of magnitude <= 2^40 per work item (sum < 2^52 in an int64), every fp32 term is represented to 2^-40 G (i.e.
// exactly, for all practical purposes), and integer addition commutes: the on-chip stage is exact and
// order-independent -- more accurate than the float atomics it replaces, and ~25x faster.  (ds_add_f64 is also
// fast, 8.1 cycles, but its same-address rate is 40 vs 26 cycles and the kernel came out 25 % slower with it, even
// without the max|grad| pre-pass it makes unnecessary: measured, not kept.)
#define FX_BITS 40
#define FX_MAGIC 6755399441055744.0                 // 1.5 * 2^52
#define FX_MAGIC_BITS 0x4338000000000000ULL
#define FX_LIMIT 1099511627776.0f                   // 2^40: the fixed-point range a scaled gradient is clamped to
__device__ __forceinline__ void fx_scales(uint32_t maxbits, float &to_fx, float &from_fx) {
    int e = (int)(maxbits >> 23) + 1;   // biased exponent of the power of two above max|grad|
    e = max(e, 60);                     // gradients below 2^-67 are accumulated with a fixed (coarser) scale
    e = min(e, 254);
    to_fx = __uint_as_float((uint32_t)(127 + FX_BITS + 127 - e) << 23);    // 2^(40 - (e-127))
    from_fx = __uint_as_float((uint32_t)(127 - FX_BITS - 127 + e) << 23);  // 2^((e-127) - 40)
}

// Workgroup = BRK_THREADS = 1024 lanes: 64 points x 16 levels in flight per iteration, 16 waves sharing one accumulator.
//
// Round-5 timing experiments on the d/dx form (2 tables at cfg3, same box; the wrong-result variants live in
// tools/micro/hashgrid_brk_exp.patch, applied to a copy by tools/build_grid_variants.sh exp:NAME -- never in this file):
//   as shipped in round 4 1.38 ms | VALU stream cut by 40 % (DPP sums, branch-free rows, contracted d/dx) 1.32 | LDS atomics
//   removed 1.21 | the eight table-row gathers of d/dx sent to ONE row 0.86 | both 0.76.
// Neither the instruction stream nor the LDS atomic pipe nor load latency (operands requested an iteration ahead: no gain) was
// the bound: a third of the kernel was the texture-address path working through 8 gathers x 64 distinct cache lines per wave
// and iteration -- the same 2.1 GB of L2 gathers per table the forward does.  But a brick's points only ever read the brick's own
// vertices, the very ones the accumulator has a slot for: the d/dx forms now STAGE the brick's table rows in LDS once per work
// item (4558 rows, against 1024 points x 16 levels x 8 corners = 131 k gathers), one 24-byte slot per vertex {sum x, sum y, row},
// and the loop reads a corner's row at the address it adds the corner's gradient to.  109 KB per workgroup: ONE workgroup per
// CU (4 waves per SIMD, 128 registers), which pays for requesting the next point's operands an iteration ahead.
// (DESIGN.md section 3 "The hash grid" has the whole table; profiles/r05_ab_hashgrid.txt the final kernels' run of it.)
#define BRK_THREADS 1024
#ifndef BRK_PIPE
#define BRK_PIPE 2             // A/B: 0 = every operand requested in the iteration that uses it, 1 = the index one iteration ahead, 2 = index two ahead, x / grad one
#endif
#ifndef BRK_STAGE
#define BRK_STAGE 1            // A/B: 0 = the d/dx forms gather their table rows from global memory (two workgroups per CU)
#endif
#ifndef BRK_STAGE_UNROLL
#define BRK_STAGE_UNROLL 8     // staging gathers in flight per lane
#endif
#ifndef BRK_STAGE_MIN_POINTS
#define BRK_STAGE_MIN_POINTS (1 << 20)
#endif
#ifndef BRK_STAGE_MIN
#define BRK_STAGE_MIN 96       // a work item with fewer points gathers its rows directly (staging = 4558 gathers, ~36 points' worth, at half the occupancy)
#endif

// j / d for 0 <= j < 4608, 1 <= d <= 16 (a level's vertex block fits BRK_NODES_MAX: <= 16 vertices per axis): one multiply and a
// shift, m = ceil(2^16 / d) -- checked exhaustively over that range
__device__ __forceinline__ int brk_div(int j, int m) { return (int)(((uint32_t)j * (uint32_t)m) >> 16); }

// work item w -> (brick, first point, end) in item[0..2]; false for a surplus workgroup (w >= work_start[NBRK]).
// The brick b with work_start[b] <= w < work_start[b + 1] is unique (the table is monotone).  A binary search is twelve DEPENDENT
// loads before the workgroup can start; the 1024 lanes look at four bricks each instead, and the brick's point range comes with
// the same round trip.  Ends in a barrier: item[] is visible to every lane on return.
__device__ __forceinline__ bool brk_find_item(const int32_t *__restrict__ brick_start, int w, int *item) {
    const int32_t *work_start = brick_start + NBRK + 2;
    constexpr int PER = NBRK / BRK_THREADS;
    const int b0 = PER * (int)threadIdx.x;
    int ws[PER + 1], bs[PER + 1];
    const int chunk = brick_start[BRK_CHUNK_WORD];
#pragma unroll
    for (int k = 0; k <= PER; k++) ws[k] = work_start[b0 + k], bs[k] = brick_start[b0 + k];
    bool mine = false;
#pragma unroll
    for (int k = 0; k < PER; k++)
        if (ws[k] <= w && w < ws[k + 1]) {
            const int first = bs[k] + (w - ws[k]) * chunk;
            item[0] = b0 + k, item[1] = first, item[2] = min(first + chunk, bs[k + 1]), mine = true;
        }
    return __syncthreads_or(mine) != 0;
}

// this lane's level of the brick: geometry of the level, of the brick's vertex block inside it, and the row-index terms
struct BrickLevel {
    uint32_t res, T, my, mz, mask, sel;     // row = dense ? cx + cy my + cz mz : (cx ^ cy my ^ cz mz) & mask   (sel = dense mask)
    bool dense, pow2;
    int nn, base, lo[3];
    const float2 *tab;
};

__device__ __forceinline__ BrickLevel brk_level(const GridMeta &meta, const BrickMeta &bm, const float2 *__restrict__ emb, int l,
                                                int brick) {
    BrickLevel v;
    v.res = (uint32_t)meta.res[l];
    v.T = (uint32_t)(meta.offsets[l + 1] - meta.offsets[l]);
    v.dense = (uint64_t)v.res * v.res * v.res <= (uint64_t)v.T;
    v.pow2 = (v.T & (v.T - 1)) == 0;
    v.my = v.dense ? v.res : 2654435761u, v.mz = v.dense ? v.res * v.res : 805459861u;
    v.mask = v.pow2 ? v.T - 1 : 0xffffffffu, v.sel = v.dense ? 0xffffffffu : 0u;
    v.nn = bm.n[l], v.base = bm.lds_off[l];
    const int bxyz[3] = {brick % BRK, (brick / BRK) % BRK, brick / (BRK * BRK)};
#pragma unroll
    for (int d = 0; d < 3; d++) {
        // smallest cell index a point of this brick can have: same fmaf as grid_locate, monotone in u
        const float pos = fminf(fmaxf(fmaf((float)bxyz[d] / (float)BRK, (float)v.res, -0.5f), 0.0f), (float)(v.res - 1));
        v.lo[d] = (int)floorf(pos);
    }
    v.tab = emb + meta.offsets[l];
    return v;
}

// Stage the table rows of the lane's level of the brick in LDS: vertex j of the level's block goes to the 8-byte word
// lds8[STRIDE * (base + j) + OFF].  The lane's level, vertices sub, sub + 64, ...: BRK_STAGE_UNROLL independent gathers in flight.
// A slot past the grid's last vertex (the brick at the upper border) is never read by a point; it gets the border row.
// (Branch-free row index as in grid_rows8: the 16 level-lanes of a wave mix dense and hashed levels.)
template <int STRIDE, int OFF>
__device__ __forceinline__ void brk_stage_rows(const BrickLevel &v, int sub, long long *lds8) {
    const int n3 = v.nn * v.nn * v.nn, m = (65536 + v.nn - 1) / v.nn;
    for (int j0 = sub; j0 < n3; j0 += 64 * BRK_STAGE_UNROLL) {
        float2 r[BRK_STAGE_UNROLL];
#pragma unroll
        for (int u = 0; u < BRK_STAGE_UNROLL; u++) {
            const int j = min(j0 + 64 * u, n3 - 1);
            const int t = brk_div(j, m), jz = brk_div(t, m);
            const int jx = j - t * v.nn, jy = t - jz * v.nn;
            const uint32_t cx = min((uint32_t)(v.lo[0] + jx), v.res - 1), ym = min((uint32_t)(v.lo[1] + jy), v.res - 1) * v.my,
                           zm = min((uint32_t)(v.lo[2] + jz), v.res - 1) * v.mz;
            uint32_t row = ((cx + ym + zm) & v.sel) | ((cx ^ ym ^ zm) & v.mask & ~v.sel);
            if (__builtin_expect(!v.dense && !v.pow2, 0)) row %= v.T;
            r[u] = v.tab[row];
        }
#pragma unroll
        for (int u = 0; u < BRK_STAGE_UNROLL; u++) {
            const int j = j0 + 64 * u;
            if (j < n3) *reinterpret_cast<float2 *>(&lds8[STRIDE * (v.base + j) + OFF]) = r[u];
        }
    }
}

// Brick-binned FORWARD.  The ungrouped forward is the texture-address path working through 8 gathers x 64 distinct cache lines
// per wave (0.46 ms per table at 2.1 M points: one line per clock and CU) -- and the training step bins its points into bricks
// for the backward anyway.  Binned BEFORE the forward, a work item stages its brick's 4558 rows in LDS (37 KB: two 1024-lane
// workgroups per CU) and its <= 1024 points read their corners there.  Per-point arithmetic is grid_fwd_kernel's: bit-identical
// features.  Points outside the box sit behind the last brick in `perm`; the surplus workgroups of the launch (there are at
// least ceil(#outside / 1024) of them) write their zero rows.
__global__ __launch_bounds__(BRK_THREADS, 8) void grid_fwd_brick_kernel(const float *__restrict__ x, const float2 *__restrict__ emb,
                                                                         GridMeta meta, BrickMeta bm,
                                                                         const int32_t *__restrict__ perm,
                                                                         const int32_t *__restrict__ brick_start,
                                                                         float2 *__restrict__ out, int L, int n_levels, float bound,
                                                                         float two_bound) {
    __shared__ long long val[BRK_NODES_MAX];    // float2 rows
    __shared__ int item[3];
    const int l = threadIdx.x & 15, sub = threadIdx.x >> 4;
    constexpr int PPI = BRK_THREADS / 16;
    if (!brk_find_item(brick_start, blockIdx.x, item)) {
        const int s = (int)blockIdx.x - brick_start[NBRK + 2 + NBRK];                  // surplus workgroup number
        const int first = brick_start[NBRK] + s * BRK_CHUNK_SMALL, stop = min(first + BRK_CHUNK_SMALL, brick_start[NBRK + 1]);
        for (int i = first + sub; i < stop; i += PPI) out[(int64_t)perm[i] * L + l] = make_float2(0.f, 0.f);
        return;
    }
    const int brick = item[0], start = item[1], end = item[2];
    const bool lev_on = l < n_levels;
    const BrickLevel lv = brk_level(meta, bm, emb, l, brick);
    if (lev_on) brk_stage_rows<1, 0>(lv, sub, val);
    __syncthreads();
    const int last = end - 1;
    // operands of the next point requested an iteration ahead, its index two ahead (as in the backward)
    int p_cur = perm[min(start + sub, last)], p_nxt = perm[min(start + sub + PPI, last)];
    float xv[3] = {x[(int64_t)p_cur * 3 + 0], x[(int64_t)p_cur * 3 + 1], x[(int64_t)p_cur * 3 + 2]};
    for (int i = start + sub; i < end; i += PPI) {
        const int p_nn = perm[min(i + 2 * PPI, last)];
        const float xn[3] = {x[(int64_t)p_nxt * 3 + 0], x[(int64_t)p_nxt * 3 + 1], x[(int64_t)p_nxt * 3 + 2]};
        float2 r = make_float2(0.f, 0.f);
        if (lev_on) {
            uint32_t g[3];
            float f[3];
            grid_locate(xv, bound, two_bound, lv.res, g, f);     // inside the box: the binning put the point into a brick
            const uint32_t g1[3] = {min(g[0] + 1, lv.res - 1), min(g[1] + 1, lv.res - 1), min(g[2] + 1, lv.res - 1)};
            const int lx[2] = {(int)g[0] - lv.lo[0], (int)g1[0] - lv.lo[0]}, ly[2] = {(int)g[1] - lv.lo[1], (int)g1[1] - lv.lo[1]},
                      lz[2] = {(int)g[2] - lv.lo[2], (int)g1[2] - lv.lo[2]};
            float2 v[8];
#pragma unroll
            for (int c = 0; c < 8; c++)
                v[c] = *reinterpret_cast<const float2 *>(
                    &val[lv.base + lx[c & 1] + __mul24(lv.nn, ly[(c >> 1) & 1] + __mul24(lv.nn, lz[(c >> 2) & 1]))]);
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const float w = ((c & 1) ? f[0] : 1.f - f[0]) * ((c & 2) ? f[1] : 1.f - f[1]) * ((c & 4) ? f[2] : 1.f - f[2]);
                r.x = fmaf(w, v[c].x, r.x);
                r.y = fmaf(w, v[c].y, r.y);
            }
        }
        out[(int64_t)p_cur * L + l] = r;
        p_cur = p_nxt, p_nxt = p_nn;
        xv[0] = xn[0], xv[1] = xn[1], xv[2] = xn[2];
    }
}

// NEED_DX 0: no d/dx; 1: grad_x = d/dx (pre-zeroed by the host side); 2: grad_x += d/dx.
// STAGED (d/dx forms): rows staged in LDS, one workgroup per CU.  The launcher takes it for calls of >= BRK_STAGE_MIN_POINTS
// points: the eight ~0.1 ms calls of a real-view training step (139 k - 830 k points, work items of a few hundred points whose
// zero / stage / flush phases nothing overlaps at one workgroup per CU) are faster in the direct form, 0.75 against 0.81 ms per
// step; cfg3's two calls of 2.1 M points 1.16 against 1.32 ms.  Inside the staged form a work item of fewer than BRK_STAGE_MIN
// points still gathers its rows directly (staging is 4558 gathers, 36 points' worth).  (Two launches over the same work-item
// table -- staged items in one, the others at two workgroups per CU in the second -- measured slower than either.)
template <int NEED_DX, bool STAGED>
__global__ __launch_bounds__(BRK_THREADS, STAGED ? 4 : 8) void grid_bwd_brick_kernel(
    const float2 *__restrict__ grad, const float *__restrict__ x, const float2 *__restrict__ emb, GridMeta meta, BrickMeta bm,
    const int32_t *__restrict__ perm, const int32_t *__restrict__ brick_start, float *__restrict__ grad_emb,
    float *__restrict__ grad_x, int L, int n_levels, float bound, float two_bound, const uint32_t *__restrict__ gmax_bits) {
    // per vertex: fixed-point sum of the x channel, of the y channel, and (d/dx forms) the vertex's table row -- 8-byte words.
    // (Interleaved or channel-major made no difference in a round-5 A/B.  The bank-conflict replays the counters show are NOT
    //  same-vertex collisions of the points a wave carries -- SQ_LDS_ADDR_CONFLICT is ~0, and walking the corners in a different
    //  order per point measured the same time (comment at the atomics below): they are the 64-bit atomics of the 16 levels' regions
    //  landing on random banks.)
    constexpr int W = STAGED ? 3 : 2;
    __shared__ long long acc[W * BRK_NODES_MAX];
    // work item -> (brick, chunk of <= 1024 / 2048 points): hot bricks (all rays converge near the camera)
    // are split over several workgroups, each with its own LDS accumulation and flush
    __shared__ int item[3];
    if (!brk_find_item(brick_start, blockIdx.x, item)) return;
    const int brick = item[0], start = item[1], end = item[2];
    const bool staged = STAGED && end - start >= BRK_STAGE_MIN;     // uniform over the workgroup
    for (int i = threadIdx.x; i < BRK_NODES_MAX; i += BRK_THREADS) acc[W * i] = 0, acc[W * i + 1] = 0;
    float to_fx, from_fx;
    fx_scales(*gmax_bits, to_fx, from_fx);
    const int l = threadIdx.x & 15, sub = threadIdx.x >> 4;
    // this lane's level
    const bool lev_on = l < n_levels;
    const BrickLevel lv = brk_level(meta, bm, emb, l, brick);
    const uint32_t res = lv.res, T = lv.T;
    const bool dense = lv.dense, pow2 = lv.pow2;
    const int nn = lv.nn, base = lv.base;
    const int lo[3] = {lv.lo[0], lv.lo[1], lv.lo[2]};
    const float2 *tab = lv.tab;
    if (staged && lev_on) brk_stage_rows<W, 2>(lv, sub, acc);
    // per-level parameters of the flush (below), from the sixteen lanes that hold one level each: the flush walks the brick's 4558
    // vertex slots in ONE flat loop, every lane looking its slot's level up here
    static_assert(MH_MAX_LEVELS >= 16, "the brick kernels run on 16-level grids (L == 16 is checked by the launchers)");
    __shared__ int flush_par[16][16];
    if (threadIdx.x < 16) {
        int *fp = flush_par[l];
        fp[0] = nn, fp[1] = (65536 + nn - 1) / nn, fp[2] = base, fp[3] = lo[0], fp[4] = lo[1], fp[5] = lo[2];
        fp[6] = (int)(res - 1), fp[7] = (int)lv.my, fp[8] = (int)lv.mz, fp[9] = (int)lv.mask, fp[10] = (int)lv.sel, fp[11] = (int)T;
        fp[12] = (int)(!dense && !pow2), fp[13] = meta.offsets[l];
    }
    __syncthreads();
    constexpr int PPI = BRK_THREADS / 16;  // points per iteration
    const int end_r = start + ((end - start + PPI - 1) / PPI) * PPI;
    // (1.5 * 2^52 held in a scalar register pair: as a literal it occupies two vector registers of every lane)
    int magic_hi, magic_lo;
    asm("s_mov_b32 %0, 0x43380000" : "=s"(magic_hi));     // FX_MAGIC_BITS >> 32 (opaque to constant folding, which would put it
    asm("s_mov_b32 %0, 0" : "=s"(magic_lo));              //  back into vector registers)
    const double magic = __longlong_as_double(((long long)magic_hi << 32) | (unsigned)magic_lo);
    // Software pipeline: an iteration works on operands requested one iteration earlier (x, grad of the next point) and on an
    // index requested two iterations earlier; loads past the chunk's end re-read its last point (a valid address, no branch
    // around a load).  (Two points per lane and iteration, for more independent LDS work per wave, measured slower: 1.19
    // against 1.16 ms -- and needs the branch around a point's work gone, which costs the sparse work items of the training
    // steps their skipped waves: 0.75 -> 0.91 ms.)
    const int last = end - 1;
    // (the direct-gather d/dx form has no registers for it at two workgroups per CU: 10 spills, and measured no faster)
    constexpr int PIPE = (NEED_DX && !STAGED) ? 0 : BRK_PIPE;
    int p_cur = 0, p_nxt = 0;
    float xv[3] = {0.f, 0.f, 0.f};
    float2 gr = make_float2(0.f, 0.f);
    if (PIPE >= 1) p_cur = perm[min(start + sub, last)];
    if (PIPE == 2) {
        p_nxt = perm[min(start + sub + PPI, last)];
#pragma unroll
        for (int d = 0; d < 3; d++) xv[d] = x[(int64_t)p_cur * 3 + d];
        gr = grad[(int64_t)p_cur * L + l];
    }
    const float inv_2b = 1.0f / two_bound;
    for (int i = start + sub; i < end_r; i += PPI) {
        const bool live = i < end;
        if (PIPE == 0) p_cur = perm[min(i, last)];
        const int64_t p = p_cur;
        int p_nn = 0;
        float xn[3] = {0.f, 0.f, 0.f};
        float2 grn = make_float2(0.f, 0.f);
        if (PIPE == 2) {
            p_nn = perm[min(i + 2 * PPI, last)];
#pragma unroll
            for (int d = 0; d < 3; d++) xn[d] = x[(int64_t)p_nxt * 3 + d];
            grn = grad[(int64_t)p_nxt * L + l];
        } else {
            if (PIPE == 1) p_nxt = perm[min(i + PPI, last)];
#pragma unroll
            for (int d = 0; d < 3; d++) xv[d] = x[p * 3 + d];
            gr = grad[p * L + l];
        }
        float gx_old[3] = {0.f, 0.f, 0.f};
        if (NEED_DX == 2 && l == 0) {       // the gradient this point already holds: requested now, added at the iteration's end
#pragma unroll
            for (int d = 0; d < 3; d++) gx_old[d] = grad_x[p * 3 + d];
        }
        float dx[3] = {0.f, 0.f, 0.f};
        if (live && lev_on) {
            uint32_t g[3];
            float f[3];
            grid_locate(xv, bound, two_bound, res, g, f);
            const uint32_t g1[3] = {min(g[0] + 1, res - 1), min(g[1] + 1, res - 1), min(g[2] + 1, res - 1)};
            const int lx0 = (int)g[0] - lo[0], ly0 = (int)g[1] - lo[1], lz0 = (int)g[2] - lo[2];
            const int lx1 = (int)g1[0] - lo[0], ly1 = (int)g1[1] - lo[1], lz1 = (int)g1[2] - lo[2];
            // fixed point of w * g: a float -> int64 conversion is ~11 VALU instructions and there are 16 per (point, level) -- a
            // quarter of the loop.  In double, w * (g 2^k) + 1.5 * 2^52 has the integer (|q| <= 2^40 < 2^51, round to nearest of
            // the EXACT 48-bit product) in its low bits: q = bits - bits(1.5 * 2^52), one v_fma_f64 and one 32-bit add.
            // (the magic add is exact only below 2^51: a NaN / Inf gradient or a stale maximum word must saturate to ONE bounded
            //  contribution, not write arbitrary bit patterns into the brick's accumulators -- one v_med3_f32 per value, outside the
            //  corner loop; v_med3 returns the smallest operand when one is a NaN)
            const double gxd = (double)__builtin_amdgcn_fmed3f(gr.x * to_fx, -FX_LIMIT, FX_LIMIT),
                         gyd = (double)__builtin_amdgcn_fmed3f(gr.y * to_fx, -FX_LIMIT, FX_LIMIT);
            // (Same-vertex collisions of the four points a wave carries are NOT what the atomics cost: with row k of the wave
            //  walking the corners in the order c ^ k -- four different vertices per instruction -- the kernel ran the same
            //  1.16 ms; without the atomics 0.77 ms.  It is the 64-bit atomic rate itself.)
            int slot[8];
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const float w = ((c & 1) ? f[0] : 1.f - f[0]) * ((c & 2) ? f[1] : 1.f - f[1]) * ((c & 4) ? f[2] : 1.f - f[2]);
                // (24-bit multiplies: a brick has < 2^12 vertices per level, v_mul_lo_u32 is a quarter-rate instruction)
                const int li = ((c & 1) ? lx1 : lx0) + __mul24(nn, ((c & 2) ? ly1 : ly0) + __mul24(nn, (c & 4) ? lz1 : lz0));
                slot[c] = W * (base + li);
                const double wd = (double)w;
                const unsigned long long qx = (unsigned long long)__double_as_longlong(__builtin_fma(wd, gxd, magic)) - FX_MAGIC_BITS;
                const unsigned long long qy = (unsigned long long)__double_as_longlong(__builtin_fma(wd, gyd, magic)) - FX_MAGIC_BITS;
                atomicAdd(reinterpret_cast<unsigned long long *>(&acc[slot[c]]), qx);
                atomicAdd(reinterpret_cast<unsigned long long *>(&acc[slot[c] + 1]), qy);
            }
            if (NEED_DX) {
                float2 v[8];
                if (staged) {
#pragma unroll
                    for (int c = 0; c < 8; c++) v[c] = *reinterpret_cast<const float2 *>(&acc[slot[c] + 2]);
                } else {
                    // (requesting the rows before the sixteen LDS atomics, to cover their round trip, needs 16 more live registers
                    //  than the 64 that keep two workgroups on a CU: 18 spills -- the rows are asked for here)
                    uint32_t row[8];
                    grid_rows8(g, g1, res, T, dense, pow2, row);
#pragma unroll
                    for (int c = 0; c < 8; c++) v[c] = tab[row[c]];
                }
                grid_ddx(v, f, gr, (float)res, dx);
            }
        }
        if (NEED_DX) {
            const float sx = sum16(dx[0]), sy = sum16(dx[1]), sz = sum16(dx[2]);
            if (live && l == 0) {
                grad_x[p * 3 + 0] = gx_old[0] + sx * inv_2b;       // NEED_DX == 1: gx_old = 0 (grad_x was zero-filled: a point
                grad_x[p * 3 + 1] = gx_old[1] + sy * inv_2b;       // outside the box is never visited)
                grad_x[p * 3 + 2] = gx_old[2] + sz * inv_2b;
            }
        }
        if (PIPE == 2) {
            p_cur = p_nxt, p_nxt = p_nn;
            xv[0] = xn[0], xv[1] = xn[1], xv[2] = xn[2], gr = grn;
        } else if (PIPE == 1) {
            p_cur = p_nxt;
        }
    }
    __syncthreads();
    // flush touched vertices: one global atomic per (vertex, channel) instead of one per (point, corner, channel).
    // One flat loop over the brick's vertex slots (five rounds of the 1024 lanes) instead of one loop per level: eleven of the sixteen
    // levels have fewer than 256 slots, their rounds ran with three quarters of the lanes idle -- and the flush is instruction time, not
    // atomic throughput (round 6, same box, cfg3's two launches: shipped 1.08 ms, the same loop without its atomics 1.06, no flush 0.92;
    // profiles/r06_ab_grid_flush.txt).  A slot's level: the number of level starts at or below it (scalar compares); the level's
    // geometry: two 16-byte LDS reads of flush_par; j / n by multiply and shift (brk_div); the branch-free row index of grid_rows8.
    {
        const int n_last = bm.n[n_levels - 1];
        const int total = n_levels > 0 ? bm.lds_off[n_levels - 1] + n_last * n_last * n_last : 0;
        for (int j = threadIdx.x; j < total; j += BRK_THREADS) {
            const long long qx = acc[W * j], qy = acc[W * j + 1];
            if (qx == 0 && qy == 0) continue;
            int lev = 0;
#pragma unroll
            for (int k = 1; k < 16; k++) lev += (k < n_levels && j >= bm.lds_off[k]) ? 1 : 0;
            const int *fp = flush_par[lev];
            const int n = fp[0], m = fp[1], jl = j - fp[2];
            const int t = brk_div(jl, m), jz = brk_div(t, m);
            const int jx = jl - t * n, jy = t - jz * n;
            const uint32_t rmax = (uint32_t)fp[6];
            const uint32_t cx = min((uint32_t)(fp[3] + jx), rmax), ym = min((uint32_t)(fp[4] + jy), rmax) * (uint32_t)fp[7],
                           zm = min((uint32_t)(fp[5] + jz), rmax) * (uint32_t)fp[8];
            const uint32_t sel = (uint32_t)fp[10];
            uint32_t row = ((cx + ym + zm) & sel) | ((cx ^ ym ^ zm) & (uint32_t)fp[9] & ~sel);
            if (__builtin_expect(fp[12] != 0, 0)) row %= (uint32_t)fp[11];
            float *ge = grad_emb + ((size_t)fp[13] + row) * 2;
            atomicAdd(ge + 0, (float)qx * from_fx);
            atomicAdd(ge + 1, (float)qy * from_fx);
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
                                  float bound, int32_t group, void *stream) {
    if (M == 0) return MH_OK;
    if (!x || !emb || !out || M < 0 || n_levels < 0 || n_levels > L || !(bound > 0.f) || group < 1) return MH_ERR_ARG;
    GridMeta meta;
    int st = fill_meta(meta, offsets_host, res_host, L);
    if (st) return st;
    if (group == 6 && M % 6 == 0) {   // the finite-difference tap layout; any other hint runs ungrouped (same results)
        const int64_t n_groups = M / 6;
        const int64_t gblocks = (n_groups * L + 255) / 256;
        if (gblocks > 0x7fffffffLL) return MH_ERR_ARG;
        hipLaunchKernelGGL(grid_fwd_grouped_kernel<6>, dim3((unsigned)gblocks), dim3(256), 0, mh_stream(stream), x,
                           reinterpret_cast<const float2 *>(emb), meta, reinterpret_cast<float2 *>(out), n_groups, (int)L,
                           (int)n_levels, bound, 2.0f * bound);
        MH_CHECK_LAUNCH();
        return MH_OK;
    }
    const int64_t threads = M * L;
    const int64_t blocks = (threads + 255) / 256;
    if (blocks > 0x7fffffffLL) return MH_ERR_ARG;
    hipLaunchKernelGGL(grid_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, mh_stream(stream), x,
                       reinterpret_cast<const float2 *>(emb), meta, reinterpret_cast<float2 *>(out), M, (int)L, (int)n_levels,
                       bound, 2.0f * bound);
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

// ---- brick binning + binned backward --------------------------------------------------------
#define BIN_BLOCKS 256
extern "C" int64_t mh_grid_bin_workspace_ints(void) { return (int64_t)BIN_BLOCKS * (NBRK + 1) + (NBRK + 1); }
extern "C" int32_t mh_grid_bin_bricks(void) { return NBRK; }
extern "C" int32_t mh_grid_bin_index_ints(void) { return 2 * NBRK + 8; }  // brick_start | work_start | scratch

// calls of at least this many points take the staged forms (grid_fwd_brick_kernel, the d/dx forms of grid_bwd_brick_kernel) and
// the larger work items; a process-wide tuning knob.  It only chooses between forms that give the same feature values and d/dx bits
// (table gradients: equal to the fixed-point resolution); an atomic, so that a thread setting it while others launch is a data race in
// no sense of the word
static std::atomic<int64_t> g_stage_min_points{BRK_STAGE_MIN_POINTS};
extern "C" int64_t mh_grid_stage_min_points(int64_t set) {
    if (set >= 0) g_stage_min_points.store(set, std::memory_order_relaxed);
    return g_stage_min_points.load(std::memory_order_relaxed);
}

extern "C" int mh_grid_bin_points(const float *x, int64_t M, float bound, int32_t *workspace, int32_t *perm,
                                  int32_t *brick_start, void *stream) {
    if (M == 0) return MH_OK;
    if (!x || !workspace || !perm || !brick_start || M < 0 || M > 0x7fffffffLL || !(bound > 0.f)) return MH_ERR_ARG;
    // histogram blocks: one per 1024 points, 8 .. BIN_BLOCKS.  The column scan walks the blocks' histograms serially (one dependent
    // L2 round trip per 8 rows: 14 us at 256 blocks whatever M is), the histogram / scatter kernels want many blocks: measured per
    // call at 22 000 / 140 000 points, 256 blocks 30 / 30 us, one per 4096 points 20 / 37 us
    int64_t G = (M + 1023) / 1024;
    G = G < 8 ? 8 : (G > BIN_BLOCKS ? BIN_BLOCKS : G);
    const int64_t chunk = (M + G - 1) / G;
    int32_t *block_hist = workspace, *brick_cnt = workspace + (int64_t)BIN_BLOCKS * (NBRK + 1);
    hipLaunchKernelGGL(bin_hist_kernel, dim3((unsigned)G), dim3(256), 0, mh_stream(stream), x, M, chunk, bound, 2.0f * bound,
                       block_hist);
    hipLaunchKernelGGL(bin_colscan_kernel, dim3((NBRK + 1 + 255) / 256), dim3(256), 0, mh_stream(stream), block_hist,
                       (int)G, brick_cnt);
    hipLaunchKernelGGL(bin_rowscan_kernel, dim3(1), dim3(1024), 0, mh_stream(stream), brick_cnt, brick_start,
                       M >= g_stage_min_points.load(std::memory_order_relaxed) ? BRK_CHUNK_LARGE : BRK_CHUNK_SMALL);
    hipLaunchKernelGGL(bin_scatter_kernel, dim3((unsigned)G), dim3(256), 0, mh_stream(stream), x, M, chunk, bound,
                       2.0f * bound, block_hist, brick_start, perm);
    MH_CHECK_LAUNCH();
    return MH_OK;
}

static int fill_brick_meta(BrickMeta &bm, const int32_t *res_host, int L) {
    int off = 0;
    for (int l = 0; l < L; l++) {
        const int n = (res_host[l] + BRK - 1) / BRK + 2;
        bm.n[l] = n;
        bm.lds_off[l] = off;
        off += n * n * n;
    }
    return off > BRK_NODES_MAX ? MH_ERR_ARG : MH_OK;
}

extern "C" int mh_grid_encode_fwd_binned(const float *x, const float *emb, const int32_t *offsets_host,
                                         const int32_t *res_host, const int32_t *perm, const int32_t *brick_start, float *out,
                                         int64_t M, int32_t L, int32_t n_levels, float bound, void *stream) {
    if (M == 0) return MH_OK;
    if (!x || !emb || !out || !perm || !brick_start || M < 0 || M > 0x7fffffffLL || n_levels < 0 || n_levels > L || L != 16 ||
        !(bound > 0.f))
        return MH_ERR_ARG;
    GridMeta meta;
    int st = fill_meta(meta, offsets_host, res_host, L);
    if (st) return st;
    BrickMeta bm;
    if ((st = fill_brick_meta(bm, res_host, L))) return st;
    // upper bound on sum_b ceil(cnt_b / chunk) + ceil(#outside / BRK_CHUNK_SMALL) for either chunk size: the surplus
    // workgroups write the zero rows
    const unsigned work_items = (unsigned)(NBRK + M / BRK_CHUNK_SMALL + 1);
    hipLaunchKernelGGL(grid_fwd_brick_kernel, dim3(work_items), dim3(BRK_THREADS), 0, mh_stream(stream), x,
                       reinterpret_cast<const float2 *>(emb), meta, bm, perm, brick_start, reinterpret_cast<float2 *>(out), (int)L,
                       (int)n_levels, bound, 2.0f * bound);
    MH_CHECK_LAUNCH();
    return MH_OK;
}

extern "C" int mh_grid_encode_bwd_binned(const float *grad, const float *x, const float *emb, const int32_t *offsets_host,
                                         const int32_t *res_host, const int32_t *perm, const int32_t *brick_start,
                                         float *grad_emb, float *grad_x, int32_t accumulate_dx, int64_t M, int32_t L,
                                         int32_t n_levels, float bound, const uint32_t *gmax_bits, void *stream) {
    if (M == 0) return MH_OK;
    if (!grad || !x || !emb || !grad_emb || !perm || !brick_start || M < 0 || n_levels < 0 || n_levels > L || L != 16 ||
        !(bound > 0.f))
        return MH_ERR_ARG;
    GridMeta meta;
    int st = fill_meta(meta, offsets_host, res_host, L);
    if (st) return st;
    BrickMeta bm;
    if ((st = fill_brick_meta(bm, res_host, L))) return st;
    // upper bound on sum_b ceil(cnt_b / chunk) for either chunk size; surplus workgroups exit at once
    const unsigned work_items = (unsigned)(NBRK + M / BRK_CHUNK_SMALL + 1);
    // max|grad| (float bits): supplied by the producer of `grad` (mh_field_bwd_data computes it on the fly), else
    // reduced here into the scratch word behind the work-item table
    const uint32_t *gmax = gmax_bits;
    if (!gmax) {
        uint32_t *own = reinterpret_cast<uint32_t *>(const_cast<int32_t *>(brick_start)) + (2 * NBRK + 4);
        if (!mh_zero_async(own, sizeof(uint32_t), mh_stream(stream))) return MH_ERR_LAUNCH;
        hipLaunchKernelGGL(absmax_kernel, dim3(1024), dim3(256), 0, mh_stream(stream), grad, M * (int64_t)L * 2, own);
        gmax = own;
    }
#define BRK_LAUNCH(DXMODE, STG)                                                                                                \
    hipLaunchKernelGGL((grid_bwd_brick_kernel<DXMODE, STG>), dim3(work_items), dim3(BRK_THREADS), 0, mh_stream(stream),       \
                       reinterpret_cast<const float2 *>(grad), x, reinterpret_cast<const float2 *>(emb), meta, bm, perm,     \
                       brick_start, grad_emb, grad_x, (int)L, (int)n_levels, bound, 2.0f * bound, gmax)
    const bool staged = BRK_STAGE && M >= g_stage_min_points.load(std::memory_order_relaxed);
    if (grad_x && accumulate_dx) {
        // grad_x already holds a gradient of the same points (the field nets' d/dx): each visited point adds to its own row
        if (staged) BRK_LAUNCH(2, true); else BRK_LAUNCH(2, false);
    } else if (grad_x) {
        // points outside the box are never visited by a brick: their d/dx is zero
        if (!mh_zero_async(grad_x, sizeof(float) * 3 * (size_t)M, mh_stream(stream))) return MH_ERR_LAUNCH;
        if (staged) BRK_LAUNCH(1, true); else BRK_LAUNCH(1, false);
    } else {
        BRK_LAUNCH(0, false);
    }
#undef BRK_LAUNCH
    MH_CHECK_LAUNCH();
    return MH_OK;
}
