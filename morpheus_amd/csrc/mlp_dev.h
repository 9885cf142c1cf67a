// Device helpers shared by the fp32-MFMA kernels (mlp.hip) and the bf16x3 kernels (mlp_b3.hip): per-tile scratch geometry,
// accumulator-row arithmetic, ReLU / mask idioms, feature-major tile stores, the frequency encoding.
#pragma once
#include "common.h"

#define TILE 32
#define BLOCK_PTS 128

// ---- per-tile scratch geometry (floats) -------------------------------------------------------
// warp acts : H0 [64 rows: 2kk+h, 40 used] | deform H1..H5 [5 x 128] | topo H1..H5 [5 x 128] | ReLU sign masks
//             [net][layer][lane][2] uint32 (bit 16t+r of the lane's 64 outputs): backward-data reads 8 bytes per lane and
//             layer instead of re-loading the 256-byte activation column (the activations stay parked for the wgrad GEMM)
// warp dpre : deform dPre0..4 [5 x 128], dPre5 [32] | topo same
#define WARP_HID_ROWS (64 + 2 * 640)        // activations proper
#define WARP_ACT_ROWS (WARP_HID_ROWS + 40)  // + ReLU masks: 10 layers x 64 lanes x 2 dwords = 40 rows of 32
#define WARP_DPRE_ROWS (2 * 672)
#define WARP_NET_WPACK (5120 + 4 * 16384 + 4096)       // fwd pack floats per net
#define WARP_NET_WPACKT (4096 /*T5*/ + 4 * 16384 + 8192 /*T0: MT=2,KS=64*/)
#define WARP_NET_BIAS (4 * 128 + 32)
// field acts: S0 [96: 2kk+h, 80 used] | S1 [64] | S2 [64] | C0 [64: 2kk+h] | C1 [64] | C2 [64] | ReLU masks [4][64 lanes]
#define FIELD_HID_ROWS (96 + 64 * 5)
#define FIELD_ACT_ROWS (FIELD_HID_ROWS + 8)  // + ReLU masks of S1, S2, C1, C2: 4 x 64 lanes x 1 dword = 8 rows of 32
#define FIELD_WPACK (5120 + 4096 * 4 + 2048)
#define FIELD_WPACKT (2048 /*TC2*/ + 4096 /*TC1*/ + 4096 /*TC0*/ + 4096 /*TS2*/ + 4096 /*TS1*/ + 6144 /*TS0 MT=3*/)
#define FIELD_BIAS (64 * 5 + 32)

__device__ __forceinline__ constexpr int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

template <int MT>
__device__ __forceinline__ void acc_bias(f32x16 (&acc)[MT], const float *__restrict__ bias, int h) {
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(bias + 32 * t + 8 * r4 + 4 * h);
#pragma unroll
            for (int c = 0; c < 4; c++) acc[t][4 * r4 + c] = v[c];
        }
}

template <int MT>
__device__ __forceinline__ void acc_zero(f32x16 (&acc)[MT]) {
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
}

// max(x, 0) as ONE instruction: fmaxf() costs two (hipcc first canonicalises its operand with v_max x, x, x), and every VALU
// instruction of a layer epilogue counts (on the fp32 MFMA they are serial with the MFMAs, on the bf16 pipe they cost power).
// Integer form on the bit pattern (x <= -0.0 is a negative int -> 0; positive floats keep their bits) rather than inline asm:
// hipcc's hazard recognizer does not see an inline-asm READ of a register an in-flight MFMA is still writing (mlp_b3.hip).
__device__ __forceinline__ float relu1(float x) { return __int_as_float(max(__float_as_int(x), 0)); }

template <int MT, bool RELU>
__device__ __forceinline__ void acc_to_bin(const f32x16 (&acc)[MT], float (&bin)[16 * MT]) {
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) bin[16 * t + r] = RELU ? relu1(acc[t][r]) : acc[t][r];
}

// m = (m << 1) | (x > 0) in two VALU instructions: compare into VCC, then add-with-carry m + m + VCC (the C++ form
// compiles to compare + select + shift + or: 188 instructions per 64 values, this is 128; exact, -0.0 and NaN -> 0)
__device__ __forceinline__ uint32_t push_gt0(uint32_t m, float x) {
    asm("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(x) : "vcc");
    return m;
}

// ReLU mask of a lane's 64 post-activation values, bit 16t+r <-> bin[16t+r] > 0
__device__ __forceinline__ uint2 relu_mask64(const float (&bin)[64]) {
    uint32_t m0 = 0, m1 = 0;
#pragma unroll
    for (int j = 31; j >= 0; j--) {
        m0 = push_gt0(m0, bin[j]);
        m1 = push_gt0(m1, bin[32 + j]);
    }
    return make_uint2(m0, m1);
}

__device__ __forceinline__ uint32_t relu_mask32(const float (&bin)[32]) {
    uint32_t m = 0;
#pragma unroll
    for (int j = 31; j >= 0; j--) m = push_gt0(m, bin[j]);
    return m;
}

// apply bit `r` of a ReLU mask word to a value in two VALU instructions: sign-extend the bit to 0 / ~0 (v_bfe_i32) and
// AND it with the value's bits (the C++ ternary compiles to and + compare + select)
__device__ __forceinline__ float mask_bit(uint32_t mw, int r, float v) {
    const int m = __builtin_amdgcn_sbfe(mw, r, 1);
    return __uint_as_float(__float_as_uint(v) & (uint32_t)m);
}

// feature-major tile store: row = 32t + acc_row(r,h)
template <int MT>
__device__ __forceinline__ void store_acc_rows(float *__restrict__ tile, const float (&v)[16 * MT], int pt, int h) {
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) tile[(32 * t + acc_row(r, h)) * TILE + pt] = v[16 * t + r];
}
template <int MT>
__device__ __forceinline__ void load_acc_rows(const float *__restrict__ tile, float (&v)[16 * MT], int pt, int h) {
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) v[16 * t + r] = tile[(32 * t + acc_row(r, h)) * TILE + pt];
}
// k-step ordered store: row = 2kk + h
template <int KS>
__device__ __forceinline__ void store_kk_rows(float *__restrict__ tile, const float (&v)[KS], int pt, int h) {
#pragma unroll
    for (int k = 0; k < KS; k++) tile[(2 * k + h) * TILE + pt] = v[k];
}

// frequency encoding of one point as B operands: k-step kk<18 = (band kk/3, dim kk%3): sin on
// lanes 0-31, cos on lanes 32-63; kk 18 = (x0 | x1); kk 19 = (x2 | 0).   encodings.py:35-57
// The two lanes of a point need the SAME 18 sincosf (one keeps the sines, the other the cosines), and an accurate sincosf is
// ~100 instructions: lane h evaluates the k-steps of parity h only and hands the half it does not keep to its partner (lane ^ 32)
// -- 9 sincosf + 9 cross-lane moves per lane instead of 18; the same function on the same argument, so the encoding is bit for
// bit what it was.  k-step pairs whose bands are both switched off (progressive level: n_bands < 6) are not evaluated at all.
// (Round 4: the field forward spent a third of its VALU issue here, and VALU issue is what bounds it -- DESIGN.md section 3.)
__device__ __forceinline__ void enc_bin(const float (&x)[3], int h, int n_bands, float *__restrict__ bin /*[20]*/) {
#pragma unroll
    for (int m = 0; m < 9; m++) {
        const int k0 = 2 * m, k1 = 2 * m + 1;                       // lane h = 0 evaluates k0, lane h = 1 evaluates k1
        const int b0 = k0 / 3, d0 = k0 % 3, b1 = k1 / 3, d1 = k1 % 3;
        if (b0 < n_bands) {                                         // wave-uniform (b0 <= b1)
            const float arg = h ? x[d1] * (float)(1 << b1) : x[d0] * (float)(1 << b0);
            float s, c;
            sincosf(arg, &s, &c);
            const float got = __shfl_xor(h ? s : c, 32);           // lane 0 receives sin(k1), lane 1 receives cos(k0)
            const bool on1 = b1 < n_bands;
            bin[k0] = h ? got : s;
            bin[k1] = on1 ? (h ? c : got) : 0.f;
        } else {
            bin[k0] = 0.f;
            bin[k1] = 0.f;
        }
    }
    bin[18] = h ? x[1] : x[0];
    bin[19] = h ? 0.f : x[2];
}

// d(encoding feature)/dx of enc_bin's 18 sin/cos k-steps from the PARKED encoding instead of 18 more sincosf calls
// (~2 200 VALU instructions per tile, serial with the fp32 MFMAs): the forward kernels park the encoding k-step-major
// (row 2k + h = half h's feature of k-step k: sin for h = 0, cos for h = 1; zero for switched-off bands), and
// d sin(f x)/dx = f cos(f x), d cos(f x)/dx = -f sin(f x) are the OTHER half's parked value times +-f.
__device__ __forceinline__ void enc_deriv_parked(const float *__restrict__ enc_tile, int pt, int h, float *__restrict__ dsc /*[18]*/) {
    // keep the 18 loads HERE: hoisted to the top of the kernel (the scheduler's default for loads) they would occupy 18
    // registers across the whole layer chain, which has none to spare
    __builtin_amdgcn_sched_barrier(0);
    const float *base = enc_tile + (1 - h) * TILE + pt;
    asm volatile("" : "+v"(base)::"memory");
#pragma unroll
    for (int k = 0; k < 18; k++) {
        const float f = (float)(1 << (k / 3));
        const float other = base[2 * k * TILE];
        dsc[k] = h ? -f * other : f * other;
    }
}

// Laplace density (models/density.py:22-31): sigma = (1/beta) (0.5 + 0.5 sign(s) expm1(-|s|/beta)), returned as beta * sigma.
// The reference's expression cancels 0.5 - 0.5 (1 - e) for points outside the surface: far from it the result (~e/2) keeps the
// absolute error of an ulp of 0.5, and the opacity of a ray that only grazes the box -- a sum of ~10^2 such terms -- carried 1.5e-4
// relative against the reference run in double with the device's expm1f (the reference's own fp32 run: 5.6e-5;
// profiles/r04_parity_f64.jsonl).  The same function without the cancellation: e = exp(-|s|/beta);  s > 0: e/2;  s < 0: 1 - e/2;
// s = 0: 1/2 -- every branch correct to an ulp of ITS value.
__device__ __forceinline__ float laplace_unit(float s, float beta) {
    const float e = expf(-fabsf(s) / beta);
    return s > 0.f ? 0.5f * e : (s < 0.f ? 1.0f - 0.5f * e : 0.5f);
}

// ---- bf16x3: exact three-way bf16 split of fp32 operands (mlp_b3.hip, the b3 weight-gradient body in mlp.hip) ------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
union Frag {
    f32x4 f;
    bf16x8 h;
    uint32_t u[4];
};

// parked tiles are written once and read much later by another kernel
#ifdef MH_B3_PLAIN_STORES
#define PARK_STORE(v, p) (*(p) = (v))
#else
#define PARK_STORE(v, p) __builtin_nontemporal_store((v), (p))
#endif

// ReLU and the sign-mask bit as compiler-visible integer instructions (one v_max_i32; v_min_u32 + v_lshl_add_u32): the float
// forms cost an extra canonicalising v_max each, inline-asm forms are invisible to the hazard recognizer (mlp_b3.hip:
// mfma_results_settle).  Bit patterns: x <= -0.0 is a negative int -> 0; positive floats and +NaN keep their bits.
__device__ __forceinline__ float relu_i(float x) { return __int_as_float(max(__float_as_int(x), 0)); }
// m = (m << 1) | (y != 0) for y >= 0 in TWO instructions: 0 - bits(y) is negative exactly when y != 0, and v_alignbit_b32
// (m : t) >> 31 shifts m up by one and brings t's sign in.  (The min(bits, 1) form compiled to compare + select + or3 + shift, 3.2
// instructions per value: a tenth of the forward kernels' VALU issue.)
__device__ __forceinline__ uint32_t push_nz(uint32_t m, float y /* >= 0 */) {
    return __builtin_amdgcn_alignbit(m, 0u - __float_as_uint(y), 31);
}

// two fp32 values -> the packed bf16 pairs of their three slices.  Round-to-nearest split (v_cvt_pk_bf16_f32): x - hi and
// (x - hi) - mid are exact in fp32 and the last residual has at most 8 significant bits, so hi + mid + lo == x exactly, like the
// truncation split -- but the residuals are half as large and of either sign: what the six-product form drops (mid.lo, lo.mid,
// lo.lo) is <= ~2^-26 of a product and unbiased, where truncation left a 3 * 2^-24 bias towards zero (measured: the bias
// gradients' error against float64 8x that of the fp32 kernels with truncation).  Same instruction count.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2(float x0, float x1, uint32_t &hi, uint32_t &mid, uint32_t &lo) {
    union {
        bf16x2_t b;
        uint32_t u;
    } h, m, l;
    h.b = __builtin_convertvector((f32x2_t){x0, x1}, bf16x2_t);
    const float r0 = x0 - __uint_as_float(h.u << 16), r1 = x1 - __uint_as_float(h.u & 0xffff0000u);
    m.b = __builtin_convertvector((f32x2_t){r0, r1}, bf16x2_t);
    const float s0 = r0 - __uint_as_float(m.u << 16), s1 = r1 - __uint_as_float(m.u & 0xffff0000u);
    l.b = __builtin_convertvector((f32x2_t){s0, s1}, bf16x2_t);
    hi = h.u;
    mid = m.u;
    lo = l.u;
}
