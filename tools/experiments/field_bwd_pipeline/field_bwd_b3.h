// Fused field backward, bf16 x 3 form, round 6 -- included by mlp.hip (uses its RowFrag / RowSl / FusedPart / stage_fused / acc_to_lds).
//
// What changed against the round-4/5 kernels (field_fused_*_kernel<.., true>, kept for the fp32-MFMA mode only) and why: the old
// form issued ~2 000 vector instructions against 264-288 MFMAs per 32-point tile on the SIMD's one wave with no overlap at all (ISA:
// every slicing block sat between two MFMA blocks; static issue model 17.2 k cycles per tile, measured 24 k), spilled 30 registers
// and sliced every pre-activation gradient TWICE -- once in column form (lane = point) as the B operand of the backward-data
// product, once in row form (lane = feature, after a transpose through an fp32 LDS scratch) as the A operand of the weight-gradient
// product.  Here:
//   * dPre is sliced ONCE, in column form.  The three bf16 planes go to LDS as 8-byte chunks {4 consecutive rows of one point} --
//     exactly what a lane holds (accumulator registers 4q .. 4q+3) -- and come back TRANSPOSED through ds_read_b64_tr_b16: lane
//     (row, k-half) receives its row's 8 consecutive points per k16 step, the MFMA A fragment, with no VALU at all
//     (tools/micro/ds_tr_probe.hip pins the instruction's lane mapping; 36-chunk row stride = conflict-free for both directions).
//     The fp32 scratch survives only as the source of the bias gradient's row sums.
//   * all-zero k16 steps of the short last layers are not multiplied (sdf: step 3 of dP2 = [d geo | d sdf | 0...]; colour: step 1
//     of dQ2 = three rows).
//   * parked input rows (the B operand of the weight-gradient product) are loaded one layer ahead into three register slots that
//     live across the tile loop, sliced just before use, one in-tile at a time (n-major order: 24 registers of slices per in-tile).
//   * the statements of a tile are emitted by a generator (tools/gen_field_bwd.py -> field_bwd_b3_gen.h) in a software-pipelined order:
//     each MFMA followed by the ~6 instructions that ride in its shadow (tools/micro/mfma_valu_gap: 5-6 single-issue instructions per
//     v_mfma_f32_32x32x16_bf16 are free inside ONE wave), values made just before their use, loads a layer ahead; the memory operations
//     are inline asm with counted waits (below).
// Arithmetic: the same slices, the same six products per k16 step in the same order per accumulator as the old kernels -- results
// are bit-identical except where a skipped all-zero step turned a -0 into a +0.
#pragma once

#define AS3 __attribute__((address_space(3)))
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

#define FB_CHUNKS 36                                  // 8-byte chunks per rowquad row: 32 points + 4 pad
#define FB_PLANE_BYTES (8 * FB_CHUNKS * 8)            // 8 rowquads = 32 rows of one plane: 2304
#define FB_IMG_BYTES (3 * FB_PLANE_BYTES)             // hi | mid | lo: 6912
#define FB_SCR_BYTES (32 * SCR_STRIDE * 4)            // fp32 rows of one out tile: 32 x 36 floats = 4608
#define FB_GS_BYTES 256                               // d(loss)/d(sdf) of the tile's points (sdf launch): [2 halves][32]
#define FB_WAVE_BYTES (FB_SCR_BYTES + FB_IMG_BYTES + FB_GS_BYTES)

// split2 (mlp_dev.h) in two statements -- 5 and 6 instructions, what rides in ONE MFMA's shadow: hi slice + the exact residuals, then
// mid / lo of the residuals.  The same operations in the same order: the same bits.
__device__ __forceinline__ void split2a(float x0, float x1, uint32_t &hi, float &r0, float &r1) {
    union {
        bf16x2_t b;
        uint32_t u;
    } h;
    h.b = __builtin_convertvector((f32x2_t){x0, x1}, bf16x2_t);
    r0 = x0 - __uint_as_float(h.u << 16);
    r1 = x1 - __uint_as_float(h.u & 0xffff0000u);
    hi = h.u;
}
__device__ __forceinline__ void split2b(float r0, float r1, uint32_t &mid, uint32_t &lo) {
    union {
        bf16x2_t b;
        uint32_t u;
    } m, l;
    m.b = __builtin_convertvector((f32x2_t){r0, r1}, bf16x2_t);
    const float s0 = r0 - __uint_as_float(m.u << 16), s1 = r1 - __uint_as_float(m.u & 0xffff0000u);
    l.b = __builtin_convertvector((f32x2_t){s0, s1}, bf16x2_t);
    mid = m.u;
    lo = l.u;
}

// Memory operations of the tile loops, in two forms behind one set of macros (the generated code is the same):
//   FB_ASM_MEM = 0 (the product): plain C++ accesses; hipcc keeps the waitcnt book.  It also sinks a load to its first use -- the
//     row-slot loads the generator issues a layer ahead end up in front of their consumers (seen in the ISA) -- which costs time, never
//     correctness.
//   FB_ASM_MEM = 1 (experiment, tools/gpu/fbwd_ab.py): inline asm in the generator's order with counted waits (FB_WAIT_*: the
//     generator knows the issue order of a wave's LDS / vector-memory operations, each class completes in order).  MEASURED UNSAFE in
//     this kernel: hipcc treats an asm load's destination as written at the statement and, under this kernel's register pressure,
//     copies it to an AGPR or to scratch before the data has landed (tools/fbwd_asm_audit.py finds v_accvgpr_write of in-flight
//     destinations: wrong results and a memory fault on the box).  Kept for the record and for a lower-pressure future.
#ifndef FB_ASM_MEM
#define FB_ASM_MEM 0
#endif
#if FB_ASM_MEM
#define FB_DS_READ128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off) : "memory")
#define FB_DS_READTR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off) : "memory")
#define FB_DS_WRITE64(addr, val, off) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(val), "i"(off) : "memory")
#define FB_DS_WRITE32(addr, val, off) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "v"(val), "i"(off) : "memory")
// global accesses in the SGPR-base form (address = wave-uniform 64-bit base + the lane's 32-bit byte offset + immediate)
#define FB_GLOAD128(dst, voff, sbase, off) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "i"(off) : "memory")
#define FB_GLOAD32(dst, voff, sbase, off) asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "i"(off) : "memory")
// (a 128-bit store reads its data registers over several cycles: the pad keeps the next instruction from overwriting them)
#define FB_GSTORE128(voff, val, sbase, off) asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1" ::"v"(voff), "v"(val), "s"(sbase), "i"(off) : "memory")
#define FB_WAIT(cnt, ...) asm volatile("s_waitcnt " cnt : __VA_ARGS__)
#define FB_PIN(...) asm volatile("" : __VA_ARGS__)
#else
#define FB_LDS_AT(T, addr, off) (*reinterpret_cast<AS3 T *>((uintptr_t)((addr) + (off))))
#define FB_GLB_AT(T, voff, sbase, off) (*reinterpret_cast<T *>(reinterpret_cast<uintptr_t>(sbase) + (voff) + (off)))
#define FB_DS_READ128(dst, addr, off) ((dst) = FB_LDS_AT(f32x4, addr, off))
#define FB_DS_READTR(dst, addr, off) \
    ((dst) = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<AS3 s16x4_t *>((uintptr_t)((addr) + (off))))))
#define FB_DS_WRITE64(addr, val, off) (FB_LDS_AT(u32x2_t, addr, off) = (val))
#define FB_DS_WRITE32(addr, val, off) (FB_LDS_AT(float, addr, off) = (val))
#define FB_GLOAD128(dst, voff, sbase, off) ((dst) = FB_GLB_AT(const f32x4, voff, sbase, off))
#define FB_GLOAD32(dst, voff, sbase, off) ((dst) = __builtin_bit_cast(__typeof__(dst), FB_GLB_AT(const uint32_t, voff, sbase, off)))
#define FB_GSTORE128(voff, val, sbase, off) (FB_GLB_AT(f32x4, voff, sbase, off) = (val))
#define FB_WAIT(cnt, ...) do { } while (0)
#define FB_PIN(...) do { } while (0)
#endif
__device__ __forceinline__ uint32_t fb_lds_u32(const void *p) { return (uint32_t)(uintptr_t)(AS3 const char *)p; }

// per-lane LDS addresses of a wave's transposition buffers
struct FbLds {
    uint32_t scr_w, scr_r, img_w, img_r, gs_w, gs_r;      // LDS byte addresses (32-bit), the "v" address operands of the asm accesses
};
__device__ __forceinline__ FbLds fb_lds(uint32_t wb, int lane) {
    const int pt = lane & 31, h = lane >> 5, i = lane & 31;
    FbLds a;
    a.scr_w = wb + (4 * h * SCR_STRIDE + pt) * 4;                  // value r of out tile t -> row (r & 3) + 8 (r >> 2) (+ 4 h)
    a.scr_r = wb + (i * SCR_STRIDE + 16 * h) * 4;                  // row i, the 16 points of this half
    const uint32_t img = wb + FB_SCR_BYTES;
    a.img_w = img + (h * FB_CHUNKS + pt) * 8;                      // rowquad 2 q + h of the out tile, point pt
    // ds_read_b64_tr_b16: result j of lane L = element (L & 3) of the chunk addressed by lane 16 (L / 16) + 4 j + ((L % 16) >> 2).
    // Wanted in lane (row m = L & 31, k-half g = L >> 5): row m's values at points 16 g + 8 s + 4 u + j.  So lane L' addresses the
    // chunk of rowquad 4 ((L' >> 4) & 1) + (L' & 3) at point 16 (L' >> 5) + ((L' & 15) >> 2) (+ 8 s + 4 u as immediates)
    a.img_r = img + ((4 * ((lane >> 4) & 1) + (lane & 3)) * FB_CHUNKS + 16 * (lane >> 5) + ((lane & 15) >> 2)) * 8;
    a.gs_w = img + FB_IMG_BYTES + (32 * h + pt) * 4;              // d(loss)/d(sdf) of point pt (half 0 holds the values)
    a.gs_r = img + FB_IMG_BYTES + 64 * h;                          // the 16 points of this half: an LDS broadcast
    return a;
}

// parked activation row tile -> a register slot, asynchronously (the generated waits cover it)
__device__ __forceinline__ void fb_row_load_asm(RowFrag &f, const float *tile_row /* uniform */, uint32_t v_rowoff) {
    FB_GLOAD128(f.v[0], v_rowoff, tile_row, 0);
    FB_GLOAD128(f.v[1], v_rowoff, tile_row, 16);
    FB_GLOAD128(f.v[2], v_rowoff, tile_row, 32);
    FB_GLOAD128(f.v[3], v_rowoff, tile_row, 48);
}
