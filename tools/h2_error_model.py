"""CPU error model for a candidate "h2" arithmetic: every fp32 operand cut into TWO fp16 slices
(h = fp16(x), l = fp16(x - h)), three slice products per MAC (hh, hl, lh) on the fp16 MFMA pipe with fp32
accumulation -- half the matrix work of the bf16x3 scheme the kernels use today (six products).

Not product code and not a kernel: numpy only, run on any host.  It answers one question before anybody writes
the kernel: is the operand-representation error of h2 (22-bit worst case) visible next to the accumulation error a
plain fp32 GEMM already has, on data shaped like the warp net's layers (K = 128, post-ReLU activations, weights of
the reference's initialisation scale)?  Slice products are exact in fp32 (11 x 11 significand bits), so the error
of a scheme is its representation error plus the dropped l*l term; both are evaluated in float64 here, i.e. the
figures are a LOWER bound of what a kernel would show (they leave out the accumulator's own fp32 roundings,
which the fp32 row shows separately).

    python tools/h2_error_model.py
"""
import numpy as np


def split_bf16(x):
    def rne_bf16(v):
        u = v.astype(np.float32).view(np.uint32).astype(np.uint64)
        u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
        return u.astype(np.uint32).view(np.float32)
    hi = rne_bf16(x)
    mid = rne_bf16(x - hi)
    lo = rne_bf16(x - hi - mid)
    return hi, mid, lo


def split_f16(x, scale):
    """x * scale -> (h, l) fp16 slices (fp16 subnormals and overflow behave as in hardware), returned as float64 / scale."""
    xs = (x * np.float32(scale)).astype(np.float32)
    h = xs.astype(np.float16)
    l = (xs - h.astype(np.float32)).astype(np.float16)
    return h.astype(np.float64) / scale, l.astype(np.float64) / scale


def rel(a, ref):
    return float(np.linalg.norm(a - ref) / np.linalg.norm(ref))


def main():
    rng = np.random.default_rng(0)
    K, M, N = 128, 128, 4096
    print(f"layer {M}x{K}, {N} points; errors are ||out - out64|| / ||out64||")
    print(f"{'activation scale':>18} {'fp32 GEMM':>11} {'bf16x3 (6)':>11} {'h2 s=1':>11} {'h2 s=2^8':>11} {'h2 per-point':>13}")
    W = (rng.standard_normal((M, K)) * np.sqrt(2.0 / K)).astype(np.float32)
    for act_scale in (1e-3, 1e-2, 1e-1, 1.0, 30.0):
        X = np.maximum(rng.standard_normal((K, N)), 0).astype(np.float32) * np.float32(act_scale)
        # a heavy tail: a few features 100x larger, as positional encodings next to small codes produce
        X[:4] *= 100.0
        ref = W.astype(np.float64) @ X.astype(np.float64)
        out32 = W @ X                                                   # fp32 products and fp32 accumulation
        wh, wm, wl = (s.astype(np.float64) for s in split_bf16(W))
        xh, xm, xl = (s.astype(np.float64) for s in split_bf16(X))
        b3 = wh @ xh + wh @ xm + wm @ xh + wh @ xl + wm @ xm + wl @ xh
        cols = []
        for mode in ("one", "fixed", "point"):
            Wh, Wl = split_f16(W, 1.0 if mode == "one" else 2.0 ** 6)
            if mode == "one":
                Xh, Xl = split_f16(X, 1.0)
            elif mode == "fixed":
                Xh, Xl = split_f16(X, 2.0 ** 8)
            else:                                                       # power-of-two scale per point: amax -> 2^14
                amax = np.maximum(np.abs(X).max(axis=0), 1e-30)
                s = 2.0 ** (14 - np.ceil(np.log2(amax)))
                Xh, Xl = split_f16(X * s.astype(np.float32), 1.0)
                Xh, Xl = Xh / s, Xl / s
            cols.append(rel(Wh @ Xh + Wh @ Xl + Wl @ Xh, ref))
        print(f"{act_scale:18g} {rel(out32.astype(np.float64), ref):11.2e} {rel(b3, ref):11.2e} "
              f"{cols[0]:11.2e} {cols[1]:11.2e} {cols[2]:13.2e}")
    print("\nreading: h2 with a scale that keeps the l slices out of the fp16 subnormals sits below the fp32 GEMM's own\n"
          "accumulation error; without one (s=1) small activations lose bits, and a fixed scale overflows large ones (inf/nan).")


if __name__ == "__main__":
    main()
