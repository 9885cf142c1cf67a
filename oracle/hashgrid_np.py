"""ORACLE -- TEST INFRASTRUCTURE ONLY.  A SECOND, independently written derivation of the reference's multires hash
grid (external/encoders/gridencoder/src/gridencoder.cu), vectorised numpy, used to cross-check oracle/hashgrid.c.

The reference encoder is CUDA-only, cannot be compiled in this image (nvcc + torch's CUDA headers) and ships no
tests, so nothing from the reference can pin the operator.  The remaining lever is redundancy: this
file was written from the .cu alone -- whole-array arithmetic over [points, corners], no scalar loops, nothing shared
with hashgrid.c -- so a misreading would have to be made twice, the same way, to go unnoticed.

    cell / fractional position   gridencoder.cu:148-152   pos = clamp(u*res - 0.5, 0, res-1); cell = floor(pos)
    corner enumeration + weights gridencoder.cu:170-184   bit d of the corner id selects (1-f_d, cell_d) / (f_d, min(cell_d+1, res-1))
    row index                    gridencoder.cu:61-79     running stride while stride <= T; hashed iff the stride outgrew T
    hash                         gridencoder.cu:45-58     xor of coordinate * {1, 2654435761, 805459861} in uint32
    out-of-range inputs          gridencoder.cu:105-130   any u_d outside [0,1] -> zeros
    d(out)/d(u)                  gridencoder.cu:206-246   per axis: sum over the 4 corner pairs of w * (right - left) * res
    scatter of the gradient      gridencoder.cu:253-349   grad_emb[row] += w * grad        (np.add.at)
"""
from __future__ import annotations

import numpy as np

P1, P2 = np.uint32(2654435761), np.uint32(805459861)


def _rows(cx, cy, cz, res, T):
    """Row of the level table for integer corner coordinates (uint32 arrays)."""
    res = np.uint32(res)
    # the CUDA loop multiplies the stride by `res` after each axis it accepts and stops accepting once stride > T;
    # for D = 3: axis 0 always (stride 1), axis 1 iff res <= T, axis 2 iff res^2 <= T; hashed iff the final stride > T
    s1 = np.uint64(res)
    s2 = s1 * np.uint64(res)
    s3 = s2 * np.uint64(res)
    T64 = np.uint64(T)
    with np.errstate(over="ignore"):
        lin = cx.astype(np.uint32).copy()
        final = s1
        if s1 <= T64:
            lin = lin + cy * np.uint32(s1 & np.uint64(0xFFFFFFFF))
            final = s2
            if s2 <= T64:
                lin = lin + cz * np.uint32(s2 & np.uint64(0xFFFFFFFF))
                final = s3
        if final > T64:
            idx = cx ^ (cy * P1) ^ (cz * P2)
        else:
            idx = lin
    return (idx % np.uint32(T)).astype(np.int64)


def _level(u, res, T):
    """Corner rows [M,8], weights [M,8] (float32), fractional position f [M,3], cell [M,3], in-range mask [M]."""
    resf = np.float32(res)
    # nvcc contracts `u*res - 0.5f` into one fused multiply-add (default --fmad=true): single rounding
    pos = np.minimum(np.maximum(_fma32(u, resf, np.float32(-0.5)), np.float32(0.0)), np.float32(res - 1)).astype(np.float32)
    cell = np.floor(pos).astype(np.uint32)
    f = (pos - cell.astype(np.float32)).astype(np.float32)
    bits = ((np.arange(8)[:, None] >> np.arange(3)[None, :]) & 1).astype(bool)            # [8 corners, 3 axes]
    hi = np.minimum(cell + np.uint32(1), np.uint32(res - 1))
    corner = np.where(bits[None], hi[:, None, :], cell[:, None, :])                        # [M,8,3]
    wax = np.where(bits[None], f[:, None, :], (np.float32(1.0) - f)[:, None, :]).astype(np.float32)
    w = ((np.float32(1.0) * wax[..., 0]) * wax[..., 1]) * wax[..., 2]                      # the kernel's product order
    rows = _rows(corner[..., 0], corner[..., 1], corner[..., 2], res, T)
    ok = np.all((u >= 0) & (u <= 1), axis=-1)
    return rows, w.astype(np.float32), f, cell, ok


def _fma32(a, b, c):
    a, b, c = np.asarray(a, dtype=np.float32), np.asarray(b, dtype=np.float32), np.asarray(c, dtype=np.float32)
    return _fma32_arrays(a, b, c)


def _fma32_arrays(a, b, c):
    """fmaf(a, b, c) for float32 arrays: the product of two float32 is exact in float64 and the one rounding of the sum
    to float32 follows (double rounding can differ from a true fma in the last bit in rare cases: tests allow 2 ulp)."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def forward(u, emb, offsets, res_tab, n_levels):
    """u [M,3] in [0,1] (float32), emb [rows,C] -> out [M, L*C] (levels >= n_levels zero-filled, grid.py:53)."""
    u = np.asarray(u, dtype=np.float32)
    M, L, C = u.shape[0], len(offsets) - 1, emb.shape[1]
    out = np.zeros((M, L * C), dtype=np.float32)
    for l in range(n_levels):
        T = int(offsets[l + 1] - offsets[l])
        rows, w, _, _, ok = _level(u, int(res_tab[l]), T)
        tab = emb[offsets[l]:offsets[l + 1]]
        acc = np.zeros((M, C), dtype=np.float32)
        for k in range(8):                                    # corner id order = accumulation order of the kernel
            acc = _fma32(w[:, k, None], tab[rows[:, k]], acc)
        out[:, l * C:(l + 1) * C] = np.where(ok[:, None], acc, np.float32(0.0))
    return out


def dy_du(u, emb, offsets, res_tab, n_levels):
    """[M, L, 3, C] derivative of the features w.r.t. u (the kernel's own rule: it ignores the border clamps)."""
    u = np.asarray(u, dtype=np.float32)
    M, L, C = u.shape[0], len(offsets) - 1, emb.shape[1]
    out = np.zeros((M, L, 3, C), dtype=np.float32)
    for l in range(n_levels):
        res, T = int(res_tab[l]), int(offsets[l + 1] - offsets[l])
        _, _, f, cell, ok = _level(u, res, T)
        tab = emb[offsets[l]:offsets[l + 1]]
        hi = np.minimum(cell + np.uint32(1), np.uint32(res - 1))
        for gd in range(3):
            others = [d for d in range(3) if d != gd]
            acc = np.zeros((M, C), dtype=np.float32)
            for k in range(4):
                w = np.full(M, np.float32(res), dtype=np.float32)
                c = [None, None, None]
                for nd, d in enumerate(others):
                    up = (k >> nd) & 1
                    w = (w * (f[:, d] if up else (np.float32(1.0) - f[:, d]))).astype(np.float32)
                    c[d] = hi[:, d] if up else cell[:, d]
                c[gd] = cell[:, gd]
                left = tab[_rows(c[0], c[1], c[2], res, T)]
                c[gd] = hi[:, gd]
                right = tab[_rows(c[0], c[1], c[2], res, T)]
                acc = _fma32(w[:, None], (right - left).astype(np.float32), acc)
            out[:, l, gd] = np.where(ok[:, None], acc, np.float32(0.0))
    return out


def backward_embeddings(u, grad_out, offsets, res_tab, n_levels, n_rows, C):
    """grad_emb [rows, C] (float64 accumulation: the reference's float atomics have no defined order)."""
    u = np.asarray(u, dtype=np.float32)
    g = np.zeros((n_rows, C), dtype=np.float64)
    for l in range(n_levels):
        T = int(offsets[l + 1] - offsets[l])
        rows, w, _, _, ok = _level(u, int(res_tab[l]), T)
        go = grad_out[:, l * C:(l + 1) * C].astype(np.float64) * ok[:, None]
        for k in range(8):
            np.add.at(g, offsets[l] + rows[:, k], w[:, k, None].astype(np.float64) * go)
    return g
