"""ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by morpheus_amd/).

ctypes binding + torch.autograd wrapper around oracle/hashgrid.c, the CPU
restatement of the reference's CUDA-only hash-grid encoder
(external/encoders/gridencoder/grid.py:25-169, src/gridencoder.cu:45-378).

`OracleGridEncoder` mirrors the reference module's constructor/forward surface
(grid.py:103-169) closely enough to stand in for it when the reference model is
imported on CPU by oracle/make_golden.py.
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess

import numpy as np
import torch
import torch.nn as nn

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_hashgrid.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "hashgrid.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-std=c11", "-ffp-contract=off", "-mfma", "-fopenmp", "-shared",
                               "-fPIC", "-o", _SO, src, "-lm"])
    return _SO


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        fp, ip = ctypes.c_void_p, ctypes.c_void_p
        _lib.oracle_grid_forward.argtypes = [fp, fp, ip, ip, fp, fp, ctypes.c_int64, ctypes.c_int32,
                                             ctypes.c_int32, ctypes.c_int32]
        _lib.oracle_grid_forward.restype = None
        _lib.oracle_grid_backward.argtypes = [fp, fp, ip, ip, fp, fp, fp, ctypes.c_int64, ctypes.c_int32,
                                              ctypes.c_int32, ctypes.c_int32]
        _lib.oracle_grid_backward.restype = None
    return _lib


def level_resolutions(L: int, per_level_scale: float, base: int) -> np.ndarray:
    """res_l = (uint32) ceil(exp2f(l * S) * H) evaluated in float32 (gridencoder.cu:133),
    with S = (float) log2(per_level_scale) (grid.py:39)."""
    S = np.float32(np.log2(per_level_scale))
    l = np.arange(L, dtype=np.float32)
    return np.ceil(np.exp2(l * S).astype(np.float32) * np.float32(base)).astype(np.int32)


def effective_levels(max_level, L: int) -> int:
    """grid.py:42."""
    return L if max_level is None else max(min(int(math.ceil(max_level * L)), L), 1)


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class _OracleGridEncode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, emb, offsets, res_tab, n_levels, need_dx):
        lib = _load()
        u = u.detach().contiguous().float()
        emb_c = emb.detach().contiguous()
        B, L, C = u.shape[0], offsets.shape[0] - 1, emb.shape[1]
        out = torch.empty(B, L * C)
        dydx = torch.empty(B, L * 3 * C) if need_dx else None
        lib.oracle_grid_forward(_ptr(u), _ptr(emb_c), _ptr(offsets), _ptr(res_tab), _ptr(out), _ptr(dydx),
                                B, L, C, n_levels)
        ctx.save_for_backward(u, offsets, res_tab, dydx if need_dx else torch.empty(0))
        ctx.dims = (B, L, C, n_levels, need_dx, tuple(emb.shape))
        return out

    @staticmethod
    def backward(ctx, grad):
        lib = _load()
        u, offsets, res_tab, dydx = ctx.saved_tensors
        B, L, C, n_levels, need_dx, eshape = ctx.dims
        grad = grad.contiguous().float()
        g_emb = torch.zeros(eshape)
        g_u = torch.zeros(B, 3) if need_dx else None
        lib.oracle_grid_backward(_ptr(grad), _ptr(u), _ptr(offsets), _ptr(res_tab),
                                 _ptr(dydx) if need_dx else None, _ptr(g_emb), _ptr(g_u), B, L, C, n_levels)
        return g_u, g_emb, None, None, None, None


def oracle_grid_encode(x, emb, offsets, res_tab, bound, max_level=None):
    """x in [-bound, bound]^3 -> [.., L*C]; normalisation done in torch as in grid.py:157
    so that autograd supplies the 1/(2*bound) factor."""
    L = offsets.shape[0] - 1
    u = (x + bound) / (2 * bound)
    lead = list(u.shape[:-1])
    u = u.view(-1, 3)
    out = _OracleGridEncode.apply(u, emb, offsets, res_tab, effective_levels(max_level, L), u.requires_grad)
    return out.view(lead + [L * emb.shape[1]])


class OracleGridEncoder(nn.Module):
    """Drop-in for the reference GridEncoder (grid.py:103-169) on CPU; hash gridtype, linear
    interpolation, align_corners=False only."""

    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16,
                 log2_hashmap_size=19, desired_resolution=None, gridtype="hash", align_corners=False,
                 interpolation="linear"):
        super().__init__()
        assert input_dim == 3 and gridtype == "hash" and not align_corners and interpolation == "linear"
        if desired_resolution is not None:
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
        self.input_dim, self.num_levels, self.level_dim = input_dim, num_levels, level_dim
        self.per_level_scale, self.base_resolution = per_level_scale, base_resolution
        self.output_dim = num_levels * level_dim
        offs, total = [], 0
        for i in range(num_levels):
            res = int(np.ceil(base_resolution * per_level_scale ** i))
            n = min(2 ** log2_hashmap_size, res ** input_dim)
            n = int(np.ceil(n / 8) * 8)
            offs.append(total)
            total += n
        offs.append(total)
        self.register_buffer("offsets", torch.from_numpy(np.asarray(offs, dtype=np.int32)))
        self.register_buffer("res_tab", torch.from_numpy(level_resolutions(num_levels, per_level_scale,
                                                                           base_resolution)), persistent=False)
        self.embeddings = nn.Parameter(torch.empty(total, level_dim).uniform_(-1e-4, 1e-4))

    def forward(self, inputs, bound=1, max_level=None):
        return oracle_grid_encode(inputs, self.embeddings, self.offsets, self.res_tab, bound, max_level)
