"""ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by morpheus_amd/).

The multires hash grid of external/encoders/gridencoder/src/gridencoder.cu evaluated in FLOAT64 (vectorised torch, forward
only): the yardstick of the counted parity gate.  oracle/make_golden.py runs the imported reference model in double with this
encoder standing where the CUDA-only one stands, so that the fixture holds, next to the reference's own fp32 result, the value
both fp32 implementations (the reference's and the HIP path's) are rounding towards -- the gate's allowance is then derived from
how far the REFERENCE's fp32 result is from it, not fitted to the HIP path's.

Same reading of the .cu as oracle/hashgrid.c (index / hash / clamp / weights: gridencoder.cu:45-79, :132-184), restated a
third time here only in that the arithmetic type is double; integer index maths is exact in either.  Level resolutions stay
the kernel's float32 table (gridencoder.cu:133) -- they are integers.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from .hashgrid import effective_levels, level_resolutions

_P1, _P2 = 2654435761, 805459861
_M32 = (1 << 32) - 1


def grid_encode_f64(x: torch.Tensor, emb: torch.Tensor, offsets, res_tab, bound: float, max_level=None) -> torch.Tensor:
    """x [M,3] float64 in world units -> [M, L*C] float64 (levels >= the effective count are zero, grid.py:42,53)."""
    assert x.dtype == torch.float64 and emb.dtype == torch.float64
    M, C = x.shape[0], emb.shape[1]
    L = len(offsets) - 1
    n_levels = effective_levels(max_level, L)
    u = (x + bound) / (2 * bound)                                    # grid.py:157
    inside = ((u >= 0) & (u <= 1)).all(-1, keepdim=True)             # gridencoder.cu:105-130
    out = torch.zeros(M, L * C, dtype=torch.float64)
    for l in range(n_levels):
        res = int(res_tab[l])
        T = int(offsets[l + 1]) - int(offsets[l])
        pos = (u * res - 0.5).clamp(0, res - 1)                      # :148 (align_corners = False)
        g = torch.floor(pos)
        f = pos - g
        g = g.long()
        acc = torch.zeros(M, C, dtype=torch.float64)
        hashed = res ** 3 > T                                        # :61-79: the running stride outgrew the table
        for corner in range(8):
            w = torch.ones(M, dtype=torch.float64)
            c = []
            for d in range(3):
                if (corner >> d) & 1:
                    w = w * f[:, d]
                    c.append(torch.clamp(g[:, d] + 1, max=res - 1))  # :182
                else:
                    w = w * (1 - f[:, d])
                    c.append(g[:, d])
            if hashed:
                idx = ((c[0] * 1) & _M32) ^ ((c[1] * _P1) & _M32) ^ ((c[2] * _P2) & _M32)
            else:
                idx = c[0] + c[1] * res + c[2] * res * res
            row = int(offsets[l]) + idx % T
            acc = acc + w[:, None] * emb[row]
        out[:, l * C:(l + 1) * C] = acc
    return torch.where(inside, out, torch.zeros_like(out))


class OracleGridEncoderF64(nn.Module):
    """Constructor / forward surface of the reference GridEncoder (grid.py:103-169), float64, forward only."""

    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16, log2_hashmap_size=19,
                 desired_resolution=None, gridtype="hash", align_corners=False, interpolation="linear"):
        super().__init__()
        assert input_dim == 3 and gridtype == "hash" and not align_corners and interpolation == "linear"
        if desired_resolution is not None:
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
        self.input_dim, self.num_levels, self.level_dim = input_dim, num_levels, level_dim
        self.output_dim = num_levels * level_dim
        offs, total = [], 0
        for i in range(num_levels):
            res = int(np.ceil(base_resolution * per_level_scale ** i))
            n = int(np.ceil(min(2 ** log2_hashmap_size, res ** input_dim) / 8) * 8)
            offs.append(total)
            total += n
        offs.append(total)
        self.register_buffer("offsets", torch.from_numpy(np.asarray(offs, dtype=np.int32)))
        self._res = level_resolutions(num_levels, per_level_scale, base_resolution)
        self.embeddings = nn.Parameter(torch.zeros(total, level_dim, dtype=torch.float64))
        # round 5: the double run of a TRAINING step (oracle/make_golden.py:gen_round5, virt24) lets autograd through the table;
        # the forward-only yardsticks keep the detached form (identical values)
        self.differentiable = False

    def forward(self, inputs, bound=1, max_level=None):
        lead = list(inputs.shape[:-1])
        emb = self.embeddings if self.differentiable else self.embeddings.detach()
        out = grid_encode_f64(inputs.reshape(-1, 3).double(), emb.double(), self.offsets.tolist(), self._res, float(bound), max_level)
        return out.view(lead + [self.output_dim])
