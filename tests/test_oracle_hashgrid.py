"""CPU: properties of the hash-grid restatement (oracle/hashgrid.c).

The reference's encoder is CUDA-only and has no tests or golden vectors, so this operator's
parity is pinned by construction + invariants: level table, trilinear partition of
unity, exact table values at cell centres, dy_dx against finite differences, OOB -> 0, scatter
conservation, dense/hash switch, and a pure-python scalar re-derivation on a few points.
"""
import numpy as np
import torch

from morpheus_amd import synth
from oracle.hashgrid import (OracleGridEncoder, effective_levels, level_resolutions, oracle_grid_encode)


def make(scale=0.1):
    offs, s = synth.grid_offsets()
    emb = synth.hash_tensor((int(offs[-1]), 2), 9001, scale)
    return emb, torch.from_numpy(offs), torch.from_numpy(level_resolutions(16, s, 16)), s


def test_level_table():
    emb, offs, res, s = make()
    assert res.tolist() == [16, 19, 22, 25, 28, 32, 37, 43, 49, 56, 64, 74, 85, 98, 112, 128]
    rows = (offs[1:] - offs[:-1]).tolist()
    assert rows == [4096, 6864, 10648, 15632, 21952] + [32768] * 11
    assert int(offs[-1]) == 419640
    assert effective_levels(None, 16) == 16 and effective_levels(0.5, 16) == 8 and effective_levels(0.01, 16) == 1
    enc = OracleGridEncoder(num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=15, desired_resolution=128)
    assert enc.offsets.tolist() == offs.tolist() and enc.res_tab.tolist() == res.tolist()


def test_partition_of_unity_and_oob():
    emb, offs, res, _ = make()
    ones = torch.ones_like(emb)
    x = synth.hash_tensor((512, 3), 9002, 1.2)
    out = oracle_grid_encode(x, ones, offs, res, 1.01)
    inside = ((x.abs() <= 1.01).all(-1))
    assert torch.allclose(out[inside], torch.ones_like(out[inside]), atol=2e-6)
    assert (out[~inside] == 0).all() and (~inside).any()
    half = oracle_grid_encode(x, ones, offs, res, 1.01, max_level=0.5)
    assert (half[:, 16:] == 0).all() and torch.allclose(half[inside][:, :16], torch.ones(int(inside.sum()), 16), atol=2e-6)


def _py_index(T, r, c):
    stride, index = 1, 0
    for d in range(3):
        if stride > T:
            break
        index += c[d] * stride
        stride *= r
    if stride > T:
        index = ((c[0] * 1) ^ (c[1] * 2654435761) ^ (c[2] * 805459861)) & 0xFFFFFFFF
    return index % T


def test_cell_centres_and_scalar_rederivation():
    emb, offs, res, _ = make()
    bound = 1.0
    rng = np.random.RandomState(0)
    for l in (0, 4, 5, 6, 11, 15):
        r, T = int(res[l]), int(offs[l + 1] - offs[l])
        cells = rng.randint(0, r, size=(32, 3))
        u = (cells + 0.5) / r                       # cell centres: pos - 0.5 is an exact integer
        x = torch.tensor(u * 2 - 1, dtype=torch.float32)
        out = oracle_grid_encode(x, emb, offs, res, bound)
        for k in range(32):
            row = _py_index(T, r, [int(v) for v in cells[k]])
            want = emb[int(offs[l]) + row]
            assert torch.allclose(out[k, 2 * l:2 * l + 2], want, atol=2e-6), (l, k)
    # dense below, hashed above (res^3 <= T  <=>  l <= 5)
    assert all((int(res[l]) ** 3 <= int(offs[l + 1] - offs[l])) == (l <= 5) for l in range(16))


def test_dydx_matches_finite_differences_and_autograd_scatter():
    emb, offs, res, _ = make()
    emb = emb.clone().requires_grad_(True)
    # keep away from the half-cell border band, where the kernel's slope deliberately ignores the clamp
    x = (synth.hash_tensor((256, 3), 9003, 0.9)).requires_grad_(True)
    out = oracle_grid_encode(x, emb, offs, res, 1.01)
    w = synth.hash_tensor(tuple(out.shape), 9004, 1.0)
    (out * w).sum().backward()
    gx = x.grad.clone()
    eps = 1e-3
    fd = torch.zeros_like(gx)
    with torch.no_grad():
        for d in range(3):
            e = torch.zeros(1, 3)
            e[0, d] = eps
            fp = oracle_grid_encode(x + e, emb, offs, res, 1.01).double()
            fm = oracle_grid_encode(x - e, emb, offs, res, 1.01).double()
            fd[:, d] = (((fp - fm) * w.double()).sum(-1) / (2 * eps)).float()
    # piecewise-linear function: central differences straddle kinks at fine levels -> loose check
    rel = (gx - fd).abs().median() / fd.abs().median()
    assert rel < 0.05, rel
    # scatter conservation: sum of grad_emb over a level == sum_b grad[b, level] (weights sum to 1)
    ge = emb.grad
    inside = (x.detach().abs() <= 1.01).all(-1)
    for l in range(16):
        tot = ge[int(offs[l]):int(offs[l + 1])].sum(0)
        want = w[inside][:, 2 * l:2 * l + 2].sum(0)
        assert torch.allclose(tot, want, rtol=1e-4, atol=1e-4), l


def test_border_band_slope_follows_kernel():
    """Inside the half-cell border band the value is flat (pos clamped) but dy_dx is NOT zeroed
    (gridencoder.cu:205-247 ignores the clamp) -- the oracle must follow the kernel."""
    emb, offs, res, _ = make()
    x = torch.tensor([[-1.0 + 1e-3, 0.1, 0.2]], requires_grad=True)   # u ~ 5e-4 < 0.5/res for all levels
    out = oracle_grid_encode(x, emb, offs, res, 1.0)
    out[:, 0].sum().backward()
    with torch.no_grad():
        flat = oracle_grid_encode(x + torch.tensor([[1e-4, 0, 0]]), emb, offs, res, 1.0)
    assert torch.allclose(flat[:, 0], out[:, 0].detach(), atol=1e-7)
    assert x.grad[0, 0].abs() > 0


def test_second_derivation_agrees_on_1e5_points():
    """oracle/hashgrid.c against oracle/hashgrid_np.py -- a second restatement written independently from the .cu
    (vectorised numpy over [points, corners]) -- on 10^5 points: interior, the border band, exact cell boundaries and
    out-of-range inputs, all 16 levels (dense rows for levels 0-5, hashed from level 6: 37^3 > 32768): features to 2
    ulp, d/du to 1e-5 of the level's slope scale, embedding gradients to 1e-6 relative."""
    from oracle import hashgrid_np as hnp
    from oracle.hashgrid import _OracleGridEncode
    emb, offs, res, _ = make()
    M = 100_000
    u = (synth.hash_tensor((M, 3), 9100, 0.55) + 0.5).clamp_(-0.05, 1.05)             # ~9% outside [0,1] before the clamp
    u[:2000] = (torch.round(u[:2000] * 37) + 0.5) / 37                                   # level-6 cell boundaries (res 37)
    u[2000:4000] = torch.round(u[2000:4000] * 32) / 32                                   # level-5 cell centres / borders
    u[4000:4200, 0] = 0.0
    u[4200:4400, 1] = 1.0
    u[4400:4500] = 1.0 + 1e-7                                                            # just outside
    offs_np, res_np, emb_np = offs.numpy(), res.numpy(), emb.numpy()
    for n_levels in (16, 8):
        ut = u.clone().requires_grad_(True)
        embt = emb.clone().requires_grad_(True)
        out_c = _OracleGridEncode.apply(ut, embt, offs, res, n_levels, True)
        out_np = hnp.forward(u.numpy(), emb_np, offs_np, res_np, n_levels)
        ulp = np.spacing(np.abs(out_np).astype(np.float32)).astype(np.float64)
        assert np.all(np.abs(out_c.detach().numpy().astype(np.float64) - out_np) <= 2 * ulp + 1e-12)
        # the dense -> hash switch really happens between levels 5 and 6
        T = offs_np[1:] - offs_np[:-1]
        assert res_np[5] ** 3 <= T[5] and res_np[6] ** 3 > T[6]
        go = synth.hash_tensor((M, 32), 9101, 1.0)
        (out_c * go).sum().backward()
        g_np = hnp.backward_embeddings(u.numpy(), go.numpy(), offs_np, res_np, n_levels, emb.shape[0], 2)
        ge = embt.grad.numpy().astype(np.float64)
        assert np.abs(ge - g_np).max() <= 1e-6 * np.abs(g_np).max() + 1e-9
        d_np = hnp.dy_du(u.numpy()[:20000], emb_np, offs_np, res_np, n_levels)          # [m, L, 3, C]
        gu_np = np.einsum("mldc,mlc->md", d_np.astype(np.float64), go.numpy()[:20000].reshape(-1, 16, 2).astype(np.float64))
        gu_c = ut.grad.numpy()[:20000].astype(np.float64)
        assert np.abs(gu_c - gu_np).max() <= 1e-5 * np.abs(gu_np).max()


def test_float64_encoder_agrees_with_the_c_restatement():
    """oracle/hashgrid_f64.py (the yardstick's encoder: vectorised torch, double) against oracle/hashgrid.c (scalar C, fp32) on
    20 000 points incl. out-of-box ones: the same cells, corners, hash and weights -- the difference is the fp32 round-off of the
    position (u * res - 0.5 carries ~res * 6e-8 of a cell) times the feature slope."""
    import torch
    from morpheus_amd import synth
    from oracle.hashgrid import OracleGridEncoder
    from oracle.hashgrid_f64 import OracleGridEncoderF64
    kw = dict(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=15, desired_resolution=128)
    a, b = OracleGridEncoder(**kw), OracleGridEncoderF64(**kw)
    emb = synth.make_state("b")["encoder.embeddings"]
    a.embeddings.data.copy_(emb)
    b.embeddings.data.copy_(emb.double())
    assert torch.equal(a.offsets, b.offsets)
    x = synth.hash_tensor((20000, 3), 300, 1.15)
    with torch.no_grad():
        for ml in (None, 0.5):
            ya, yb = a(x, bound=1.01, max_level=ml), b(x.double(), bound=1.01, max_level=ml)
            assert yb.dtype == torch.float64
            assert float((ya.double() - yb).abs().max()) < 5e-6 and float(yb.abs().max()) > 0.05
            assert torch.equal(ya == 0, yb == 0) or float(((ya == 0) != (yb == 0)).float().mean()) < 1e-4   # OOB points and skipped levels
