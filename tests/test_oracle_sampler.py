"""CPU: properties of the two samplers' oracle restatements (no reference counterpart exists -- nerfacc's marcher
is absent from the reference tree -- so these pin the definitions the HIP kernels are checked against)."""
import torch

from morpheus_amd import synth
from oracle import field as of


def _rays(hw=24):
    o, d, t, rid = synth.frame_rays(25, hw, hw)
    o, d = o[0], d[0]
    o = torch.cat([o, torch.tensor([[3.0, 3.0, 3.0], [0.0, 0.0, 2.0]])])
    d = torch.cat([d, torch.tensor([[1.0, 0.0, 0.0], [0.0, 0.0, -1.0]])])
    return o, d


def _slab(o, d, bound=1.01):
    ta, tb = (-bound - o) / d, (bound - o) / d
    tmin = torch.minimum(ta, tb).amax(-1).clamp(min=0)
    tmax = torch.maximum(ta, tb).amin(-1)
    hit = tmax > tmin
    return torch.where(hit, tmin, torch.zeros_like(tmin)), torch.where(hit, tmax, torch.zeros_like(tmax)), hit


def test_uniform_sampler_properties():
    o, d = _rays()
    N, S = o.shape[0], 64
    jit = synth.ray_jitter(N)
    ri, ts, te = of.uniform_samples(o, d, jit, S, 1.01)
    tmin, tmax, hit = _slab(o, d)
    ts, te = ts.view(N, S), te.view(N, S)
    assert torch.equal(ri, torch.arange(N).repeat_interleave(S))
    assert (ts >= tmin[:, None] - 1e-6).all() and (te <= tmax[:, None] + 1e-6).all()          # never leaves the clipped segment
    assert (te[hit] > ts[hit]).all() and (ts[hit][:, 1:] >= te[hit][:, :-1] - 1e-6).all()      # ordered, non-overlapping bins
    assert (te[~hit] == 0).all() and (ts[~hit] == 0).all() and (~hit).sum() == 1               # the miss: zero-width samples
    # points lie inside the box
    x = o[ri] + d[ri] * ((ts.reshape(-1) + te.reshape(-1)) / 2)[:, None]
    assert (x[hit.repeat_interleave(S)].abs() <= 1.01 + 1e-5).all()


def test_marcher_properties():
    o, d = _rays()
    N = o.shape[0]
    jit = synth.ray_jitter(N)
    tmin, tmax, hit = _slab(o, d)
    full = torch.ones(128, 128, 128, dtype=torch.uint8)
    ri, ts, te = of.march_samples(o, d, jit, 0.01, 1.01, full)
    cnt = torch.bincount(ri, minlength=N)
    # full grid: every step of the clipped segment, contiguous, last interval clipped to t_far
    expect = torch.ceil((tmax - (tmin + jit * 0.01)) / 0.01).clamp(min=0)
    assert ((cnt.float() - expect).abs() <= 1).all() and cnt[~hit].sum() == 0
    for r in (0, 100, N - 1):
        a, b = ts[ri == r], te[ri == r]
        assert torch.allclose(a[1:], b[:-1], atol=2e-6) and abs(float(b[-1]) - float(tmax[r])) < 1e-6
        assert abs(float(a[0]) - float(tmin[r] + jit[r] * 0.01)) < 1e-6
    assert (ri[1:] >= ri[:-1]).all()                                                          # packed, ray-major
    # sphere grid: samples are a subset of the full march and all midpoints fall in occupied cells
    c = (torch.arange(128).float() + 0.5) / 128 * 2.02 - 1.01
    X, Y, Z = torch.meshgrid(c, c, c, indexing="ij")
    ball = ((X ** 2 + Y ** 2 + Z ** 2).sqrt() < 0.6).to(torch.uint8)
    ri2, ts2, te2 = of.march_samples(o, d, jit, 0.01, 1.01, ball)
    assert 0 < ri2.numel() < ri.numel()
    x = o[ri2] + d[ri2] * ((ts2 + te2) / 2)[:, None]
    assert (x.norm(dim=-1) < 0.6 + 0.03).all()
    # empty grid
    assert of.march_samples(o, d, jit, 0.01, 1.01, torch.zeros(128, 128, 128, dtype=torch.uint8))[0].numel() == 0
