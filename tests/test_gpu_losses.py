"""GPU: the caller-side glue kernels (csrc/losses.hip) against the torch operator chains they replace -- the expressions of the
reference's loss code (morpheus.py:518-528, :556, :764-777, :1090-1145) written with torch operators, values and gradients."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def _torch_mean(kind, a, b, valid, n_valid, w_row):
    if kind == "identity":
        f = a
    elif kind == "square":
        f = a ** 2
    elif kind == "abs":
        f = a.abs()
    elif kind == "entropy":
        x = a.clamp(1e-5, 1 - 1e-5)
        f = -x * torch.log2(x) - (1 - x) * torch.log2(1 - x)
    elif kind == "eikonal":
        f = (torch.linalg.norm(a, ord=2, dim=-1) - 1.0) ** 2
    elif kind == "absdiff":
        f = (a - b).abs()
    else:
        f = torch.square(a - b)
    per_row = f.numel() // f.shape[0]
    if w_row is not None:
        w = w_row.view(-1, *([1] * (f.dim() - 1)))
        return (f * w).sum() / (per_row * w_row.sum()).clamp(min=1.0)
    if valid is None:
        return f.mean()
    w = valid.view(-1, *([1] * (f.dim() - 1))).to(f.dtype)
    return (f * w).sum() / (n_valid.clamp(min=1).to(f.dtype) * per_row)


@pytest.mark.parametrize("kind,C", [("identity", 1), ("square", 1), ("abs", 3), ("entropy", 1), ("eikonal", 3), ("absdiff", 3),
                                    ("absdiff", 2), ("sqdiff", 3)])
@pytest.mark.parametrize("mode", ["all_rows", "n_valid", "row_weight"])
def test_masked_mean_equals_the_torch_chain(kind, C, mode):
    from morpheus_amd import ops
    torch.manual_seed(3)
    M = 70001
    shape = (M,) if C == 1 and kind != "eikonal" else (M, C)
    a = torch.randn(shape, device=DEV) * (0.6 if kind != "entropy" else 0.4) + (0.5 if kind == "entropy" else 0.0)
    if kind == "entropy":
        a[:50] = 0.0                     # clamped entries: value at the clamp, no gradient
        a[50:100] = 1.5
    if kind == "eikonal":
        a[7] = 0.0                       # |row| = 0: the norm's gradient is 0 there
    b = torch.randn(shape, device=DEV) if kind in ("absdiff", "sqdiff") else None
    if b is not None:
        b[:100] = a[:100]                # a == b: sign(0) = 0
    n_valid = valid = w_row = None
    if mode == "n_valid":
        n_valid = torch.tensor(51234, dtype=torch.int32, device=DEV)
        valid = torch.arange(M, device=DEV) < n_valid
        a[51234:] = float("nan") if kind not in ("absdiff", "sqdiff") else a[51234:]     # padding is never read into the sum
    if mode == "row_weight":
        w_row = (torch.rand(M, device=DEV) < 0.7).float()
    a1, a2 = a.clone().requires_grad_(True), a.clone().requires_grad_(True)
    b1 = b2 = None
    if b is not None:
        b1, b2 = b.clone().requires_grad_(True), b.clone().requires_grad_(True)
    got = ops.masked_mean(kind, a1, b1, n_valid=n_valid, row_weight=w_row)
    a2m = a2 if mode != "n_valid" else torch.where(valid.view(-1, *([1] * (a2.dim() - 1))), a2, torch.zeros_like(a2))
    want = _torch_mean(kind, a2m, b2, valid, n_valid, w_row)
    assert _rel(got, want) <= 3e-6, (float(got), float(want))
    (got * 1.7).backward()
    (want * 1.7).backward()
    ga, wa = a1.grad, a2.grad
    if mode == "n_valid":
        assert float(ga[51234:].abs().max()) == 0.0
    assert _rel(ga, wa) <= 3e-6, _rel(ga, wa)
    if b is not None:
        assert _rel(b1.grad, b2.grad) <= 3e-6


def test_masked_mean_of_an_empty_selection_is_zero():
    from morpheus_amd import ops
    a = torch.randn(100, 3, device=DEV, requires_grad=True)
    got = ops.masked_mean("abs", a, n_valid=torch.tensor(0, dtype=torch.int32, device=DEV))
    got.backward()
    assert float(got) == 0.0 and float(a.grad.abs().max()) == 0.0


def _torch_ortho(x, normals, phi, scale):
    n = torch.nn.functional.normalize(normals, dim=-1)
    u = torch.nn.functional.normalize(torch.stack([n[..., 1], -n[..., 0], n[..., 2] * 0.0], -1), dim=-1)
    v = torch.cross(n, u, dim=-1)
    return x + (torch.cos(phi) * u + torch.sin(phi) * v) * scale


def test_ortho_perturb_equals_the_torch_chain():
    from morpheus_amd import ops
    torch.manual_seed(5)
    M = 40000
    x = torch.randn(M, 3, device=DEV)
    nrm = torch.randn(M, 3, device=DEV) * torch.rand(M, 1, device=DEV) * 3
    nrm[0] = torch.tensor([0.0, 0.0, 1.0])          # u_raw = 0: u = 0 / eps = 0, the point does not move
    nrm[1] = 0.0                                     # zero normal
    phi = torch.rand(M, 1, device=DEV) * 2 * math.pi
    gout = torch.randn(M, 3, device=DEV)
    for scale in (0.004, 1.0):
        x1, n1 = x.clone().requires_grad_(True), nrm.clone().requires_grad_(True)
        x2, n2 = x.double().clone().requires_grad_(True), nrm.double().clone().requires_grad_(True)
        got = ops.ortho_perturb(x1, n1, phi, scale)
        want = _torch_ortho(x2, n2, phi.double(), scale)
        assert float((got.double() - want).abs().max()) <= 2e-6 * max(scale, 1.0) + 1e-6 * float(x.abs().max())
        # the displacement itself (the point minus x) to fp32 round-off of the unit vectors
        assert float(((got.double() - x.double()) - (want - x2)).abs().max()) <= 4e-7 * float(x.abs().max()) + 2e-6 * scale
        got.backward(gout)
        want.backward(gout.double())
        assert torch.equal(x1.grad, gout)
        sel = torch.ones(M, dtype=torch.bool, device=DEV)
        sel[:2] = False                              # degenerate rows: compared separately below
        err = (n1.grad.double() - n2.grad)[sel].abs().max() / n2.grad[sel].abs().max()
        assert float(err) <= 2e-5, float(err)
        assert torch.isfinite(n1.grad).all()


def test_weighted_sum_equals_the_chain_of_adds():
    from morpheus_amd import ops
    ts = [torch.tensor(v, device=DEV, requires_grad=True) for v in (0.3, -1.2, 4.0, 0.07)]
    ws = (1.0, 0.5, 0.01, 30.0)
    got = ops.weighted_sum(list(zip(ws, ts)))
    want = sum(w * t.detach().double() for w, t in zip(ws, ts))
    assert abs(float(got) - float(want)) <= 1e-6 * abs(float(want))
    (got * 2.0).backward()
    for w, t in zip(ws, ts):
        assert abs(float(t.grad) - 2.0 * w) <= 1e-6 * abs(2.0 * w)


@pytest.mark.parametrize("B,n", [(1, 2048), (3, 700), (2, 2500)])
def test_pose_apply_equals_the_operator_chain(B, n):
    """models/pose.py + model.py:335-346 as torch operators (PoseArray.get_rotation_matrices / get_translations, float64) against the
    single launch; rows 0 and B-1 share a frame, so the per-frame gradient has to add rows up."""
    from morpheus_amd import ops
    from morpheus_amd.model import PoseArray
    torch.manual_seed(11)
    F = 9
    pa = PoseArray(F).to(DEV)
    with torch.no_grad():
        pa.data.copy_(torch.randn(F, 6, device=DEV) * 0.3)
    ids = torch.tensor([4, 7, 4][:B] if B > 1 else [4], device=DEV)
    o, d = torch.randn(B * n, 3, device=DEV), torch.nn.functional.normalize(torch.randn(B * n, 3, device=DEV), dim=-1)
    go, gd = torch.randn(B * n, 3, device=DEV), torch.randn(B * n, 3, device=DEV)
    o1, d1 = ops.pose_apply(o, d, pa.data, ids, n)
    (o1 * go).sum().backward(retain_graph=True)
    (d1 * gd).sum().backward()
    got = pa.data.grad.clone()
    pa64 = PoseArray(F).to(DEV).double()
    with torch.no_grad():
        pa64.data.copy_(pa.data.double())
    R, t = pa64.get_rotation_matrices(ids), pa64.get_translations(ids)
    o2 = (o.double().view(B, n, 3) + t[:, None]).view(-1, 3)
    d2 = (d.double().view(B, n, 1, 3) * R[:, None]).sum(-1).view(-1, 3)
    ((o2 * go.double()).sum() + (d2 * gd.double()).sum()).backward()
    assert float((o1.double() - o2).abs().max()) <= 1e-6 and float((d1.double() - d2).abs().max()) <= 1e-6
    # bit for bit what the fp32 operator chain gives (same roundings in the same order): sample positions do not move by an ulp
    R32, t32 = pa.get_rotation_matrices(ids), pa.get_translations(ids)
    assert torch.equal(o1, (o.view(B, n, 3) + t32[:, None]).view(-1, 3).detach())
    assert torch.equal(d1, (d.view(B, n, 1, 3) * R32[:, None]).sum(-1).view(-1, 3).detach())
    want = pa64.data.grad
    assert float((got.double() - want).abs().max()) <= 2e-5 * float(want.abs().max()), (got, want)
    assert float(got[[0, 1, 2, 3, 5, 6, 8]].abs().max()) == 0.0           # frames outside the batch: zero rows


def test_fused_render_loss_equals_the_reference_chain():
    """trainstep.get_gt_from_data + get_real_view_render_loss (the reference's morpheus.py:930-983 as torch operators) against
    ops.real_view_render_loss: the weighted loss, its three terms, the composited target, the valid-depth mask and the gradients."""
    from morpheus_amd import harness, ops
    from bench_support import trainstep
    torch.manual_seed(2)
    N = 3000
    tr = harness.load_config()["train"]
    data = dict(image=torch.rand(1, 3, N, 1, device=DEV), depth=(torch.rand(1, N, 1, device=DEV) * 2.0 - 0.3).clamp(min=0.0),
                mask=(torch.rand(1, N, 1, device=DEV) < 0.7).float())
    rays_o = torch.randn(1, N, 3, device=DEV) * 0.2 + torch.tensor([0.0, 0.0, -1.2], device=DEV)
    rays_d = torch.nn.functional.normalize(torch.randn(1, N, 3, device=DEV) * 0.3 + torch.tensor([0.0, 0.0, 1.0], device=DEV), dim=-1)
    bg = torch.rand(N, 3, device=DEV)
    image = torch.rand(1, N, 3, device=DEV)
    depth = torch.rand(1, N, device=DEV) * 2
    opac = torch.rand(1, N, 1, device=DEV)
    opac[0, :20] = 0.0                      # outside the clip: no gradient
    opac[0, 20:40] = 1.0
    leaves = lambda: [t.clone().requires_grad_(True) for t in (image, depth, opac)]
    # reference chain
    i1, d1, o1 = leaves()
    B, H, W = 1, N, 1
    gt_rgb, gt_depth, gt_mask = trainstep.get_gt_from_data(data, bg, B, H, W)
    pred_rgb = i1.reshape(B, H, W, 3).permute(0, 3, 1, 2).contiguous()
    want = trainstep.get_real_view_render_loss(tr, pred_rgb, d1.reshape(B, 1, H, W), o1.reshape(B, 1, H, W), gt_rgb, gt_depth, gt_mask,
                                               rays_o, rays_d)
    want_mask, _ = trainstep._valid_depth_mask(gt_depth, gt_mask, rays_o, rays_d)
    want.backward()
    # fused
    i2, d2, o2 = leaves()
    got, terms, gt_flat, valid = ops.real_view_render_loss(i2, d2, o2, data["image"], data["depth"], data["mask"], bg, rays_o, rays_d,
                                                           tr["rgb_weight"], tr["mask_weight"], tr["depth_weight"])
    got.backward()
    assert abs(float(got) - float(want)) <= 2e-6 * abs(float(want)), (float(got), float(want))
    assert torch.equal(gt_flat.view(1, 3, N, 1), gt_rgb) and torch.equal(valid.view(1, N, 1), want_mask)
    assert 0 < float(valid.sum()) < N
    for a, b, name in ((i2.grad, i1.grad, "image"), (d2.grad, d1.grad, "depth"), (o2.grad, o1.grad, "opacity")):
        assert _rel(a, b) <= 3e-6, (name, _rel(a, b))
    assert float(o2.grad[0, :40].abs().max()) == 0.0


def test_smooth_points_equals_the_torch_chain():
    """ops.smooth_points (mh_smooth_points_*): the points of get_normal_smoothness_loss (morpheus.py:530-547) -- values bit for bit
    the operator chain (csrc/losses.hip is compiled without FMA contraction), the 0 / 1 sphere weight, gradients to depth and rays."""
    from morpheus_amd import ops
    torch.manual_seed(9)
    N, K = 2048, 11
    depth = (torch.rand(1, N, device=DEV) * 2.0 + 0.2).requires_grad_()
    off = torch.linspace(-0.05, 0.05, K, device=DEV) + 0.01 * torch.rand(K, device=DEV)
    o = (torch.randn(N, 3, device=DEV) * 0.1 + torch.tensor([0.0, 0.0, 1.5], device=DEV)).requires_grad_()
    d = torch.randn(N, 3, device=DEV).mul(0.5).requires_grad_()
    pts_t = ((depth + off[:, None])[..., None] * d[None] + o[None]).view(-1, 3)
    keep_t = (torch.linalg.norm(pts_t, ord=2, dim=-1) < 1.1).float()
    pts, keep = ops.smooth_points(depth, off, o, d)
    assert torch.equal(pts, pts_t)
    near = (torch.linalg.norm(pts_t.double(), dim=-1) - 1.1).abs() < 1e-6        # the comparison may fall either way within an ulp
    assert torch.equal(keep[~near], keep_t[~near]) and 0 < float(keep.mean()) < 1
    g = torch.randn_like(pts_t)
    gt = torch.autograd.grad(pts_t, [depth, o, d], g)
    gh = torch.autograd.grad(pts, [depth, o, d], g)
    for a, b in zip(gh, gt):
        assert a.shape == b.shape and _rel(a, b) < 2e-6
    # a subset of the gradients only (rays without a gradient of their own: no pose optimisation)
    pts2, _ = ops.smooth_points(depth, off, o.detach(), d.detach())
    (g2,) = torch.autograd.grad(pts2, [depth], g)
    assert torch.equal(g2, gh[0])


def test_bg_blend_equals_the_torch_chain():
    """ops.bg_blend (mh_bg_blend_*): image = color + (1 - opacity) * bg (morpheus.py:686-694) -- the chain's bits forward, its
    gradients to colour, opacity and a background that has one (the background net of a virtual view)."""
    from morpheus_amd import ops
    torch.manual_seed(10)
    N = 5000
    color = torch.rand(N, 3, device=DEV).requires_grad_()
    opacity = torch.rand(N, 1, device=DEV).requires_grad_()
    bg = torch.rand(N, 3, device=DEV).requires_grad_()
    ref = color + (1 - opacity) * bg
    out = ops.bg_blend(color, opacity, bg)
    assert torch.equal(out, ref)
    g = torch.randn_like(ref)
    gt = torch.autograd.grad(ref, [color, opacity, bg], g)
    gh = torch.autograd.grad(out, [color, opacity, bg], g)
    assert torch.equal(gh[0], gt[0]) and torch.equal(gh[2], gt[2]) and gh[1].shape == gt[1].shape and _rel(gh[1], gt[1]) < 1e-6
    out2 = ops.bg_blend(color, opacity, bg.detach())
    g2 = torch.autograd.grad(out2, [color, opacity], g)
    assert torch.equal(g2[0], gh[0]) and torch.equal(g2[1], gh[1])
