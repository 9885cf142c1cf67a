"""GPU parity: each C-ABI operator of libmorpheus_hip.so against the CPU oracle, same seeded inputs.

Tolerances: integer/index work (sampler, packed info) is bit-exact; fp32 kernels are compared with
rel = |a-b| / max(|b|, floor) and the floor is written next to every check.
"""
import numpy as np
import pytest
import torch

from morpheus_amd import synth
from oracle import field as of
from oracle.hashgrid import level_resolutions, oracle_grid_encode
from tests.util import assert_close, max_rel

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _set_mlp(monkeypatch, ops, mode):
    """arithmetic of the MLP kernels for this test: "b3" (bf16 x 3 slices, the fp32-faithful default)
    or "f32" (native fp32 MFMA); restored by monkeypatch at teardown"""
    monkeypatch.setattr(ops, "_mode", ops.mlp_mode())      # registers the restore
    ops.set_mlp_mode(mode)


def _grid_setup(scale=0.1):
    offs, s = synth.grid_offsets()
    emb = synth.hash_tensor((int(offs[-1]), 2), 9001, scale)
    return emb, offs, level_resolutions(16, s, 16)


@pytest.fixture
def grid_rows_staged():
    """every brick-binned d/dx backward of the test takes the form that stages a brick's table rows in LDS (the library takes it
    for calls of >= 2^20 points only); restored at teardown"""
    from morpheus_amd import _lib
    lib = _lib.load()
    before = lib.mh_grid_stage_min_points(-1)
    assert lib.mh_grid_stage_min_points(0) == 0
    yield
    lib.mh_grid_stage_min_points(before)


@pytest.mark.parametrize("staged", [False, True])
@pytest.mark.parametrize("max_level", [None, 0.5])
def test_grid_encode_forward_backward(max_level, staged, request):
    from morpheus_amd import ops
    if staged:
        request.getfixturevalue("grid_rows_staged")
    emb, offs, res = _grid_setup()
    x = synth.hash_tensor((20000, 3), 9002, 1.1)            # ~25% of the points fall outside the +-1.01 box
    w = synth.hash_tensor((20000, 32), 9004, 1.0)
    # oracle
    xo, eo = x.clone().requires_grad_(True), emb.clone().requires_grad_(True)
    out_o = oracle_grid_encode(xo, eo, torch.from_numpy(offs), torch.from_numpy(res), 1.01, max_level)
    (out_o * w).sum().backward()
    # HIP
    xg, eg = x.to(DEV).requires_grad_(True), emb.to(DEV).requires_grad_(True)
    out_g = ops.grid_encode(xg, eg, offs, res, 1.01, max_level)
    (out_g * w.to(DEV)).sum().backward()
    assert_close(out_g, out_o, 1e-5, "hash features", floor=1e-2)
    oob = ~(x.abs() <= 1.01).all(-1)
    assert oob.sum() > 1000 and (out_g.cpu()[oob] == 0).all() and (xg.grad.cpu()[oob] == 0).all()
    assert_close(xg.grad, xo.grad, 1e-4, "d/dx (kernel dy_dx definition)", floor=1e-2 * float(xo.grad.abs().max()))
    # embedding gradient: atomics reorder the sums -> compare against the row scale
    ge, go = eg.grad.cpu(), eo.grad
    assert float((ge - go).abs().max()) <= 2e-5 * float(go.abs().max())
    assert torch.equal(ge == 0, go == 0) or float(((ge == 0) != (go == 0)).float().mean()) < 1e-4
    if max_level is not None:
        assert (out_g[:, 16:] == 0).all()


@pytest.mark.parametrize("n_levels", [16, 9])
def test_grid_forward_binned_bit_identical(n_levels):
    """mh_grid_encode_fwd_binned (points binned into bricks first, a brick's rows staged in LDS) against mh_grid_encode_fwd on
    the same points: the same bits -- rays converging near a camera (hot bricks split into several work items), uniform points,
    a quarter of them outside the box (zero rows, written by the launch's surplus workgroups), points on the box's faces, and
    levels switched off (zero columns)."""
    from morpheus_amd import ops, _lib
    lib = _lib.load()
    emb, offs, res = _grid_setup()
    g = torch.Generator().manual_seed(5)
    n_r, S = 2048, 96
    o = torch.tensor([0.0, 0.0, 2.2]) + 0.05 * torch.randn(n_r, 3, generator=g)
    d = torch.nn.functional.normalize(torch.cat([torch.randn(n_r, 2, generator=g) * 0.35, -torch.ones(n_r, 1)], 1), dim=1)
    ts = 1.2 + 2.0 * (torch.arange(S).float()[None] + torch.rand(n_r, S, generator=g)) / S
    rays = (o[:, None] + d[:, None] * ts[..., None]).reshape(-1, 3)
    faces = torch.rand(4096, 3, generator=g) * 2.02 - 1.01
    faces[torch.arange(4096), torch.randint(0, 3, (4096,), generator=g)] = 1.01 * (torch.randint(0, 2, (4096,), generator=g) * 2 - 1).float()
    x = torch.cat([rays, torch.rand(150_000, 3, generator=g) * 2.6 - 1.3, faces]).to(DEV).contiguous()
    M = x.shape[0]
    embg = emb.to(DEV)
    o_np, o_p = ops._i32arr(offs)
    r_np, r_p = ops._i32arr(res)
    ref = torch.full((M, 32), 7.0, device=DEV)
    ops.check(lib.mh_grid_encode_fwd(ops.ptr(x), ops.ptr(embg), o_p, r_p, ops.ptr(ref), M, 16, n_levels, 1.01, 1, ops.stream()), "fwd")
    perm, bstart = ops._bin_points(lib, x, 1.01)
    out = torch.full((M, 32), 7.0, device=DEV)                # every row must be written, the outside points' too
    ops.check(lib.mh_grid_encode_fwd_binned(ops.ptr(x), ops.ptr(embg), o_p, r_p, ops.ptr(perm), ops.ptr(bstart), ops.ptr(out), M, 16,
                                            n_levels, 1.01, ops.stream()), "fwd_binned")
    outside = ~(x.abs() <= 1.01).all(-1)
    assert int(outside.sum()) > 20_000 and bool((out[outside] == 0).all())
    assert torch.equal(out, ref)
    # ... and through the autograd op, which takes the binned form by call size and hands its binning to the backward
    before = lib.mh_grid_stage_min_points(0)
    try:
        xs, e = x.clone().requires_grad_(True), embg.clone().requires_grad_(True)
        feat = ops.grid_encode(xs, e, offs, res, 1.01, max_level=n_levels / 16.0)
        assert torch.equal(feat.detach(), ref)
        n6 = M // 6 * 6                                        # a large call with the finite-difference-tap hint is binned as well
        assert torch.equal(ops.grid_encode(x[:n6], embg, offs, res, 1.01, max_level=n_levels / 16.0, group=6), ref[:n6])
        gw = torch.randn(M, 32, generator=g).to(DEV)
        (feat * gw).sum().backward()
        lib.mh_grid_stage_min_points(1 << 40)
        xs2, e2 = x.clone().requires_grad_(True), embg.clone().requires_grad_(True)
        (ops.grid_encode(xs2, e2, offs, res, 1.01, max_level=n_levels / 16.0) * gw).sum().backward()
    finally:
        lib.mh_grid_stage_min_points(before)
    assert torch.equal(xs.grad, xs2.grad)
    assert float((e.grad - e2.grad).abs().max()) / float(e2.grad.abs().max()) <= 1e-6


@pytest.mark.parametrize("M", [1, 5, 63, 1025, 2049])
def test_grid_staged_forms_on_tiny_and_ragged_calls(M, grid_rows_staged):
    """the brick-staged forward and d/dx backward forced onto calls far below their size class -- one point, a handful, one short of
    a wave, one past a work item, one past a large work item; some points outside the box, all of them outside in a second pass --
    against the gathering forms: the same features and d/dx bits, table gradients to the fixed-point resolution."""
    from morpheus_amd import ops, _lib
    lib = _lib.load()
    emb, offs, res = _grid_setup()
    g = torch.Generator().manual_seed(100 + M)
    for spread in (1.2, 0.0):
        x = (torch.rand(M, 3, generator=g) * 2 - 1) * spread if spread else torch.full((M, 3), 1.5)
        x, gw, embg = x.to(DEV), torch.randn(M, 32, generator=g).to(DEV), emb.to(DEV)

        def run():
            xs, e = x.clone().requires_grad_(True), embg.clone().requires_grad_(True)
            f = ops.grid_encode(xs, e, offs, res, 1.01)
            (f * gw).sum().backward()
            return f.detach(), xs.grad, e.grad

        f1, gx1, ge1 = run()                                   # staged (the fixture set the knob to 0)
        lib.mh_grid_stage_min_points(1 << 40)
        try:
            f2, gx2, ge2 = run()
        finally:
            lib.mh_grid_stage_min_points(0)
        assert torch.equal(f1, f2) and torch.equal(gx1, gx2)
        assert float((ge1 - ge2).abs().max()) <= 1e-6 * float(ge2.abs().max()) + 1e-30
        if not spread:
            assert not f1.any() and not gx1.any() and not ge1.any()


def test_grid_encode_edge_cases():
    from morpheus_amd import ops
    emb, offs, res = _grid_setup()
    eg = emb.to(DEV)
    # empty input, a single point, exact box corners and cell centres
    assert ops.grid_encode(torch.zeros(0, 3, device=DEV), eg, offs, res, 1.01).shape == (0, 32)
    pts = torch.tensor([[0.0, 0.0, 0.0], [1.01, 1.01, 1.01], [-1.01, -1.01, -1.01], [1.0100001, 0, 0], [0.3, -0.7, 1.0]])
    o = oracle_grid_encode(pts, emb, torch.from_numpy(offs), torch.from_numpy(res), 1.01)
    g = ops.grid_encode(pts.to(DEV), eg, offs, res, 1.01)
    assert_close(g, o, 1e-5, "edge points", floor=1e-2)


def _ragged_samples(n_rays, seed):
    rng = np.random.RandomState(seed)
    cnt = rng.randint(0, 200, size=n_rays)
    cnt[3] = 0
    cnt[7] = 1
    cnt[11] = 64
    cnt[12] = 65
    cnt[13] = 350
    ri = np.repeat(np.arange(n_rays), cnt)
    M = int(cnt.sum())
    ts = np.concatenate([np.sort(rng.rand(c)).astype(np.float32) * 2.5 for c in cnt]) if M else np.zeros(0, np.float32)
    dt = (rng.rand(M).astype(np.float32) * 0.02 + 0.002)
    return torch.from_numpy(ri), torch.from_numpy(ts), torch.from_numpy(ts + dt), cnt


def test_composite_forward_backward_ragged():
    from morpheus_amd import ops
    N = 300
    ri, ts, te, cnt = _ragged_samples(N, 5)
    M = ri.shape[0]
    sig = synth.hash_tensor((M,), 77, 20.0, 20.0)
    rgb = synth.hash_tensor((M, 3), 78, 0.5, 0.5)
    wgt = synth.hash_tensor((M,), 79, 1.0)
    timg, tdep, topa = synth.hash_tensor((N, 3), 80, 0.5, 0.5), synth.hash_tensor((N,), 81, 1.0, 1.0), synth.hash_tensor((N,), 82, 0.5, 0.5)

    def loss(w, o, d, c, dev):
        return ((c - timg.to(dev)) ** 2).sum() + ((d - tdep.to(dev)) ** 2).sum() + ((o - topa.to(dev)) ** 2).sum() + \
            (w * wgt.to(dev)).sum()

    so, ro = sig.clone().requires_grad_(True), rgb.clone().requires_grad_(True)
    w_o, _, _ = of.render_weights(ts, te, so, ri, N)
    o_o = of.accumulate(w_o, None, ri, N)[:, 0]
    d_o = of.accumulate(w_o, ((ts + te) / 2)[:, None], ri, N)[:, 0]
    c_o = of.accumulate(w_o, ro, ri, N)
    loss(w_o, o_o, d_o, c_o, "cpu").backward()
    # the sequential per-ray loop agrees with the cumsum formulation
    assert_close(w_o, of.render_weights_loop(ts, te, sig, ri, N), 1e-4, "oracle loop vs cumsum", floor=1e-4)

    sg, rg = sig.to(DEV).requires_grad_(True), rgb.to(DEV).requires_grad_(True)
    rs, rc = ops.packed_info(ri.to(DEV), N)
    assert torch.equal(rc.cpu(), torch.from_numpy(cnt).int())
    w_g, o_g, d_g, c_g = ops.composite(sg, ts.to(DEV), te.to(DEV), rg, rs, rc)
    loss(w_g, o_g, d_g, c_g, DEV).backward()
    assert_close(w_g, w_o, 1e-4, "weights (wave scan vs fp64 cumsum)", floor=1e-4)
    assert_close(o_g, o_o, 1e-5, "opacity", floor=1e-3)
    assert_close(d_g, d_o, 1e-5, "depth", floor=1e-3)
    assert_close(c_g, c_o, 1e-5, "color", floor=1e-3)
    assert_close(sg.grad, so.grad, 1e-4, "d sigma", floor=1e-4)
    assert_close(rg.grad, ro.grad, 2e-4, "d rgb (= w * g: inherits the weights' scan round-off)", floor=1e-4)


def test_sampler_and_raygen_bit_exact():
    from morpheus_amd import ops
    o, d, t, rid = synth.frame_rays(25, 64, 64)
    o, d = o[0], d[0]
    # add rays that miss the box and axis-parallel rays (division by zero in the slab test)
    o = torch.cat([o, torch.tensor([[3.0, 3.0, 3.0], [0.0, 0.0, 2.0], [0.5, 0.5, 2.0]])])
    d = torch.cat([d, torch.tensor([[1.0, 0.0, 0.0], [0.0, 0.0, -1.0], [0.0, 0.0, 1.0]])])
    N = o.shape[0]
    jit = synth.ray_jitter(N)
    for S in (64, 128, 7):
        ri_o, ts_o, te_o = of.uniform_samples(o, d, jit, S, 1.01)
        ri, ts, te, xyz, rs, rc = ops.sample_uniform(o.to(DEV), d.to(DEV), jit.to(DEV), S, 1.01, with_xyz=True)
        assert torch.equal(ri.cpu().long(), ri_o)
        assert torch.equal(ts.cpu(), ts_o), float((ts.cpu() - ts_o).abs().max())
        assert torch.equal(te.cpu(), te_o)
        assert torch.equal(rs.cpu().long(), torch.arange(N) * S) and (rc.cpu() == S).all()
        xo = o[ri_o] + d[ri_o] * ((ts_o + te_o) / 2.0)[:, None]
        assert torch.equal(xyz.cpu(), xo)
    pose = synth.look_at_pose(70.0, 33.0)
    ro, rd = synth.camera_rays(48, 40, pose)
    go, gd = ops.generate_rays(1.2 * 40, 1.2 * 40, 20.0, 24.0, pose, 48, 40, DEV)
    assert torch.equal(go.cpu(), ro) and torch.equal(gd.cpu(), rd)


def _state(kind, dev=None, grad=True):
    st = synth.make_state(kind)
    out = {}
    for k, v in st.items():
        v = v.clone() if dev is None else v.to(dev)
        out[k] = v.requires_grad_(True) if (grad and v.is_floating_point()) else v
    return out


def _wn(p, pre, l):
    return of.wn_weight(p[f"{pre}.net.{l}.weight_g"], p[f"{pre}.net.{l}.weight_v"])


@pytest.mark.parametrize("kind", ["a", "b"])
@pytest.mark.parametrize("max_level", [None, 0.5])
@pytest.mark.parametrize("mlp", ["b3", "f32"])
def test_warp_mlp(kind, max_level, mlp, monkeypatch):
    """deform_net + topo_net (fused MFMA kernels: bf16 x 3 sliced operands on the bf16 matrix pipe, and native fp32 MFMA) vs the oracle, values and every gradient."""
    from morpheus_amd import ops
    _set_mlp(monkeypatch, ops, mlp)
    M = 1000                                              # not a multiple of 128: exercises the ragged tail
    x = synth.hash_tensor((M, 3), 500, 1.0)
    tvals = torch.tensor([37 / 200, 0.5, 0.91])
    slot = (torch.arange(M) % 3).int()
    t = tvals[slot.long()][:, None]
    wd_, wt_ = synth.hash_tensor((M, 3), 501, 1.0), synth.hash_tensor((M, 2), 502, 1.0)
    n_bands = 6 if max_level is None else int(max_level * 6)
    # oracle
    po = _state(kind)
    xo = x.clone().requires_grad_(True)
    f = of.OracleField(po, 1.01, max_level)
    d_o, t_o = f.warp(xo, t)
    ((d_o * wd_).sum() + (t_o * wt_).sum()).backward()
    # HIP: parameters prepared in torch exactly as morpheus_amd.model does
    pg = _state(kind, DEV)
    xg = x.to(DEV).requires_grad_(True)
    code = of.multicode_sample([pg[f"deform_code.volumes.{k}"] for k in range(3)], tvals.to(DEV)[:, None])
    plist, b0s = [], []
    for pre, nout in (("deform_net", 3), ("topo_net", 2)):
        W = [_wn(pg, pre, l) for l in range(6)]
        b = [pg[f"{pre}.net.{l}.bias"] for l in range(6)]
        plist.append([W[0][:, :39]] + W[1:] + b)
        b0s.append(torch.addmm(b[0], code, W[0][:, 39:].t()))
    d_g, t_g = ops.warp_mlp(xg, slot.to(DEV), b0s[0], b0s[1], n_bands, ops.prepare_warp_operands(plist[0], plist[1]))
    ((d_g * wd_.to(DEV)).sum() + (t_g * wt_.to(DEV)).sum()).backward()
    assert_close(d_g, d_o, 1e-4, "deform", floor=1e-2 * float(d_o.abs().max()))
    assert_close(t_g, t_o, 1e-4, "topo", floor=1e-2 * float(t_o.abs().max()))
    assert_close(xg.grad, xo.grad, 2e-4, "d/dx", floor=1e-2 * float(xo.grad.abs().max()))
    n = 0
    for k, v in po.items():
        if v.is_floating_point() and v.grad is not None and ("deform" in k or "topo" in k):
            gg = pg[k].grad
            assert gg is not None, k
            assert_close(gg, v.grad, 3e-4, "grad " + k, floor=1e-2 * float(v.grad.abs().max()) + 1e-12)
            n += 1
    assert n >= 39


@pytest.mark.parametrize("kind", ["a", "b"])
@pytest.mark.parametrize("with_color", [True, False])
@pytest.mark.parametrize("fwd", ["b3", "f32"])
def test_field_mlp(kind, with_color, fwd, monkeypatch):
    """sdf_net + Laplace density + color_net (fused MFMA kernels; bf16 x 3 slices = the default, forward and fused backward; native fp32 MFMA in
    the f32 mode) vs the oracle, given identical hash features (the
    hash grid itself is checked above)."""
    from morpheus_amd import ops
    _set_mlp(monkeypatch, ops, fwd)
    M = 777
    x = synth.hash_tensor((M, 3), 600, 1.0)
    fs, fc = synth.hash_tensor((M, 32), 601, 0.1), synth.hash_tensor((M, 32), 602, 0.1)
    topo = synth.hash_tensor((M, 2), 603, 0.3)
    ws, wg, wc = synth.hash_tensor((M,), 604, 1.0), synth.hash_tensor((M,), 605, 0.01), synth.hash_tensor((M, 3), 606, 1.0)
    # oracle (features injected)
    po = _state(kind)
    leaves_o = [t.clone().requires_grad_(True) for t in (x, fs, fc, topo)]
    xo, fso, fco, tpo = leaves_o
    feat = torch.cat([of.freq_encode(xo, 6, None), fso, tpo], -1)
    h = of.mlp_apply(feat, po, "sdf_net", 3, False)
    sdf_o = h[:, 0]
    sig_o = of.laplace_density(sdf_o, po["sdf2density.beta"])
    lo = (sdf_o * ws).sum() + (sig_o * wg).sum()
    if with_color:
        alb_o = torch.sigmoid(of.mlp_apply(torch.cat([fco, h[:, 1:]], -1), po, "color_net", 3, True))
        lo = lo + (alb_o * wc).sum()
    lo.backward()
    # HIP
    pg = _state(kind, DEV)
    xg, fsg, fcg, tpg = [t.to(DEV).requires_grad_(True) for t in (x, fs, fc, topo)]
    Ws = [pg[f"sdf_net.net.{l}.weight"] for l in range(3)]
    Wc = [_wn(pg, "color_net", l) for l in range(3)]
    bs = [pg[f"sdf_net.net.{l}.bias"] for l in range(3)]
    bc = [pg[f"color_net.net.{l}.bias"] for l in range(3)]
    beta = pg["sdf2density.beta"].abs() + 1e-4
    sdf_g, sig_g, alb_g = ops.field_mlp(xg, fsg, fcg if with_color else None, tpg, beta, 6, with_color,
                                        ops.prepare_field_operands(Ws + Wc + bs + bc))
    lg = (sdf_g * ws.to(DEV)).sum() + (sig_g * wg.to(DEV)).sum()
    if with_color:
        lg = lg + (alb_g * wc.to(DEV)).sum()
    lg.backward()
    assert_close(sdf_g, sdf_o, 1e-4, "sdf", floor=1e-2)
    assert_close(sig_g, sig_o, 2e-4, "sigma (x10 gain on sdf round-off)", floor=1e-2)
    if with_color:
        assert_close(alb_g, alb_o, 1e-4, "albedo", floor=1e-2)
        assert_close(fcg.grad, fco.grad, 3e-4, "d feat_c", floor=1e-2 * float(fco.grad.abs().max()) + 1e-12)
    assert_close(xg.grad, xo.grad, 3e-4, "d xc (freq path)", floor=1e-2 * float(xo.grad.abs().max()))
    assert_close(fsg.grad, fso.grad, 3e-4, "d feat_s", floor=1e-2 * float(fso.grad.abs().max()) + 1e-12)
    assert_close(tpg.grad, tpo.grad, 3e-4, "d topo", floor=1e-2 * float(tpo.grad.abs().max()) + 1e-12)
    n = 0
    for k, v in po.items():
        if v.is_floating_point() and v.grad is not None:
            gg = pg[k].grad
            assert gg is not None, k
            assert_close(gg, v.grad, 3e-4, "grad " + k, floor=1e-2 * float(v.grad.abs().max()) + 1e-12)
            n += 1
    assert n >= (16 if with_color else 7)


def _sphere_grid(R=128, radius=0.6, bound=1.01):
    c = (torch.arange(R).float() + 0.5) / R * 2 * bound - bound
    X, Y, Z = torch.meshgrid(c, c, c, indexing="ij")
    return ((X ** 2 + Y ** 2 + Z ** 2).sqrt() < radius).to(torch.uint8).contiguous()


def test_occupancy_marcher_bit_exact_and_ragged():
    """Fixed-step marching of a binary occupancy grid (the reference's sampler call shape): ragged packed samples,
    bit-exact against the oracle; empty grid -> no samples; full grid -> every step of every ray."""
    from morpheus_amd import ops
    from morpheus_amd.occgrid import OccupancyGrid
    o, d, t, rid = synth.frame_rays(25, 48, 48)
    o, d = o[0], d[0]
    o = torch.cat([o, torch.tensor([[3.0, 3.0, 3.0], [0.0, 0.0, 2.0]])])         # a miss and an axis-parallel ray
    d = torch.cat([d, torch.tensor([[1.0, 0.0, 0.0], [0.0, 0.0, -1.0]])])
    N = o.shape[0]
    jit = synth.ray_jitter(N)
    rnd = (synth.hash_tensor((128, 128, 128), 4242, 0.5, 0.5) > 0.7).to(torch.uint8)
    for name, grid in (("sphere", _sphere_grid()), ("random", rnd), ("full", torch.ones(128, 128, 128, dtype=torch.uint8)),
                       ("empty", torch.zeros(128, 128, 128, dtype=torch.uint8))):
        ri_o, ts_o, te_o = of.march_samples(o, d, jit, 0.01, 1.01, grid)
        ri, ts, te, rs, rc = ops.march_rays(o.to(DEV), d.to(DEV), jit.to(DEV), 0.01, 1.01, grid.to(DEV))
        assert ri.shape[0] == ri_o.shape[0], (name, ri.shape, ri_o.shape)
        assert torch.equal(ri.cpu().long(), ri_o) and torch.equal(ts.cpu(), ts_o) and torch.equal(te.cpu(), te_o), name
        cnt = torch.bincount(ri_o, minlength=N) if ri_o.numel() else torch.zeros(N, dtype=torch.long)
        assert torch.equal(rc.cpu().long(), cnt) and torch.equal(rs.cpu().long(), torch.cumsum(cnt, 0) - cnt)
        if name == "empty":
            assert ri.numel() == 0
        if name == "full":
            assert int(rc.max()) > 150 and int(rc[N - 2]) == 0            # ~2.5 units of path / 0.01; the miss has none
    # nerfacc-shaped object: sampling() + update_every_n_steps() with a density callback
    g = OccupancyGrid([-1.01, -1.01, -1.01, 1.01, 1.01, 1.01], 128).to(DEV)
    ri, ts, te = g.sampling(o.to(DEV), d.to(DEV), render_step_size=0.01, stratified=True)
    assert ri.numel() == 0                                                  # grid starts empty
    g.update_every_n_steps(0, lambda x: (x.norm(dim=-1) < 0.5).float() * 0.05)
    frac = float(g.binaries.float().mean())
    assert 0.03 < frac < 0.09, frac                                         # ball of radius 0.5 in a 2.02 box: 6.4 %
    g.fixed_jitter = jit.to(DEV)
    ri, ts, te = g.sampling(o.to(DEV), d.to(DEV), render_step_size=0.01, stratified=True)
    ri_o, ts_o, te_o = of.march_samples(o, d, jit, 0.01, 1.01, g.binaries[0].to(torch.uint8).cpu())
    assert torch.equal(ts.cpu(), ts_o) and g.packed[1].sum().item() == ri.numel() > 1000
    # the persistent buffers of nerfacc 0.5.x's OccGridEstimator (levels = 1): what the reference's checkpoint key
    # 'estimator' holds (morpheus.py:341,355)
    sd = g.state_dict()
    assert sd["binaries"].dtype == torch.bool and tuple(sd["binaries"].shape) == (1, 128, 128, 128)
    assert sd["resolution"].tolist() == [128, 128, 128] and tuple(sd["aabbs"].shape) == (1, 6) and sd["occs"].numel() == 128 ** 3
    g2 = OccupancyGrid([-1.01, -1.01, -1.01, 1.01, 1.01, 1.01], 128).to(DEV)
    g2.load_state_dict(sd, strict=True)
    assert torch.equal(g2.binaries, g.binaries)


def test_flat_adam_matches_torch_adam():
    """mh_adam_step vs torch.optim.Adam (the reference's optimiser, morpheus.py:154-155) over several steps with
    per-group learning rates that change between steps (update_learning_rate), sparse (mostly-zero) table gradients
    and ragged group sizes."""
    from morpheus_amd.optim import FlatAdam
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(7)
    shapes = [[(4099, 2), (64, 73)], [(33,)], [(200, 6), (5,), (1,)]]
    mk = lambda: [[torch.nn.Parameter((torch.randn(*s, generator=g) * 0.1).to(dev)) for s in grp] for grp in shapes]
    mine = mk()
    ref = [[torch.nn.Parameter(p.detach().clone()) for p in grp] for grp in mine]
    groups = lambda ps: [{"name": f"g{i}", "params": grp, "lr": lr} for i, (grp, lr) in enumerate(zip(ps, (1e-2, 5e-3, 1e-3)))]
    opt = FlatAdam(groups(mine), betas=(0.9, 0.99), eps=1e-15)
    ropt = torch.optim.Adam(groups(ref), betas=(0.9, 0.99), eps=1e-15, foreach=False, fused=False)
    for it in range(6):
        opt.zero_grad()
        for gm, gr in zip(mine, ref):
            for pm, pr in zip(gm, gr):
                gv = (torch.randn(pm.shape, generator=g) * 10.0 ** float(torch.randint(-6, 1, (1,), generator=g))).to(dev)
                if pm.shape == (4099, 2):
                    gv = gv * (torch.rand(4099, 1, generator=g) < 0.05).to(dev)   # hash-table-like: 95 % exact zeros
                pm.grad = gv.clone()                                               # what backward would hand over
                pr.grad = gv.clone()
        if it % 2 == 1:
            # a step on which the 'pose-like' group gets NO gradient (the reference's virtual-view step under freeze_lr,
            # morpheus.py:1399-1408): torch.optim.Adam leaves those parameters, their moments and their step counts alone
            for pm, pr in zip(mine[2], ref[2]):
                pm.grad, pr.grad = None, None
        if it == 3:
            for o in (opt, ropt):
                o.param_groups[2]["lr"] = 2e-4
                o.param_groups[0]["lr"] *= 0.5
        opt.step()
        ropt.step()
        for gm, gr in zip(mine, ref):
            for pm, pr in zip(gm, gr):
                # an Adam step moves a parameter by at most ~lr: compare at that scale
                assert float((pm - pr).abs().max()) <= 1e-2 * 2e-5, (it, tuple(pm.shape))
    sd = opt.state_dict()
    assert float(sd["state"][0]["step"]) == 6.0 and float(sd["state"][3]["step"]) == 3.0 == float(ropt.state[ref[2][0]]["step"])
    e_p = ropt.state[ref[2][0]]["exp_avg"]
    assert float((sd["state"][3]["exp_avg"] - e_p).abs().max()) <= 1e-6 * float(e_p.abs().max())
    e_m = ropt.state[ref[0][0]]["exp_avg"]
    assert float((sd["state"][0]["exp_avg"] - e_m).abs().max()) <= 1e-6 * float(e_m.abs().max())


def test_weight_norm_all_matches_torch():
    """mh_weight_norm_fwd/bwd vs torch._weight_norm (what nn.utils.weight_norm of decoders.py:51-52 evaluates) on the
    hot path's layer shapes, with one output left unused (NULL gradient)."""
    from morpheus_amd import ops
    g = torch.Generator().manual_seed(3)
    shapes = [(128, 87), (128, 128), (128, 128), (3, 128), (2, 128), (64, 64), (3, 64)]
    vs = [torch.randn(*s, generator=g).to(DEV).requires_grad_(True) for s in shapes]
    gs = [(torch.rand(s[0], 1, generator=g) + 0.5).to(DEV).requires_grad_(True) for s in shapes]
    probes = [torch.randn(*s, generator=g).to(DEV) for s in shapes]
    ws = ops.weight_norm_all(vs, gs)
    loss = sum((w * p).sum() for i, (w, p) in enumerate(zip(ws, probes)) if i != 2)     # layer 2 gets no gradient
    loss.backward()
    for i, (v, gg, w, p) in enumerate(zip(vs, gs, ws, probes)):
        v2, g2 = v.detach().clone().requires_grad_(True), gg.detach().clone().requires_grad_(True)
        w2 = torch._weight_norm(v2, g2, 0)
        assert_close(w, w2, 1e-6, f"W[{i}]", floor=1e-3)
        if i == 2:
            assert float(v.grad.abs().max()) == 0.0 and float(gg.grad.abs().max()) == 0.0
            continue
        (w2 * p).sum().backward()
        assert_close(v.grad, v2.grad, 2e-5, f"dv[{i}]", floor=1e-2)
        assert_close(gg.grad, g2.grad, 2e-5, f"dg[{i}]", floor=1e-2)


SWEEP_SIZES = [1, 31, 32, 33, 127, 128, 129, 255, 257, 1025, 4099]


def test_mlp_and_grid_size_sweep():
    """Ragged sizes around every tile / workgroup boundary (32-point tiles, 128- and 256-point workgroups, persistent
    loops).  Per-point results do not depend on how many points ride along, so a run on the first M points must
    reproduce the first M rows of one big run BIT FOR BIT, and its parameter gradients must equal the big run's with
    the upstream gradient zeroed beyond M (same terms, different summation order) -- any contribution of padding
    lanes, clamped duplicate points or unwritten scratch tiles would show up here."""
    from morpheus_amd import ops
    MB = 4099
    pg = _state("b", DEV, grad=False)
    x = synth.hash_tensor((MB, 3), 700, 1.0).to(DEV)
    slot = (torch.arange(MB) % 2).int().to(DEV)
    tvals = torch.tensor([0.2, 0.7], device=DEV)
    code = of.multicode_sample([pg[f"deform_code.volumes.{k}"] for k in range(3)], tvals[:, None])
    wd_, wt_ = synth.hash_tensor((MB, 3), 701, 1.0).to(DEV), synth.hash_tensor((MB, 2), 702, 1.0).to(DEV)
    fs, fc = synth.hash_tensor((MB, 32), 703, 0.1).to(DEV), synth.hash_tensor((MB, 32), 704, 0.1).to(DEV)
    topo = synth.hash_tensor((MB, 2), 705, 0.3).to(DEV)
    ws, wc = synth.hash_tensor((MB,), 706, 1.0).to(DEV), synth.hash_tensor((MB, 3), 707, 1.0).to(DEV)
    offs, sc = synth.grid_offsets()
    res = level_resolutions(16, sc, 16)
    gw = synth.hash_tensor((MB, 32), 708, 1.0).to(DEV)

    def run(M, zero_beyond=None):
        """-> per-point outputs [M rows] and a dict of parameter gradients"""
        leaves = {}

        def leaf(name, t):
            leaves[name] = t.detach().clone().requires_grad_(True)
            return leaves[name]
        plist, b0s = [], []
        for pre in ("deform_net", "topo_net"):
            W = [of.wn_weight(leaf(f"{pre}.g{l}", pg[f"{pre}.net.{l}.weight_g"]), leaf(f"{pre}.v{l}", pg[f"{pre}.net.{l}.weight_v"]))
                 for l in range(6)]
            b = [leaf(f"{pre}.b{l}", pg[f"{pre}.net.{l}.bias"]) for l in range(6)]
            plist.append([W[0][:, :39]] + W[1:] + b)
            b0s.append(torch.addmm(b[0], code, W[0][:, 39:].t()))
        mask = torch.ones(M, 1, device=DEV)
        if zero_beyond is not None:
            mask[zero_beyond:] = 0
        xs = x[:M].clone().requires_grad_(True)
        d, t = ops.warp_mlp(xs, slot[:M].contiguous(), b0s[0], b0s[1], 6, ops.prepare_warp_operands(plist[0], plist[1]))
        Ws = [leaf(f"sdf.w{l}", pg[f"sdf_net.net.{l}.weight"]) for l in range(3)]
        Wc = [of.wn_weight(leaf(f"col.g{l}", pg[f"color_net.net.{l}.weight_g"]), leaf(f"col.v{l}", pg[f"color_net.net.{l}.weight_v"]))
              for l in range(3)]
        bs = [leaf(f"sdf.b{l}", pg[f"sdf_net.net.{l}.bias"]) for l in range(3)]
        bc = [leaf(f"col.b{l}", pg[f"color_net.net.{l}.bias"]) for l in range(3)]
        beta = pg["sdf2density.beta"].abs() + 1e-4
        sdf, sig, alb = ops.field_mlp(x[:M].contiguous(), fs[:M].contiguous(), fc[:M].contiguous(), topo[:M].contiguous(), beta, 6,
                                      True, ops.prepare_field_operands(Ws + Wc + bs + bc))
        emb = leaf("emb", pg["encoder.embeddings"])
        feat = ops.grid_encode(x[:M].contiguous().clone().requires_grad_(True), emb, offs, res, 1.01)
        loss = ((d * wd_[:M] + 0).sum(-1, keepdim=True) * mask).sum() + ((t * wt_[:M]).sum(-1, keepdim=True) * mask).sum() + \
            (sdf[:, None] * ws[:M, None] * mask).sum() + ((alb * wc[:M]).sum(-1, keepdim=True) * mask).sum() + \
            ((feat * gw[:M]).sum(-1, keepdim=True) * mask).sum()
        loss.backward()
        return (d.detach(), t.detach(), sdf.detach(), sig.detach(), alb.detach(), feat.detach()), \
            {k: v.grad.detach().clone() for k, v in leaves.items() if v.grad is not None}

    outs_big, _ = run(MB)
    for M in SWEEP_SIZES:
        outs, grads = run(M)
        for a, b, name in zip(outs, outs_big, ("deform", "topo", "sdf", "sigma", "albedo", "hash features")):
            assert torch.equal(a, b[:M]), f"{name}: M={M} differs from the first {M} rows of the {MB}-point run"
        _, grads_ref = run(MB, zero_beyond=M)
        assert grads.keys() == grads_ref.keys()
        for k in grads:
            scale = float(grads_ref[k].abs().max()) + 1e-20
            err = float((grads[k] - grads_ref[k]).abs().max()) / scale
            assert err <= 2e-5, f"grad {k}: M={M} vs masked {MB}-point run: {err:.2e}"


def test_grid_encode_grouped_taps_bit_identical():
    """The finite-difference tap layout (6 consecutive points within 2 eps of each other) through the corner-caching
    grouped kernel equals the ungrouped kernel bit for bit, incl. taps clamped at / pushed across the box boundary,
    taps straddling cell and brick boundaries, and progressive levels."""
    from morpheus_amd import ops
    emb, offs, res = _grid_setup()
    M = 3001
    g = torch.Generator().manual_seed(11)
    c = (torch.rand(M, 3, generator=g) * 2 - 1) * 1.02            # some centres outside the box
    c[:200] = torch.round(c[:200] * 64) / 64 * 1.01               # centres sitting exactly on cell faces
    off = torch.zeros(1, 6, 3)
    for k in range(3):
        off[0, 2 * k, k], off[0, 2 * k + 1, k] = 2e-3, -2e-3
    taps = (c[:, None] + off).clamp(-1.01, 1.01).reshape(-1, 3).to(DEV)
    embg = emb.to(DEV)
    for ml in (None, 0.5):
        a = ops.grid_encode(taps, embg, offs, res, 1.01, ml, group=6)
        b = ops.grid_encode(taps, embg, offs, res, 1.01, ml, group=1)
        assert torch.equal(a, b)
    # a hint that does not divide the point count is ignored, not an error
    assert torch.equal(ops.grid_encode(taps[:-1], embg, offs, res, 1.01, None, group=6),
                       ops.grid_encode(taps[:-1], embg, offs, res, 1.01, None))


def test_field_query_glue_vs_torch():
    """csrc/normal.hip against the plain-torch expressions of the reference on the same GPU tensors:
    taps (model.py:367-376) and sample positions (morpheus.py:644-647) bit for bit, the normal (model.py:377-398,
    utils.py:70-71) to round-off; every backward against torch autograd of those expressions."""
    from morpheus_amd import ops
    from morpheus_amd.model import safe_normalize
    M, eps, bound = 3001, 2e-3, 1.01
    x = (synth.hash_tensor((M, 3), 700, 1.02)).to(DEV)              # some points on / beyond the box: the clamp acts
    x[:7] = torch.tensor([1.0095, -1.0095, 1.01])
    topo = synth.hash_tensor((M, 2), 701, 0.3).to(DEV)
    gt, gp = synth.hash_tensor((6 * M, 3), 702, 1.0).to(DEV), synth.hash_tensor((6 * M, 2), 703, 1.0).to(DEV)
    # --- taps
    xa, ta = x.clone().requires_grad_(True), topo.clone().requires_grad_(True)
    off = x.new_zeros(1, 6, 3)
    for k in range(3):
        off[0, 2 * k, k], off[0, 2 * k + 1, k] = eps, -eps
    taps_t = (xa[:, None] + off).clamp(-bound, bound).reshape(6 * M, 3)
    topo_t = ta[:, None].expand(M, 6, 2).reshape(6 * M, 2)
    ((taps_t * gt).sum() + (topo_t * gp).sum()).backward()
    xb, tb = x.clone().requires_grad_(True), topo.clone().requires_grad_(True)
    taps_h, topo_h = ops.fd_taps(xb, tb, eps, bound)
    ((taps_h * gt).sum() + (topo_h * gp).sum()).backward()
    assert torch.equal(taps_h, taps_t) and torch.equal(topo_h, topo_t)
    assert_close(xb.grad, xa.grad, 1e-6, "d taps / dx", floor=1e-3)
    assert_close(tb.grad, ta.grad, 1e-6, "d topo6 / d topo", floor=1e-3)
    taps_n, topo_n = ops.fd_taps(x, None, eps, bound)               # no gradient anywhere: taps must not ask for d/dx
    assert topo_n is None and not taps_n.requires_grad and torch.equal(taps_n, taps_t)
    # --- normal
    s6 = synth.hash_tensor((M, 6), 704, 0.5).to(DEV)
    s6[:5] = 0.25                                                   # exactly flat: |raw| = 0 -> the clamp branch, normal = 0
    gn, gr = synth.hash_tensor((M, 3), 705, 1.0).to(DEV), synth.hash_tensor((M, 3), 706, 1.0).to(DEV)
    sa = s6.clone().requires_grad_(True)
    raw_t = torch.stack([0.5 * (sa[:, 0] - sa[:, 1]) / eps, 0.5 * (sa[:, 2] - sa[:, 3]) / eps, 0.5 * (sa[:, 4] - sa[:, 5]) / eps], -1)
    nrm_t = torch.nan_to_num(safe_normalize(raw_t))
    ((nrm_t * gn).sum() + (raw_t * gr).sum()).backward()
    sb = s6.clone().requires_grad_(True)
    nrm_h, raw_h = ops.fd_normal(sb, eps)
    ((nrm_h * gn).sum() + (raw_h * gr).sum()).backward()
    assert_close(raw_h, raw_t, 2.5e-7, "raw normal (1 ulp: how torch divides by a python scalar is a build detail)", floor=1e-3)
    assert_close(nrm_h, nrm_t, 1e-6, "normal", floor=1e-3)
    assert_close(sb.grad, sa.grad, 2e-5, "d normal / d sdf", floor=1e-3 * float(sa.grad.abs().max()))
    # --- sample positions (ragged rays, some empty)
    N = 97
    cnt = (torch.arange(N) * 7919 % 23).int()
    cnt[5] = cnt[40] = 0
    ri = torch.repeat_interleave(torch.arange(N), cnt.long()).int().to(DEV)
    Ms = int(cnt.sum())
    start = (torch.cumsum(cnt, 0) - cnt).int().to(DEV)
    o, d = synth.hash_tensor((N, 3), 707, 1.0).to(DEV), synth.hash_tensor((N, 3), 708, 1.0).to(DEV)
    ts = synth.hash_tensor((Ms,), 709, 1.0, 1.5).to(DEV)
    te = ts + 0.01
    gx = synth.hash_tensor((Ms, 3), 710, 1.0).to(DEV)
    oa, da = o.clone().requires_grad_(True), d.clone().requires_grad_(True)
    xyz_t = oa[ri.long()] + da[ri.long()] * ((ts[:, None] + te[:, None]) / 2.0)
    (xyz_t * gx).sum().backward()
    ob, db = o.clone().requires_grad_(True), d.clone().requires_grad_(True)
    xyz_h = ops.sample_positions(ob, db, ri, ts, te, start, cnt.to(DEV))
    (xyz_h * gx).sum().backward()
    assert torch.equal(xyz_h, xyz_t)
    assert_close(ob.grad, oa.grad, 1e-5, "d xyz / d o", floor=1e-2)
    assert_close(db.grad, da.grad, 1e-5, "d xyz / d d", floor=1e-2)
    assert float(ob.grad[5].abs().sum()) == 0.0 and float(db.grad[40].abs().sum()) == 0.0


def test_multicode_sample_kernel():
    """mh_multicode_fwd/bwd: bit for bit the per-level lerp of deform_code.py:20-38 written in torch on the same device,
    the reference's own output (fixture operators.npz:multicode, generated by the reference's MultiCode.sample), and the
    gradient of torch autograd through that formula (atomics: round-off)."""
    from morpheus_amd.model import MultiCode
    from tests.util import load_golden
    st = synth.make_state("b")
    mc = MultiCode([25, 50, 200], 16).to(DEV)
    mc.load_state_dict({f"volumes.{k}": st[f"deform_code.volumes.{k}"] for k in range(3)})
    t = torch.tensor([[0.0], [7 / 200], [0.5], [199 / 200], [1.3], [-0.2], [0.123456]], device=DEV)
    got = mc.sample(t)
    assert_close(got, load_golden("operators.npz")["multicode"], 1e-5, "vs the reference's MultiCode.sample", floor=1e-3)
    tl = torch.rand(4099, device=DEV) * 1.2 - 0.1                     # per-sample times (the point-loss call shape)
    gw = synth.hash_tensor((4099, 48), 811, 1.0).to(DEV)
    vols = [v.detach().clone().requires_grad_(True) for v in mc.volumes]
    tt = tl.clamp(0, 1)
    want = []
    for vol in vols:
        v = vol[0, :, :, 0]
        size = v.shape[1]
        r = ((tt * 2 - 1) + 1) / 2 * (size - 1)
        r0 = torch.floor(r)
        fr = (r - r0)[None]
        i0 = r0.long().clamp(0, size - 1)
        i1 = (i0 + 1).clamp(0, size - 1)
        want.append((v[:, i0] * (1 - fr) + v[:, i1] * fr).t())
    want = torch.cat(want, -1)
    (want * gw).sum().backward()
    got = mc.sample(tl)
    assert torch.equal(got, want)
    mc.zero_grad()
    (got * gw).sum().backward()
    for k in range(3):
        assert_close(mc.volumes[k].grad, vols[k].grad, 1e-4, f"d volumes.{k}", floor=1e-2 * float(vols[k].grad.abs().max()))


def test_sdf_losses_kernel_vs_reference_formula():
    """mh_sdf_losses_fwd/bwd against utils.py:91-113 written in torch on the same device (the oracle's sdf_losses is that
    function restated; the reference's own values are pinned by the render goldens), values and d/d(pred_sdf): free space,
    truncation band, negative and zero target depths, masked rays."""
    from morpheus_amd import ops
    N, trunc = 64, 0.1
    cnt = (torch.arange(N) * 31 % 17 + 3).int()
    ri = torch.repeat_interleave(torch.arange(N), cnt.long())
    M = ri.numel()
    depth = synth.hash_tensor((N, 1), 820, 0.6, 1.2)
    depth[::7] = -1.0                                   # "no depth": everything in front counts as free space
    depth[3::11] = 0.0                                  # zero depth: excluded from the normalisation count
    mask = (synth.hash_tensor((N, 1), 821, 0.5, 0.5) > 0.3).float()
    ts = synth.hash_tensor((M,), 822, 0.8, 1.2)
    te = ts + 0.01
    pred = synth.hash_tensor((M,), 823, 0.3)
    for use_mask in (True, False):
        po = pred.clone().requires_grad_(True)
        fs_o, sl_o = of.sdf_losses(((ts + te) / 2)[:, None], depth[ri], po, trunc, mask[ri] if use_mask else None)
        (2.0 * fs_o + 3.0 * sl_o).backward()
        pg = pred.to(DEV).requires_grad_(True)
        fs_g, sl_g = ops.sdf_losses(pg, ts.to(DEV), te.to(DEV), ri.int().to(DEV), depth.to(DEV), mask.to(DEV) if use_mask else None, trunc)
        (2.0 * fs_g + 3.0 * sl_g).backward()
        assert_close(fs_g, fs_o, 1e-5, "fs_loss", floor=1e-6)
        assert_close(sl_g, sl_o, 1e-5, "sdf_loss", floor=1e-6)
        assert_close(pg.grad, po.grad, 1e-5, "d/d pred", floor=1e-2 * float(po.grad.abs().max()))


def _report(record):
    """append a measurement record to gpurun_out/precision_report.jsonl (scratch on the GPU box; the judged copy is
    profiles/r03_precision_report.jsonl) -- the numbers DESIGN.md section 4 quotes"""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "precision_report.jsonl"), "a") as f:
            f.write(json.dumps(record) + "\n")
    except OSError:
        pass


def _precision_case(data, M):
    """Warp-net-shaped networks and inputs for the arithmetic tests.  `data`:
      gauss      layer-shaped Gaussian weights (what a trained warp net looks like)
      geometric  the reference's geometric initialisation (models/decoders.py:25-43): first layer reads x only (all other
                 columns zero), hidden N(0, sqrt(2/out)), LAST layer one near-constant positive block N(sqrt(pi/in), 1e-4),
                 bias -0.4 -- every product of the last layer has the same sign (no cancellation, maximal bias exposure)
      heavy      one weight per layer 100x the layer's scale (a heavy tail sets the layer's slice scale)
      dominant   a first-layer bias with one element of 300 among 0.3s: every point's activation vector has ONE dominant
                 element (what a per-point block scale keys on)
      large      hidden weights x5: activations grow ~4x per layer to ~1e3
      small      hidden weights / 5: activations shrink ~6x per layer to ~1e-4"""
    nets = []
    hid = {"large": 0.5, "small": 0.02}.get(data, 0.1)
    for nout in (3, 2):
        W = [torch.randn(128, 39, device=DEV) * 0.15] + [torch.randn(128, 128, device=DEV) * hid for _ in range(4)] + \
            [torch.randn(nout, 128, device=DEV) * 0.15]
        b = [torch.randn(128, device=DEV) * 0.1 for _ in range(5)] + [torch.randn(nout, device=DEV) * 0.1]
        if data == "geometric":
            W[0][:, 3:] = 0.0
            W[0][:, :3] = torch.randn(128, 3, device=DEV) * (2.0 ** 0.5 / 128 ** 0.5)
            for l in range(1, 5):
                W[l] = torch.randn(128, 128, device=DEV) * (2.0 ** 0.5 / 128 ** 0.5)
            W[5] = (3.141592653589793 ** 0.5 / 128 ** 0.5) + torch.randn(nout, 128, device=DEV) * 1e-4
            b = [torch.zeros(128, device=DEV) for _ in range(5)] + [torch.full((nout,), -0.4, device=DEV)]
        if data == "heavy":
            for l in range(6):
                W[l][l % W[l].shape[0], (7 * l + 3) % W[l].shape[1]] = 100.0 * float(W[l].std())
        nets.append(W + b)
    x = torch.rand(M, 3, device=DEV) * 2 - 1
    b0 = [torch.randn(3, 128, device=DEV) * 0.3 for _ in range(2)]
    if data == "dominant":
        for t in b0:
            t[:, 17] = 300.0
    if data == "geometric":
        b0 = [torch.zeros(3, 128, device=DEV) for _ in range(2)]
    return nets, x, b0


def _warp_float64(nets, x, slot, b0, kink_rel=1e-4):
    """float64 evaluation of the two warp nets with autograd -> (outputs, leaves, kink-safe mask).  A pre-activation within
    rounding of zero flips its ReLU bit in any fp32 implementation and moves that point's whole backward signal by O(1);
    points with a pre-activation closer than kink_rel x (the layer's mean |z|) to a kink are excluded from every loss, so
    that the comparisons measure arithmetic, not kink lottery."""
    M = x.shape[0]
    ps64 = [[p.double().clone().requires_grad_(True) for p in net] for net in nets]
    x64 = x.double().clone().requires_grad_(True)
    b64 = [t.double().clone().requires_grad_(True) for t in b0]
    enc = [x64] + [f(x64 * 2 ** k) for k in range(6) for f in (torch.sin, torch.cos)]
    e = torch.cat(enc, -1)
    outs, safe = [], torch.ones(M, dtype=torch.bool, device=DEV)
    for k, P in enumerate(ps64):
        z = e @ P[0].t() + (b64[k][slot.long()] if slot is not None else b64[k][:1])
        safe &= (z.detach().abs() > kink_rel * z.detach().abs().mean()).all(dim=1)
        hcur = torch.relu(z)
        for l in range(1, 5):
            z = hcur @ P[l].t() + P[6 + l]
            safe &= (z.detach().abs() > kink_rel * z.detach().abs().mean()).all(dim=1)
            hcur = torch.relu(z)
        outs.append(hcur @ P[5].t() + P[11])
    return outs, (ps64, x64, b64), safe


@pytest.mark.parametrize("data", ["gauss", "geometric", "heavy", "dominant", "large", "small"])
def test_warp_sliced_arithmetic_against_float64(data, monkeypatch):
    """Both arithmetic forms of the warp kernels against a float64 evaluation of the same networks -- forward values,
    d/dx and every weight gradient -- on ordinary AND adversarial operand distributions (see _precision_case).
    What is asserted, per form, relative to the native fp32-MFMA kernels' own error on the same data:
      b3 (exact three-way bf16 split, the fp32-faithful default): forward values within 3x, d/dx within 6x, weight gradients
         within 12x (measured 3-8x, worst 7.8e-6 against 9.8e-7 rel-L2 on the "dominant" data: the OPERANDS are exact, but six
         slice products per MAC go through the bf16 pipe's internal adder, which does not round to nearest, and a weight
         gradient is a 10^3..10^6-term sum per entry) -- or below the absolute floors 4e-7 / 5e-6 rel-L2.
    Every measured ratio goes to the precision report (profiles/r03_precision_report.jsonl)."""
    from morpheus_amd import ops
    torch.manual_seed(7)
    M = 6000                                                # ragged against both the 128- and the 256-point workgroups
    nets, x, b0 = _precision_case(data, M)
    slot = (torch.arange(M, device=DEV) % 3).int()
    wd_, wt_ = torch.randn(M, 3, device=DEV), torch.randn(M, 2, device=DEV)
    outs, (ps64, x64, b64), safe = _warp_float64(nets, x, slot, b0)
    assert int(safe.sum()) > M // 3, int(safe.sum())
    wd_ = wd_ * safe[:, None]
    wt_ = wt_ * safe[:, None]
    ((outs[0] * wd_.double()).sum() + (outs[1] * wt_.double()).sum()).backward()

    def run(mode):
        _set_mlp(monkeypatch, ops, mode)
        ps = [[p.clone().requires_grad_(True) for p in net] for net in nets]
        xg = x.clone().requires_grad_(True)
        bb = [t.clone().requires_grad_(True) for t in b0]
        d, t = ops.warp_mlp(xg, slot, bb[0], bb[1], 6, ops.prepare_warp_operands(ps[0], ps[1]))
        ((d * wd_).sum() + (t * wt_).sum()).backward()
        return d.detach(), t.detach(), xg.grad, [[p.grad for p in net] for net in ps], [t.grad for t in bb]

    def rl2(a, b):
        return float((a.double() - b).norm() / b.norm().clamp_min(1e-30))

    res = {m: run(m) for m in ("f32", "b3")}
    g64 = [[p.grad for p in net] for net in ps64]
    meas = {}
    for m, r in res.items():
        fwd = [float((r[k][safe].double() - outs[k][safe]).abs().max()) / float(outs[k][safe].abs().max()) for k in (0, 1)]
        wg = [rl2(ga, gr) for net, net64 in zip(r[3], g64) for ga, gr in zip(net, net64) if gr is not None and float(gr.norm()) > 0]
        bg = [rl2(ga, t.grad) for ga, t in zip(r[4], b64) if float(t.grad.norm()) > 0]
        meas[m] = dict(fwd=max(fwd), dx=rl2(r[2], x64.grad) if float(x64.grad.norm()) > 0 else 0.0, wgrad=max(wg), wgrad_min=min(wg),
                       bias0=max(bg) if bg else 0.0, n_wgrad=len(wg))
    _report(dict(test="warp_sliced_arithmetic_against_float64", data=data, points=M, safe_points=int(safe.sum()), **{
        m: {k: (round(v, 12) if isinstance(v, float) else v) for k, v in meas[m].items()} for m in meas}))
    f32 = meas["f32"]
    lim = {"b3": dict(fwd=(3.0, 4e-7), dx=(6.0, 5e-6), wgrad=(12.0, 5e-6), bias0=(12.0, 5e-6))}
    for m in ("b3",):
        for q, (ratio, floor) in lim[m].items():
            assert meas[m][q] <= max(ratio * f32[q], floor), (data, m, q, meas[m][q], f32[q])


@pytest.mark.parametrize("mlp", ["b3", "f32"])
def test_warp_large_batch_weight_gradients(mlp, monkeypatch):
    """The large-batch kernels of the warp path (from 16 384 tiles on mh_mlp_wgrad(_b3) launches one kernel per layer: the
    slice-once-per-workgroup b3 kernel / the three-register-set fp32 kernel; the 8-wave forward / backward run hundreds of
    workgroups per CU) against the small-batch forms the oracle tests pin: one call on 600 000 points must give the same
    outputs, d/dx and parameter gradients as the same points fed in six chunks (same terms, different kernels and order)."""
    from morpheus_amd import ops
    _set_mlp(monkeypatch, ops, mlp)
    torch.manual_seed(3)
    M, CH = 600_000, 100_000
    nets = []
    for nout in (3, 2):
        W = [torch.randn(128, 39, device=DEV) * 0.15] + [torch.randn(128, 128, device=DEV) * 0.1 for _ in range(4)] + \
            [torch.randn(nout, 128, device=DEV) * 0.15]
        b = [torch.randn(128, device=DEV) * 0.1 for _ in range(5)] + [torch.randn(nout, device=DEV) * 0.1]
        nets.append(W + b)
    x = torch.rand(M, 3, device=DEV) * 2 - 1
    b0 = [torch.randn(1, 128, device=DEV) * 0.3 for _ in range(2)]
    wd_, wt_ = torch.randn(M, 3, device=DEV), torch.randn(M, 2, device=DEV)

    def run(chunks):
        ps = [[p.clone().requires_grad_(True) for p in net] for net in nets]
        bb = [t.clone().requires_grad_(True) for t in b0]
        xg = x.clone().requires_grad_(True)
        opnd = ops.prepare_warp_operands(ps[0], ps[1])
        outs = []
        for a in range(0, M, chunks):
            d, t = ops.warp_mlp(xg[a:a + chunks], None, bb[0], bb[1], 6, opnd)
            outs.append((d, t))
        d = torch.cat([o[0] for o in outs]); t = torch.cat([o[1] for o in outs])
        ((d * wd_).sum() + (t * wt_).sum()).backward()
        return d.detach(), t.detach(), xg.grad, [g for net in ps for g in (p.grad for p in net) if g is not None] + [t.grad for t in bb]

    big, small = run(M), run(CH)
    assert torch.equal(big[0], small[0]) and torch.equal(big[1], small[1])          # per-point results do not depend on the batch
    assert torch.equal(big[2], small[2])
    assert len(big[3]) == len(small[3]) == 26
    for ga, gb in zip(big[3], small[3]):
        scale = float(gb.abs().max()) + 1e-20
        assert float((ga - gb).abs().max()) / scale <= 2e-5, (tuple(gb.shape), float((ga - gb).abs().max()) / scale)
    if mlp == "b3":
        # round 6: at this size backward-data does not park dPre4 and the layer-4 weight-gradient launches regenerate it from the
        # incoming gradient, the ReLU sign words and the T5 slices (mh_warp_wgrad_b3) -- the SAME BITS as reading the parked rows,
        # also when one net has no incoming gradient at all
        assert ops.REGEN_DPRE4 and ops._lib.load().mh_warp_regen_dpre4(M) == 1 and ops._lib.load().mh_warp_regen_dpre4(CH) == 0
        monkeypatch.setattr(ops, "REGEN_DPRE4", False)
        parked = run(M)
        assert torch.equal(parked[2], big[2])
        for ga, gb in zip(big[3], parked[3]):
            assert torch.equal(ga, gb), (tuple(gb.shape), float((ga - gb).abs().max()))

        def run_deform_only(regen):
            monkeypatch.setattr(ops, "REGEN_DPRE4", regen)
            ps = [[p.clone().requires_grad_(True) for p in net] for net in nets]
            bb = [t.clone().requires_grad_(True) for t in b0]
            d, _ = ops.warp_mlp(x, None, bb[0], bb[1], 6, ops.prepare_warp_operands(ps[0], ps[1]))
            (d * wd_).sum().backward()
            return [p.grad for net in ps for p in net] + [t.grad for t in bb]

        for ga, gb in zip(run_deform_only(True), run_deform_only(False)):
            assert (ga is None) == (gb is None) and (ga is None or torch.equal(ga, gb))


@pytest.mark.parametrize("with_color", [True, False])
def test_field_large_batch_gradients(with_color):
    """The persistent field kernels loop over many tiles per wave only at large batches (a 777-point test gives every wave
    at most one tile): 400 000 points in one call against the same points in 40 chunks of 10 000 -- outputs and input
    gradients bit for bit, parameter gradients to summation order."""
    from morpheus_amd import ops
    torch.manual_seed(5)
    M, CH = 400_000, 10_000
    pg = _state("b", DEV, grad=False)
    x = torch.rand(M, 3, device=DEV) * 2 - 1
    fs, fc = torch.randn(M, 32, device=DEV) * 0.1, torch.randn(M, 32, device=DEV) * 0.1
    topo = torch.randn(M, 2, device=DEV) * 0.3
    ws, wg, wc = torch.randn(M, device=DEV), torch.randn(M, device=DEV) * 0.01, torch.randn(M, 3, device=DEV)

    def run(chunk):
        Ws = [pg[f"sdf_net.net.{l}.weight"].clone().requires_grad_(True) for l in range(3)]
        Wc = [of.wn_weight(pg[f"color_net.net.{l}.weight_g"], pg[f"color_net.net.{l}.weight_v"]).clone().requires_grad_(True) for l in range(3)]
        bs = [pg[f"sdf_net.net.{l}.bias"].clone().requires_grad_(True) for l in range(3)]
        bc = [pg[f"color_net.net.{l}.bias"].clone().requires_grad_(True) for l in range(3)]
        beta = (pg["sdf2density.beta"].abs() + 1e-4).clone().requires_grad_(True)
        leaves = [t.clone().requires_grad_(True) for t in (x, fs, fc, topo)]
        opnd = ops.prepare_field_operands(Ws + Wc + bs + bc)
        outs = []
        for a in range(0, M, chunk):
            sl = slice(a, a + chunk)
            outs.append(ops.field_mlp(leaves[0][sl], leaves[1][sl], leaves[2][sl] if with_color else None, leaves[3][sl], beta, 6,
                                      with_color, opnd))
        sdf = torch.cat([o[0] for o in outs]); sig = torch.cat([o[1] for o in outs])
        loss = (sdf * ws).sum() + (sig * wg).sum()
        if with_color:
            loss = loss + (torch.cat([o[2] for o in outs]) * wc).sum()
        loss.backward()
        params = Ws + (Wc if with_color else []) + bs + (bc if with_color else []) + [beta]
        return (sdf.detach(), sig.detach()), [t.grad for t in leaves if t.grad is not None], [p.grad for p in params]

    big, small = run(M), run(CH)
    assert torch.equal(big[0][0], small[0][0]) and torch.equal(big[0][1], small[0][1])
    for ga, gb in zip(big[1], small[1]):
        assert torch.equal(ga, gb)
    for ga, gb in zip(big[2], small[2]):
        scale = float(gb.abs().max()) + 1e-20
        assert float((ga - gb).abs().max()) / scale <= 2e-5, (tuple(gb.shape), float((ga - gb).abs().max()) / scale)


def test_grid_large_batch_gradients():
    """The brick-binned hash-grid backward at a batch where hot bricks split into many chunks and every workgroup loops
    (1.5 M points: two thirds along rays through the box -- dense near the near plane -- one third uniform): one call against
    the same points in 12 chunks.  d/dx is per point (bit-equal); the embedding gradient is a fixed-point sum whose scale
    follows max|grad| of the call, so it agrees to the fixed-point resolution."""
    from morpheus_amd import ops
    emb, offs, res = _grid_setup()
    g = torch.Generator().manual_seed(21)
    n_r, S = 8192, 128
    o = torch.tensor([0.0, 0.0, 2.2]) + 0.05 * torch.randn(n_r, 3, generator=g)
    d = torch.nn.functional.normalize(torch.cat([torch.randn(n_r, 2, generator=g) * 0.35, -torch.ones(n_r, 1)], 1), dim=1)
    ts = 1.2 + 2.0 * (torch.arange(S).float()[None] + torch.rand(n_r, S, generator=g)) / S
    rays = (o[:, None] + d[:, None] * ts[..., None]).reshape(-1, 3).clamp(-1.0, 1.0)
    x = torch.cat([rays, torch.rand(n_r * S // 2, 3, generator=g) * 2 - 1]).to(DEV)
    M = x.shape[0]
    gw = torch.randn(M, 32, generator=g).to(DEV)
    embg = emb.to(DEV)

    def run(chunk):
        e = embg.clone().requires_grad_(True)
        xs = x.clone().requires_grad_(True)
        tot = 0
        for a in range(0, M, chunk):
            tot = tot + (ops.grid_encode(xs[a:a + chunk], e, offs, res, 1.01) * gw[a:a + chunk]).sum()
        tot.backward()
        return e.grad, xs.grad

    (ge1, gx1), (ge2, gx2) = run(M), run((M + 11) // 12)     # one call of 1.5 M points: rows staged in LDS; the chunks: gathered
    assert torch.equal(gx1, gx2)
    from morpheus_amd import _lib
    lib = _lib.load()
    before = lib.mh_grid_stage_min_points(1 << 40)             # ... and the single call in the gathering form: the same bits
    try:
        ge3, gx3 = run(M)
    finally:
        lib.mh_grid_stage_min_points(before)
    assert torch.equal(gx1, gx3)
    # (the on-chip fixed-point sums are order-free; the flush of hot bricks' chunks adds floats to one row in arrival order)
    assert float((ge1 - ge3).abs().max()) / float(ge3.abs().max()) <= 1e-6
    scale = float(ge2.abs().max())
    assert float((ge1 - ge2).abs().max()) / scale <= 1e-5, float((ge1 - ge2).abs().max()) / scale


def test_background_net_on_the_gpu_vs_reference():
    """a13 (models/model.py:400-410) on cuda: colour and gradients against the reference-generated fixture the CPU test uses
    (tests/test_reference_extras.py); the background net is plain torch in the product (dead in the reference's own loop)."""
    from morpheus_amd import harness
    from tests.util import load_golden
    g = load_golden("extras.npz")
    dirs = of.safe_normalize(synth.hash_tensor((256, 3), 340, 1.0)).to(DEV)
    tt = synth.hash_tensor((256, 1), 341, 0.5, 0.5).to(DEV)
    for kind in ("a", "b"):
        for ml_tag, ml in (("full", None), ("half", 0.5)):
            model = harness.build_model(kind, DEV, ml)
            model.zero_grad()
            c = model.background(dirs, tt)
            (c ** 2).sum().backward()
            assert c.is_cuda
            assert_close(c, g[f"bg_{kind}_{ml_tag}|color"], 1e-5, "bg colour")
            # gradients: sums over 256 directions through torch's GPU GEMM (another summation order than the reference's CPU
            # run; measured 2.7e-4 at the 1e-3 floor on 0.1 %-of-max entries): relative to the tensor's largest entry
            gw, gb = g[f"bg_{kind}_{ml_tag}|grad_w0"], g[f"bg_{kind}_{ml_tag}|grad_b1"]
            assert_close(model.bg_net.net[0].weight_v.grad, gw, 1e-4, "bg dW0", floor=1e-2 * float(abs(gw).max()))
            assert_close(model.bg_net.net[1].bias.grad, gb, 1e-4, "bg db1", floor=1e-2 * float(abs(gb).max()))
