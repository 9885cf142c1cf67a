"""GPU parity: the whole render_rays path (HIP) against (i) the committed goldens generated from the
reference's own Python and (ii) the CPU oracle on the same seeded rays / samples / weights.

Bar (BASELINE.json north_star): rendered RGB / depth and SDF within 1e-4 relative, checked twice:
  * at SURVEY 8(d)'s floor, rel = |a-b| / max(|b|, 1e-3), as a COUNTED gate (tests/util.py: assert_close_counted): every
    element within 5e-4 and at most 0.2 % of the SDF samples (2 % of the rays for depth / opacity) above 1e-4 -- the
    exceedances are |reference| < ~1e-3 values carrying one or two fp32 ulps of the O(1) quantities that cancel to them;
  * strictly (no exceedance) at the relaxed floors 1e-2 for RGB / SDF / opacity (1 % of their O(1) range) and 5e-2 for
    depth (2 % of its [0, 2.6] range).
The reference's Laplace density 0.5 + 0.5*sign(s)*expm1(-|s|/beta) cancels catastrophically far from
the surface, so sigma itself (and anything dominated by far-field density, e.g. the depth of rays that
only graze the box) carries ~1e-4 relative libm noise between ANY two fp32 implementations.
"""
import pytest
import torch

from morpheus_amd import synth
from oracle import field as of
from tests.util import assert_close, assert_close_counted, assert_close_vs_f64, grad_digest_check, load_golden, max_rel

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL, FLOOR = 1e-4, 1e-2
DEPTH_FLOOR = 5e-2   # depth lives on [0, ~2.6]: 2% of its range


def _setup(kind, case, max_level=None, train=False, mlp_mode=None):
    from morpheus_amd import harness
    hw, S, nray = {"cfg1": (32, 64, None), "cfg3head": (128, 128, 256)}[case]
    o, d, t, rid = synth.frame_rays(25, hw, hw)
    if nray is not None:
        o, d, t, rid = o[:, :nray], d[:, :nray], t[:, :nray], rid[:, :nray]
    N = o.shape[1]
    jit = synth.ray_jitter(N)
    model = harness.build_model(kind, DEV, max_level)
    model.mlp_mode = mlp_mode          # None: the process default (b3); "f32": native fp32 MFMA in every MLP kernel
    model.train(train)
    # samples are inputs of the parity runs: the goldens were rendered with the oracle's
    # uniform samples of the UN-corrected rays, so feed exactly those (the HIP sampler is checked
    # bit-exact against the same oracle function in test_gpu_ops.py)
    smp = of.uniform_samples(o[0], d[0], jit, S, 1.01)
    rend = harness.make_renderer(model, S, samples=tuple(v.to(DEV) for v in smp))
    light = of.safe_normalize(o[0] + torch.tensor([0.3, -0.2, 0.5]))
    return model, rend, (o, d, t, rid), light, hw, S, N, jit


MODES = ("eval_albedo_deform", "eval_albedo_cano", "eval_lambertian_deform", "train_albedo_deform_pose")


# the default arithmetic (b3: exact 3 x bf16 operand split) on every case; the caveat-free native-fp32-MFMA mode ("f32") on cfg1, under
# the same gates -- both are fp32-faithful, and the driver's GPU run sees both
@pytest.mark.parametrize("kind,case,mlp_mode", [("a", "cfg1", None), ("b", "cfg1", None), ("a", "cfg3head", None), ("b", "cfg3head", None),
                                                ("a", "cfg1", "f32"), ("b", "cfg1", "f32")])
def test_render_rays_vs_reference_goldens(kind, case, mlp_mode):
    g = load_golden("render.npz")
    g4 = load_golden("round4.npz")
    for mode in MODES:
        train = mode.startswith("train")
        model, rend, rays, light, hw, S, N, jit = _setup(kind, case, train=train, mlp_mode=mlp_mode)
        cfg = rend.config
        cfg["train"]["normal_smooth_3d"] = 0.0     # randomised regularisers are compared statistically elsewhere
        cfg["train"]["normal_smoothness"] = 0.0
        o, d, t, rid = [v.to(DEV) for v in rays]
        kw = {}
        if train:
            kw = dict(rays_depth=synth.hash_tensor((1, N, 1), 400, 0.3, 1.5).to(DEV),
                      rays_mask=(synth.hash_tensor((1, N, 1), 401, 0.5, 0.5) > 0.3).float().to(DEV),
                      optimize_pose=True, real_view=False)
        res = rend.render_rays(o, d, t, rid, hw, hw, ambient_ratio=0.3, light_d=light.to(DEV),
                               shading="lambertian" if "lambertian" in mode else "albedo", cano="cano" in mode, **kw)
        key = f"{kind}_{case}_{mode}"
        lam = "lambertian" in mode     # lambertian colour goes through FD normals: x250 round-off gain
        assert_close(res["image"], g[key + "|image"], 3e-3 if lam else TOL, key + " image", floor=FLOOR)
        assert_close(res["depth"], g[key + "|depth"], TOL, key + " depth", floor=DEPTH_FLOOR)
        assert_close(res["weights_sum"], g[key + "|weights_sum"], TOL, key + " opacity", floor=FLOOR)
        assert_close(res["sdf"][::16], g[key + "|sdf_s16"], TOL, key + " sdf", floor=FLOOR)
        if not lam:
            # the contract floor (1e-3).  Rendered RGB / depth / opacity: STRICT 1e-4 (round 4: the Laplace density and the
            # compositor's alpha are evaluated without their cancelling subtractions -- csrc/mlp_dev.h: laplace_unit,
            # csrc/composite.hip -- measured <= 5.6e-5, all of it the reference's own fp32 error against its double run); SDF
            # per sample: counted, every element within 3e-4 (measured 1.6e-4; the reference's own fp32 SDF is up to 4.6e-4 off
            # its double value on these samples: tests/test_oracle_golden.py)
            assert_close_counted(res["sdf"][::16], g[key + "|sdf_s16"], key + " sdf @1e-3")
            strict = 1e-4 if not train else 2e-4      # the pose-optimising training render moves the rays by an ulp
            assert_close(res["image"], g[key + "|image"], strict, key + " image @1e-3", floor=1e-3)
            assert_close(res["depth"], g[key + "|depth"], strict, key + " depth @1e-3", floor=1e-3)
            assert_close(res["weights_sum"], g[key + "|weights_sum"], strict, key + " opacity @1e-3", floor=1e-3)
        if mode in ("eval_albedo_deform", "eval_albedo_cano"):
            # ... and against the reference run in DOUBLE on the same inputs, with the allowance derived from the reference's own
            # fp32 error at each output (tests/util.py: assert_close_vs_f64): the counted gate's numbers are not fitted to this path
            for out, gk in ((res["sdf"][::16], "sdf_s16"), (res["image"], "image"), (res["depth"], "depth"),
                            (res["weights_sum"], "weights_sum")):
                assert_close_vs_f64(out, g[key + "|" + gk], g4[key + "|f64|" + gk], key + " " + gk + " vs float64")
        assert_close(res["weights"][::16], g[key + "|weights_s16"], 5e-4, key + " weights", floor=1e-3)
        if res["deform"] is not None:
            assert_close(res["deform"][::16], g[key + "|deform_s16"], TOL, key + " deform", floor=1e-3)
        for lk in ("loss_code", "sdf_loss", "fs_loss"):
            if key + "|" + lk in g.files:
                assert_close(res[lk], g[key + "|" + lk], TOL, key + " " + lk, floor=1e-3)
        timg, tdep = synth.targets(N)
        loss = ((res["image"][0] - timg.to(DEV)) ** 2).mean() + ((res["depth"][0] - tdep.to(DEV)) ** 2).mean()
        if train:
            loss = loss + res["loss_code"] + res["sdf_loss"] + 0.1 * res["fs_loss"]
        model.zero_grad()
        loss.backward()
        assert_close(loss, g[key + "|loss"], TOL, key + " loss", floor=1e-3)
        n_ok = grad_digest_check({k: p.grad for k, p in model.named_parameters() if p.grad is not None}, g, key,
                                 3e-3 if lam else 5e-4)
        assert n_ok >= 10, n_ok


@pytest.mark.parametrize("kind", ["a", "b"])
def test_model_entry_points_vs_reference_goldens(kind):
    """forward() in all shading modes, density(), normal(), warp() on 2048 probe points incl. OOB."""
    from morpheus_amd import harness
    g = load_golden("model.npz")
    n = 2048
    x = synth.hash_tensor((n, 3), 330, 1.15).to(DEV)
    t = torch.full((n, 1), 37 / 200, device=DEV)
    light = of.safe_normalize(synth.hash_tensor((n, 3), 331, 1.0)).to(DEV)
    for ml_tag, ml in (("full", None), ("half", 0.5)):
        model = harness.build_model(kind, DEV, ml).eval()
        for shading in ("albedo", "lambertian", "textureless", "normal"):
            for cano in (False, True):
                if ml is not None and shading in ("textureless", "normal"):
                    continue
                model.zero_grad()
                sdf, sig, col, nrm, dfm, raw = model(x, t, light, ratio=0.3, shading=shading, cano=cano)
                key = f"{kind}_{ml_tag}_{shading}_{'cano' if cano else 'deform'}"
                assert_close(sdf, g[key + "|sdf"], TOL, key + " sdf", floor=FLOOR)
                assert_close(sig, g[key + "|sigma"], 1e-3, key + " sigma (x10 gain on sdf round-off)", floor=FLOOR)
                assert_close(col, g[key + "|color"], TOL if shading == "albedo" else 5e-3, key + " color", floor=FLOOR)
                if nrm is not None:
                    assert_close(raw, g[key + "|normal_raw"], 5e-3, key + " normal_raw (FD)", floor=5e-2)
                if dfm is not None:
                    assert_close(dfm, g[key + "|deform"], TOL, key + " deform", floor=1e-3)
                if shading in ("albedo", "lambertian") and ml is None:
                    probe = (col ** 2).sum() + 0.01 * (sig ** 2).mean() + (sdf ** 2).sum()
                    probe.backward()
                    # d(hash)/dx is piecewise constant: a probe point whose warped position differs by 1e-7
                    # between two fp32 implementations can change cell at some level and flip its
                    # contribution, so with only 2048 points the deform-branch gradients agree to ~2e-3
                    # (relative L2; tests/parity_report.py); the canonical branch and the 65k-point
                    # render cases agree to <1e-4.
                    gtol = 5e-4 if (shading == "albedo" and cano) else 5e-3
                    n_ok = grad_digest_check({k: p.grad for k, p in model.named_parameters() if p.grad is not None},
                                             g, key, gtol)
                    assert n_ok >= 10
        with torch.no_grad():
            dd = model.density(x, t)
            assert_close(dd["sdf"], g[f"{kind}_{ml_tag}_density|sdf"], TOL, "density sdf", floor=FLOOR)
            assert_close(dd["albedo"], g[f"{kind}_{ml_tag}_density|albedo"], TOL, "density albedo", floor=FLOOR)
            assert_close(model.normal(x, t)[1], g[f"{kind}_{ml_tag}_normal_warped|raw"], 5e-3, "normal raw", floor=5e-2)
            assert_close(model.warp(x, t)[1], g[f"{kind}_{ml_tag}_warp|topo"], TOL, "topo", floor=1e-3)


def test_full_size_properties():
    """BASELINE full size (16384 rays x 128 samples, deform on): size-independent properties --
    opacity in [0,1] and equal to the sum of weights, image = colour + (1-opacity)*bg, compositor
    linearity in the colour, first 256 rays equal to the committed cfg3head golden, and a
    deterministic forward (no atomics on the forward path)."""
    from morpheus_amd import harness, ops
    g = load_golden("render.npz")
    model = harness.build_model("b", DEV).eval()
    o, d, t, rid = [v.to(DEV) for v in synth.frame_rays(25, 128, 128)]
    N, S = o.shape[1], 128
    jit = synth.ray_jitter(N).to(DEV)
    rend = harness.make_renderer(model, S, jitter=jit)
    light = of.safe_normalize(o[0].cpu() + torch.tensor([0.3, -0.2, 0.5])).to(DEV)
    with torch.no_grad():
        res = rend.render_rays(o, d, t, rid, 128, 128, ambient_ratio=0.3, light_d=light, shading="albedo")
        res2 = rend.render_rays(o, d, t, rid, 128, 128, ambient_ratio=0.3, light_d=light, shading="albedo")
    assert torch.equal(res["image"], res2["image"]) and torch.equal(res["sdf"], res2["sdf"])
    op = res["weights_sum"][:, 0]
    assert float(op.min()) >= 0.0 and float(op.max()) <= 1.0 + 1e-5
    assert_close(res["weights"].view(N, S).sum(-1), op, 1e-5, "sum of weights", floor=1e-3)
    key = "b_cfg3head_eval_albedo_deform"
    assert_close(res["image"][0, :256], g[key + "|image"][0], TOL, "first 256 rays image", floor=FLOOR)
    assert_close(res["depth"][0, :256], g[key + "|depth"][0], TOL, "first 256 rays depth", floor=DEPTH_FLOOR)
    # compositor linearity: C(a*rgb1 + b*rgb2) = a*C(rgb1) + b*C(rgb2)
    rs, rc = rend.occupancy_grid.packed
    ri, ts, te = rend.occupancy_grid.sampling(o[0], d[0])
    M = ts.shape[0]
    sig = torch.rand(M, device=DEV) * 30
    c1, c2 = torch.rand(M, 3, device=DEV), torch.rand(M, 3, device=DEV)
    f = lambda c: ops.composite(sig, ts, te, c, rs, rc)[3]
    assert_close(f(0.3 * c1 + 0.7 * c2), 0.3 * f(c1) + 0.7 * f(c2), 1e-5, "linearity", floor=1e-3)


@pytest.mark.parametrize("mlp_mode", [None, "f32"])
def test_full_size_forward_backward_equals_chunked_renders(mlp_mode):
    """BASELINE full size WITH gradients (16 384 rays x 128 samples, deform on, the bench's loss): the one-call render and its
    backward -- the large-batch kernels: per-layer weight-gradient launches, persistent field kernels looping over hundreds
    of tiles, brick-binned hash backward with hot bricks in many chunks -- against the SAME rays rendered as 64 calls of 256
    rays whose gradients accumulate (the small-batch forms the oracle / golden tests pin), and the first chunk against the
    committed reference golden (b_cfg3head).  Same terms, different kernels and summation orders."""
    from morpheus_amd import harness
    g = load_golden("render.npz")
    o, d, t, rid = [v.to(DEV) for v in synth.frame_rays(25, 128, 128)]
    N, S, CH = o.shape[1], 128, 256
    jit = synth.ray_jitter(N).to(DEV)
    timg, tdep = [v.to(DEV) for v in synth.targets(N)]
    light = of.safe_normalize(o[0].cpu() + torch.tensor([0.3, -0.2, 0.5])).to(DEV)

    def loss_of(res, a, b):            # the bench's MSE(image) + MSE(depth), normalised by the FULL ray count so that chunks add up
        return (((res["image"][0] - timg[a:b]) ** 2).sum() / (3 * N)) + (((res["depth"][0] - tdep[a:b]) ** 2).sum() / N)

    def run(chunk):
        model = harness.build_model("b", DEV).train()
        model.mlp_mode = mlp_mode                      # None: the default (b3); "f32": native fp32 MFMA
        for k in ("normal_smoothness", "normal_smooth_3d", "code_reg", "ori_weight"):
            model.config["train"][k] = 0.0
        total, first = 0.0, None
        for a in range(0, N, chunk):
            b = min(a + chunk, N)
            rend = harness.make_renderer(model, S, jitter=jit[a:b])
            res = rend.render_rays(o[:, a:b], d[:, a:b], t[:, a:b], rid[:, a:b], 128, 128, ambient_ratio=1.0, light_d=light[a:b],
                                   shading="albedo")
            l = loss_of(res, a, b)
            l.backward()
            total += float(l)
            if first is None:
                first = dict(image=res["image"].detach()[:, :CH], depth=res["depth"].detach()[:, :CH], sdf=res["sdf"].detach()[:CH * S])
        return total, first, {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}

    loss_full, first_full, g_full = run(N)
    loss_chunk, first_chunk, g_chunk = run(CH)
    key = "b_cfg3head_eval_albedo_deform"
    for first, tag in ((first_full, "one call"), (first_chunk, "first chunk")):
        assert_close(first["image"], g[key + "|image"], TOL, tag + " image vs golden", floor=FLOOR)
        assert_close(first["depth"], g[key + "|depth"], TOL, tag + " depth vs golden", floor=DEPTH_FLOOR)
        assert_close(first["sdf"][::16], g[key + "|sdf_s16"], TOL, tag + " sdf vs golden", floor=FLOOR)
    assert torch.equal(first_full["sdf"], first_chunk["sdf"])          # forward values do not depend on the batch size
    assert abs(loss_full - loss_chunk) <= 1e-5 * abs(loss_chunk), (loss_full, loss_chunk)
    assert set(g_full) == set(g_chunk) and len(g_full) >= 50       # every parameter of the render path (pose / background excluded)
    worst = {}
    for k, gc in g_chunk.items():
        gf = g_full[k]
        rel = float((gf.double() - gc.double()).norm() / gc.double().norm().clamp_min(1e-30))
        worst[k] = rel
        # embedding gradients: fixed-point bricks vs atomics-free small batches; MLP gradients: fp32 sums of 2 M terms in two orders
        assert rel <= 2e-4, (k, rel)
    assert max(worst.values()) > 0.0      # the two runs really took different kernels / orders


def test_render_with_occupancy_marcher_ragged():
    """The reference's call shape end to end: OccupancyGrid.sampling (ragged, fixed step 0.01) -> render_rays.
    HIP vs the CPU oracle on the oracle-marched samples (bit-identical to the HIP marcher's), values and gradients."""
    from morpheus_amd import harness
    from morpheus_amd.occgrid import OccupancyGrid
    from morpheus_amd.render import HotPathRenderer
    hw = 24
    o, d, t, rid = synth.frame_rays(25, hw, hw)
    N = o.shape[1]
    jit = synth.ray_jitter(N)
    c = (torch.arange(128).float() + 0.5) / 128 * 2.02 - 1.01
    X, Y, Z = torch.meshgrid(c, c, c, indexing="ij")
    ball = ((X ** 2 + Y ** 2 + Z ** 2).sqrt() < 0.7).to(torch.uint8).contiguous()
    light = of.safe_normalize(o[0] + torch.tensor([0.3, -0.2, 0.5]))
    timg, tdep = synth.targets(N)
    # oracle
    p = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in synth.make_state("b").items()}
    f = of.OracleField(p, 1.01, None)
    smp = of.march_samples(o[0], d[0], jit, 0.01, 1.01, ball)
    assert smp[0].numel() > 20000 and len(torch.unique(torch.bincount(smp[0], minlength=N))) > 5      # genuinely ragged
    ro = of.render_rays(f, o, d, t, rid, smp, ambient_ratio=1.0, light_d=light, shading="albedo")
    (((ro["image"][0] - timg) ** 2).mean() + ((ro["depth"][0] - tdep) ** 2).mean()).backward()
    # HIP
    model = harness.build_model("b", DEV).eval()
    grid = OccupancyGrid([-1.01] * 3 + [1.01] * 3, 128).to(DEV)
    grid.set_binary(ball.to(DEV))
    grid.fixed_jitter = jit.to(DEV)
    rend = HotPathRenderer(model, model.config, grid, 200)
    rg = rend.render_rays(o.to(DEV), d.to(DEV), t.to(DEV), rid.to(DEV), hw, hw, ambient_ratio=1.0, light_d=light.to(DEV),
                          shading="albedo")
    assert rg["sdf"].shape[0] == smp[0].numel()
    (((rg["image"][0] - timg.to(DEV)) ** 2).mean() + ((rg["depth"][0] - tdep.to(DEV)) ** 2).mean()).backward()
    assert_close(rg["image"], ro["image"], TOL, "image", floor=FLOOR)
    assert_close(rg["depth"], ro["depth"], TOL, "depth", floor=DEPTH_FLOOR)
    assert_close(rg["sdf"], ro["sdf"], TOL, "sdf", floor=FLOOR)
    assert_close(rg["weights_sum"], ro["weights_sum"], TOL, "opacity", floor=FLOOR)
    worst = 0.0
    for k, prm in model.named_parameters():
        if prm.grad is not None and p[k].grad is not None and float(p[k].grad.norm()) > 0:
            worst = max(worst, float((prm.grad.cpu().double() - p[k].grad.double()).norm() / p[k].grad.double().norm()))
    assert worst < 5e-4, worst


def test_training_regularisers_wiring():
    """The in-render regularisers of morpheus.py:708-792 on the HIP path.  They draw random perturbations, so the
    comparison with the oracle pins the perturbation to zero (smoothness_std = 0, topo_none = True):
      loss_normal_perturb = mean |normal(x; topo) - normal(x; topo=None)|   and   loss_orient, loss_code
    and then the default config (all regularisers on, random perturbations) must run, give finite losses and reach
    every parameter group with finite gradients."""
    from morpheus_amd import harness
    hw, S = 16, 32
    o, d, t, rid = synth.frame_rays(25, hw, hw)
    N = o.shape[1]
    jit = synth.ray_jitter(N)
    smp = of.uniform_samples(o[0], d[0], jit, S, 1.01)
    light = of.safe_normalize(o[0] + torch.tensor([0.3, -0.2, 0.5]))
    # --- oracle, zero perturbation
    p = {k: v for k, v in synth.make_state("b").items()}
    f = of.OracleField(p, 1.01, None)
    ri, ts, te = smp
    xyz = o[0][ri] + d[0][ri] * ((ts + te) / 2)[:, None]
    tstep = t[0][ri]
    sdf, sig, col, nrm, dfm, raw = f.forward(xyz, tstep, light[ri], ratio=0.3, shading="lambertian")
    nrm_none, _ = f.normal(xyz, topo=None)
    want_perturb = (nrm - nrm_none).abs().mean()
    w, _, _ = of.render_weights(ts, te, sig, ri, N)
    tdirs = of.safe_normalize(d[0][ri])
    want_orient = (w * (nrm * tdirs).sum(-1).clamp(min=0) ** 2).sum(-1).mean()
    # --- HIP
    model = harness.build_model("b", DEV).train()
    cfg = model.config
    cfg["train"]["smoothness_std"] = 0.0
    cfg["train"]["normal_smoothness"] = 0.0
    rend = harness.make_renderer(model, S, samples=tuple(v.to(DEV) for v in smp))
    res = rend.render_rays(o.to(DEV), d.to(DEV), t.to(DEV), rid.to(DEV), hw, hw, ambient_ratio=0.3, light_d=light.to(DEV),
                           shading="lambertian", real_view=False)
    assert_close(res["loss_normal_perturb"], want_perturb, 2e-2, "loss_normal_perturb (FD normals)", floor=1e-3)
    assert_close(res["loss_orient"], want_orient, 2e-2, "loss_orient", floor=1e-4)
    assert "loss_code" in res and "normal_reg" not in res
    # --- default config: everything on, random perturbations
    model = harness.build_model("b", DEV).train()
    rend = harness.make_renderer(model, S, samples=tuple(v.to(DEV) for v in smp))
    dep = synth.hash_tensor((1, N, 1), 400, 0.3, 1.5).to(DEV)
    res = rend.render_rays(o.to(DEV), d.to(DEV), t.to(DEV), rid.to(DEV), hw, hw, ambient_ratio=0.3, light_d=light.to(DEV),
                           shading="lambertian", real_view=True, rays_depth=dep, rays_mask=torch.ones_like(dep),
                           optimize_pose=True)
    keys = ("loss_normal_perturb", "loss_code", "normal_reg", "sdf_loss", "fs_loss")
    for k in keys:
        assert k in res and torch.isfinite(res[k]).all(), k
    total = (res["image"] ** 2).mean() + sum(res[k] for k in keys)
    total.backward()
    for name, prm in model.named_parameters():
        if "bg_net" in name:
            continue
        assert prm.grad is not None and torch.isfinite(prm.grad).all(), name


def test_two_frames_in_one_batch():
    """B = 2 frames flattened into one render_rays call (how cfg5 / a multi-frame batch reaches the path,
    morpheus.py:596-600): per-frame deform codes are selected per sample through the slot table, and their
    gradients are segment-summed per frame."""
    from morpheus_amd import harness
    hw, S = 16, 48
    fr = [synth.frame_rays(fid, hw, hw) for fid in (0, 25)]
    o, d, t, rid = [torch.cat([f[k] for f in fr], 0) for k in range(4)]          # [2, N, .]
    N = o.shape[1]
    jit = synth.ray_jitter(2 * N)
    smp = of.uniform_samples(o.reshape(-1, 3), d.reshape(-1, 3), jit, S, 1.01)
    light = of.safe_normalize(o.reshape(-1, 3) + torch.tensor([0.3, -0.2, 0.5]))
    timg, tdep = synth.targets(2 * N)
    p = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in synth.make_state("b").items()}
    f = of.OracleField(p, 1.01, None)
    ro = of.render_rays(f, o, d, t, rid, smp, ambient_ratio=1.0, light_d=light, shading="albedo")
    (((ro["image"].reshape(-1, 3) - timg) ** 2).mean() + ((ro["depth"].reshape(-1) - tdep) ** 2).mean()).backward()
    model = harness.build_model("b", DEV).eval()
    rend = harness.make_renderer(model, S, samples=tuple(v.to(DEV) for v in smp))
    rg = rend.render_rays(o.to(DEV), d.to(DEV), t.to(DEV), rid.to(DEV), hw, hw, ambient_ratio=1.0, light_d=light.to(DEV),
                          shading="albedo")
    (((rg["image"].reshape(-1, 3) - timg.to(DEV)) ** 2).mean() + ((rg["depth"].reshape(-1) - tdep.to(DEV)) ** 2).mean()).backward()
    assert rg["image"].shape == (2, N, 3) and rg["depth"].shape == (2, N)
    assert_close(rg["image"], ro["image"], TOL, "image", floor=FLOOR)
    assert_close(rg["depth"], ro["depth"], TOL, "depth", floor=DEPTH_FLOOR)
    assert_close(rg["deform"], ro["deform"], TOL, "deform", floor=1e-3)
    for k in ("deform_code.volumes.0", "deform_code.volumes.2", "deform_net.net.0.bias", "deform_net.net.0.weight_v",
              "topo_net.net.0.weight_g", "encoder.embeddings"):
        gg, go = dict(model.named_parameters())[k].grad.cpu().double(), p[k].grad.double()
        assert float((gg - go).norm() / go.norm()) < 5e-4, k


def test_training_steps_track_the_oracle():
    """End to end through everything a step touches: render fwd+bwd (HIP), weight-norm backward, gradient un-packing,
    the flat gradient bucket and mh_adam_step -- against the oracle rendered on the CPU and stepped by torch.optim.Adam
    with the reference's groups (lr, density lr/2, pose lr/10; morpheus.py:154-155, model.py:309-333).
    Adam turns a gradient into ~+-lr whatever its size, so individual near-zero-gradient entries may differ; the LOSS
    trajectory is the robust end-to-end observable: it must agree to 1e-3 relative over three steps and decrease."""
    from morpheus_amd import harness
    from morpheus_amd.optim import FlatAdam
    hw, S, lr = 16, 32, 2e-3
    o, d, t, rid = synth.frame_rays(25, hw, hw)
    N = o.shape[1]
    jit = synth.ray_jitter(N)
    light = of.safe_normalize(o[0] + torch.tensor([0.3, -0.2, 0.5]))
    timg, tdep = synth.targets(N)
    smp = of.uniform_samples(o[0], d[0], jit, S, 1.01)
    # HIP side
    model = harness.build_model("b", DEV).train()
    for k in ("normal_smoothness", "normal_smooth_3d", "code_reg", "ori_weight"):
        model.config["train"][k] = 0.0
    rend = harness.make_renderer(model, S, samples=tuple(v.to(DEV) for v in smp))
    opt = FlatAdam(model.get_params_all(lr), betas=(0.9, 0.99), eps=1e-15)
    gpu_args = [v.to(DEV) for v in (o, d, t, rid)]
    losses_hip = []
    for _ in range(3):
        opt.zero_grad()
        res = rend.render_rays(*gpu_args, hw, hw, ambient_ratio=1.0, light_d=light.to(DEV), shading="albedo")
        loss = harness.bench_loss(res, timg.to(DEV), tdep.to(DEV))
        loss.backward()
        opt.bucket.allreduce_mean()
        opt.step()
        losses_hip.append(float(loss.detach()))
    # oracle side
    p = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in synth.make_state("b").items()}
    scale = lambda k: 0.5 if k.startswith("sdf2density") else (0.1 if k.startswith("pose_array") else 1.0)
    ropt = torch.optim.Adam([{"params": [v], "lr": lr * scale(k)} for k, v in p.items() if v.is_floating_point()],
                            betas=(0.9, 0.99), eps=1e-15)
    losses_ref = []
    for _ in range(3):
        ropt.zero_grad()
        f = of.OracleField(p, 1.01, None)
        ro = of.render_rays(f, o, d, t, rid, smp, ambient_ratio=1.0, light_d=light, shading="albedo")
        lo = ((ro["image"][0] - timg) ** 2).mean() + ((ro["depth"][0] - tdep) ** 2).mean()
        lo.backward()
        ropt.step()
        losses_ref.append(float(lo.detach()))
    for a, b in zip(losses_hip, losses_ref):
        assert abs(a - b) <= 1e-3 * abs(b), (losses_hip, losses_ref)
    assert losses_hip[2] < losses_hip[0]


def test_real_view_step_vs_reference_golden():
    """The reference's real-view training call (morpheus.py:1191 with shading='albedo_normal', depth/mask supervision, pose
    optimisation, normal_smooth_3d / normal_smoothness / code_reg ON) and its three caller-side loss groups, against
    fixtures produced by the reference's OWN render_rays + loss methods with the random draws injected (the same
    closed-form draws are injected here): numeric values of loss_normal_perturb and normal_reg, not finiteness.
    FD normals amplify round-off x250, so quantities that pass through them are compared at 2e-2."""
    import numpy as np
    from morpheus_amd import harness
    from bench_support import trainstep
    from tests.util import DrawInjector
    g = load_golden("extras.npz")
    sel = torch.from_numpy(g["realview|sel"].astype(np.int64))
    hw, S = 32, 64
    o, d, t, rid = [v[:, sel] for v in synth.frame_rays(25, hw, hw)]
    N = o.shape[1]
    smp = of.uniform_samples(o[0], d[0], synth.ray_jitter(hw * hw)[sel], S, 1.01)
    model = harness.build_model("b", DEV, 0.75).train()
    rend = harness.make_renderer(model, S, samples=tuple(v.to(DEV) for v in smp))
    frame = trainstep.make_frames([25], hw, hw, DEV)[0]
    data = trainstep.sample_real_view_rays(frame, N, sel.to(DEV))
    bg = synth.hash_tensor((N, 3), 350, 0.5, 0.5).to(DEV)
    light = of.safe_normalize(o[0] + 0.3).to(DEV)
    with DrawInjector() as inj:
        res = rend.render_rays(o.to(DEV), d.to(DEV), t.to(DEV), rid.to(DEV), N, 1, bg_color=bg, ambient_ratio=1.0,
                               light_d=light, shading="albedo_normal", real_view=True, cano=False,
                               rays_depth=data["depth"].view(1, -1, 1), rays_mask=data["mask"].view(1, -1, 1),
                               optimize_pose=True)
        assert inj.k == int(g["realview|n_draws"]), "the HIP path must draw what the reference draws, in its order"
    assert_close(res["image"], g["realview|image"], TOL, "image", floor=FLOOR)
    assert_close(res["depth"], g["realview|depth"], TOL, "depth", floor=DEPTH_FLOOR)
    assert_close(res["weights_sum"], g["realview|weights_sum"], TOL, "opacity", floor=FLOOR)
    assert_close(res["sdf"][::8], g["realview|sdf_s8"], TOL, "sdf", floor=FLOOR)
    assert_close(res["normal"][::8], g["realview|normal_s8"], 2e-2, "normal (FD)", floor=5e-2)
    assert_close(res["loss_code"], g["realview|loss_code"], TOL, "loss_code")
    assert_close(res["sdf_loss"], g["realview|sdf_loss"], TOL, "sdf_loss")
    assert_close(res["fs_loss"], g["realview|fs_loss"], 1e-3, "fs_loss", floor=1e-4)
    assert_close(res["loss_normal_perturb"], g["realview|loss_normal_perturb"], 2e-2, "loss_normal_perturb")
    assert_close(res["normal_reg"], g["realview|normal_reg"], 2e-2, "normal_reg")
    B, H, W = 1, N, 1
    tr = model.config["train"]
    pred_depth, pred_mask = res["depth"].reshape(B, 1, H, W), res["weights_sum"].reshape(B, 1, H, W)
    pred_rgb = res["image"].reshape(B, H, W, 3).permute(0, 3, 1, 2).contiguous()
    gt_rgb, gt_depth, gt_mask = trainstep.get_gt_from_data(data, bg, B, H, W)
    l_render = trainstep.get_real_view_render_loss(tr, pred_rgb, pred_depth, pred_mask, gt_rgb, gt_depth, gt_mask,
                                                   data["rays_o"], data["rays_d"])
    # N surface points through model.density(x, t) with colour, x and t of equal length (morpheus.py:1013-1026)
    l_point = trainstep.get_real_view_point_loss(tr, model, gt_rgb, gt_depth, gt_mask, data["rays_o"], data["rays_d"],
                                                 data["rays_t"], res)
    l_reg = trainstep.get_regularization_loss(tr, model, res, None, 1000, 220000)
    # the same query with the time as an expanded scalar (what RealViewTrainStep passes for a one-row batch): one per-frame code
    # bias instead of N per-sample ones, the same value
    with model.operand_scope():
        l_point_sf = trainstep.get_real_view_point_loss(tr, model, gt_rgb, gt_depth, gt_mask, data["rays_o"], data["rays_d"],
                                                        data["rays_t"], res, single_frame=True)
    assert_close(l_point_sf, l_point, 2e-6, "point loss, single-frame time")
    assert_close(l_render, g["realview|loss_render"], TOL, "render loss")
    assert_close(l_point, g["realview|loss_point"], 2e-4, "point loss")
    assert_close(l_reg, g["realview|loss_reg"], 1e-2, "regularisation loss")
    model.zero_grad()
    (l_render + l_point + l_reg).backward()
    n_ok = grad_digest_check({k: p.grad for k, p in model.named_parameters() if p.grad is not None}, g, "realview", 3e-2)
    assert n_ok >= 40, n_ok
    for k in ("encoder.embeddings", "encoder_c.embeddings", "deform_code.volumes.2", "pose_array.data"):
        assert dict(model.named_parameters())[k].grad.abs().sum() > 0, k


def test_generate_rays_kernel_vs_reference_golden():
    """mh_generate_rays against the reference's get_camera_rays + c2w application (fixture), bit for bit."""
    import numpy as np
    from morpheus_amd import ops
    g = load_golden("extras.npz")
    for tag, (H, W) in (("sq", (24, 24)), ("rect", (20, 28))):
        for b, pose in enumerate((synth.look_at_pose(60.0, -40.0), synth.look_at_pose(75.0, 130.0, 1.3))):
            o, d = ops.generate_rays(np.float32(1.2 * W), np.float32(1.2 * W), 0.5 * W, 0.5 * H, pose, H, W, torch.device(DEV))
            assert np.array_equal(o.cpu().numpy().reshape(H, W, 3), g[f"raygen_{tag}|rays_o"][b]), (tag, b)
            assert np.array_equal(d.cpu().numpy().reshape(H, W, 3), g[f"raygen_{tag}|rays_d"][b]), (tag, b)


def test_fused_raygen_sampler_is_the_two_kernels_bit_for_bit():
    """mh_rays_sample_uniform (north_star's fused ray-generate + stratified sampler) against mh_generate_rays + the
    per-iteration pixel draw (dataset.py:412-423) + mh_sample_uniform, and -- through the first of those -- the
    reference-generated ray fixture; a random pixel subset with repeats, and the whole image."""
    import numpy as np
    from morpheus_amd import ops
    dev = torch.device(DEV)
    g = load_golden("extras.npz")
    H, W, S, bound = 20, 28, 24, 1.0
    pose = synth.look_at_pose(75.0, 130.0, 1.3)
    K = (np.float32(1.2 * W), np.float32(1.2 * W), 0.5 * W, 0.5 * H)
    o_all, d_all = ops.generate_rays(*K, pose, H, W, dev)
    gen = torch.Generator().manual_seed(7)
    cases = {"subset": torch.randint(0, H * W, (300,), generator=gen).to(torch.int32).to(dev), "image": None}
    for tag, pix in cases.items():
        N = H * W if pix is None else pix.shape[0]
        jit = torch.rand(N, generator=gen).to(dev)
        sel = slice(None) if pix is None else pix.long()
        o_ref, d_ref = o_all[sel].contiguous(), d_all[sel].contiguous()
        want = ops.sample_uniform(o_ref, d_ref, jit, S, bound, with_xyz=True)
        got = ops.rays_sample_uniform(*K, pose, H, W, pix, jit, S, bound, with_xyz=True)
        assert torch.equal(got[0], o_ref) and torch.equal(got[1], d_ref), tag
        for name, a, b in zip(("ray_idx", "t_starts", "t_ends", "xyz", "ray_start", "ray_cnt"), got[2:], want):
            assert torch.equal(a, b), (tag, name)
        assert float((got[4] - got[3]).min()) >= 0.0 and int(got[7].sum()) == N * S
    assert np.array_equal(got[1].cpu().numpy().reshape(H, W, 3), g["raygen_rect|rays_d"][1])
    no_xyz = ops.rays_sample_uniform(*K, pose, H, W, None, jit, S, bound)
    assert no_xyz[5] is None and torch.equal(no_xyz[3], got[3])
    with pytest.raises(ValueError):
        ops.rays_sample_uniform(*K, pose, H, W, cases["subset"].long(), jit[:300], S, bound)   # int64 pixels
    with pytest.raises(ValueError):
        ops.rays_sample_uniform(*K, pose, H, W, None, torch.rand(H * W + 1).to(dev), S, bound)


def test_full_size_canonical_properties():
    """BASELINE configs[1] at full size (16384 rays x 128 samples, cano=True: hash grids + sdf/colour nets + compositor):
    deterministic forward, opacity = sum of weights in [0,1], first 256 rays equal to the reference-generated golden,
    no deformation outputs, gradients reach both tables / both field nets and nothing on the warp side."""
    from morpheus_amd import harness
    g = load_golden("render.npz")
    model = harness.build_model("b", DEV).eval()
    o, d, t, rid = [v.to(DEV) for v in synth.frame_rays(25, 128, 128)]
    N, S = o.shape[1], 128
    rend = harness.make_renderer(model, S, jitter=synth.ray_jitter(N).to(DEV))
    light = of.safe_normalize(o[0].cpu() + torch.tensor([0.3, -0.2, 0.5])).to(DEV)
    with torch.no_grad():
        r2 = rend.render_rays(o, d, t, rid, 128, 128, ambient_ratio=0.3, light_d=light, shading="albedo", cano=True)
    res = rend.render_rays(o, d, t, rid, 128, 128, ambient_ratio=0.3, light_d=light, shading="albedo", cano=True)
    assert torch.equal(res["image"], r2["image"]) and torch.equal(res["sdf"], r2["sdf"])
    assert res["deform"] is None and res["normal"] is None
    op = res["weights_sum"][:, 0]
    assert float(op.min()) >= 0.0 and float(op.max()) <= 1.0 + 1e-5
    assert_close(res["weights"].view(N, S).sum(-1), op, 1e-5, "sum of weights", floor=1e-3)
    key = "b_cfg3head_eval_albedo_cano"
    assert_close(res["image"][0, :256], g[key + "|image"][0], TOL, "first 256 rays image", floor=FLOOR)
    assert_close(res["depth"][0, :256], g[key + "|depth"][0], TOL, "first 256 rays depth", floor=DEPTH_FLOOR)
    assert_close(res["weights_sum"][:256], g[key + "|weights_sum"], TOL, "first 256 rays opacity", floor=FLOOR)
    timg, tdep = [v.to(DEV) for v in synth.targets(N)]
    model.zero_grad()
    harness.bench_loss(res, timg, tdep).backward()
    grads = {k: p.grad for k, p in model.named_parameters()}
    for k in ("encoder.embeddings", "encoder_c.embeddings", "sdf_net.net.0.weight", "color_net.net.2.weight_v", "sdf2density.beta"):
        assert grads[k] is not None and torch.isfinite(grads[k]).all() and float(grads[k].abs().sum()) > 0, k
    for k in ("deform_net.net.0.weight_v", "topo_net.net.3.bias", "deform_code.volumes.0"):
        assert grads[k] is None or float(grads[k].abs().sum()) == 0.0, k


def test_normal_image_carries_the_density_gradient():
    """morpheus.py:775 accumulates the normal image with the LIVE weights: d(normal_image)/d(beta) is non-zero
    (beta only acts through the density)."""
    from morpheus_amd import harness
    hw, S = 12, 32
    o, d, t, rid = [v.to(DEV) for v in synth.frame_rays(25, hw, hw)]
    N = o.shape[1]
    model = harness.build_model("b", DEV).train()
    tr = model.config["train"]
    tr["normal_smooth_2d"], tr["normal_smoothness"], tr["normal_smooth_3d"] = 0.1, 0.0, 0.0
    rend = harness.make_renderer(model, S, jitter=synth.ray_jitter(N).to(DEV))
    light = of.safe_normalize(o[0].cpu() + 0.3).to(DEV)
    res = rend.render_rays(o, d, t, rid, hw, hw, ambient_ratio=0.3, light_d=light, shading="lambertian", real_view=False)
    assert res["normal_image"].shape == (N, 3)
    (gb,) = torch.autograd.grad(res["normal_image"].sum(), model.sdf2density.beta, retain_graph=True)
    assert float(gb.abs()) > 0


def test_eval_step_chunked_ragged_forward():
    """f-4: MorpheuS.eval_step's call shape (morpheus.py:1238-1269) -- a whole view in chunks, no_grad, ragged
    occupancy-marched samples -- equals the un-chunked render bit for bit (rays are independent), and the un-chunked
    render equals the CPU oracle on the oracle-marched samples."""
    from morpheus_amd import harness
    from morpheus_amd.occgrid import OccupancyGrid
    from morpheus_amd.render import HotPathRenderer
    hw = 40
    o, d, t, rid = synth.frame_rays(25, hw, hw)
    N = o.shape[1]
    c = (torch.arange(128).float() + 0.5) / 128 * 2.02 - 1.01
    X, Y, Z = torch.meshgrid(c, c, c, indexing="ij")
    ball = ((X ** 2 + Y ** 2 + Z ** 2).sqrt() < 0.7).to(torch.uint8).contiguous()
    model = harness.build_model("b", DEV).eval()
    grid = OccupancyGrid([-1.01] * 3 + [1.01] * 3, 128).to(DEV)
    grid.set_binary(ball.to(DEV))
    grid.fixed_jitter = 0.25
    rend = HotPathRenderer(model, model.config, grid, 200)
    data = dict(rays_o=o.to(DEV), rays_d=d.to(DEV), rays_t=t.to(DEV), rays_id=rid.to(DEV), H=hw, W=hw)
    with torch.no_grad():
        rgb_c, dep_c = rend.eval_step(data, max_chunk=500)           # 1600 rays -> 4 chunks of 401/401/401/397
        full = rend.render_rays(data["rays_o"], data["rays_d"], data["rays_t"], data["rays_id"], hw, hw, shading="albedo")
    assert rgb_c.shape == (1, hw, hw, 3) and dep_c.shape == (1, hw, hw)
    assert torch.equal(rgb_c.reshape(1, N, 3), full["image"]) and torch.equal(dep_c.reshape(1, N), full["depth"])
    smp = of.march_samples(o[0], d[0], torch.full((N,), 0.25), 0.01, 1.01, ball)
    f = of.OracleField({k: v for k, v in synth.make_state("b").items()}, 1.01, None)
    with torch.no_grad():
        ro = of.render_rays(f, o, d, t, rid, smp, ambient_ratio=1.0, light_d=of.safe_normalize(o[0] + 0.3), shading="albedo")
    assert_close(full["image"], ro["image"], TOL, "image", floor=FLOOR)
    assert_close(full["depth"], ro["depth"], TOL, "depth", floor=DEPTH_FLOOR)


def test_virtual_view_training_outputs_vs_reference_golden():
    """A virtual-view training call (lambertian shading, orientation loss, 2-D normal image accumulated with the live weights,
    morpheus.py:708-776) against the reference's own render_rays (fixture extras.npz:virt|*): outputs and the gradient of a
    loss that reaches the parameters THROUGH the normal image's weights."""
    from morpheus_amd import harness
    g = load_golden("extras.npz")
    hw, S = 16, 32
    o, d, t, rid = synth.frame_rays(25, hw, hw)
    N = o.shape[1]
    smp = of.uniform_samples(o[0], d[0], synth.ray_jitter(N), S, 1.01)
    light = of.safe_normalize(o[0] + torch.tensor([0.3, -0.2, 0.5]))
    model = harness.build_model("b", DEV).train()
    model.config["train"].update(normal_smooth_2d=0.1, normal_smooth_3d=0.0, normal_smoothness=0.0)
    rend = harness.make_renderer(model, S, samples=tuple(v.to(DEV) for v in smp))
    res = rend.render_rays(o.to(DEV), d.to(DEV), t.to(DEV), rid.to(DEV), hw, hw, bg_color=torch.tensor([0.2, 0.5, 0.7], device=DEV),
                           ambient_ratio=0.3, light_d=light.to(DEV), shading="lambertian", real_view=False, cano=False)
    assert_close(res["image"], g["virt|image"], 3e-3, "image (lambertian: FD normals)", floor=FLOOR)
    assert_close(res["normal_image"], g["virt|normal_image"], 3e-3, "normal image", floor=FLOOR)
    assert_close(res["loss_orient"], g["virt|loss_orient"], 1e-2, "loss_orient")
    assert_close(res["loss_code"], g["virt|loss_code"], TOL, "loss_code")
    wimg = synth.hash_tensor((N, 3), 360, 1.0).to(DEV)
    total = (res["normal_image"] * wimg).sum() + res["loss_orient"] + res["loss_code"] + (res["image"] ** 2).mean()
    assert_close(total, g["virt|loss"], 1e-2, "loss")
    model.zero_grad()
    total.backward()
    n_ok = grad_digest_check({k: p.grad for k, p in model.named_parameters() if p.grad is not None}, g, "virt", 3e-2)
    assert n_ok >= 40, n_ok


VIRT72 = {"lam": (140, 70.0, 35.0, "lambertian", 0.55, [0.2, 0.5, 0.7]), "tex": (31, 95.0, -120.0, "textureless", 0.3, None)}


@pytest.mark.parametrize("tag", ["lam", "tex"])
def test_virtual_view_step_72_vs_reference_golden(tag):
    """The virtual-view half of the reference's loop (morpheus.py:1393-1408: one step in eleven): ALL 72 x 72 rays of a novel view
    (datasets/dataset.py:503-578 at novel_view_scale 0.2), get_shading's two random outcomes (lambertian / textureless at a
    random ambient ratio, :873-885), a colour / no background, the orientation loss, normal_smooth_3d, normal_smoothness and
    code_reg ON, the SDS guidance replaced by its interface -- a fixed gradient on pred_rgb (trainstep.InjectedGuidance) --
    then the regularisation loss and backward.  Fixture: the reference's OWN render_rays + get_regularization_loss on the same
    rays / samples / weights with the random draws injected (oracle/make_golden.py:gen_round4).  The reference draws the
    smoothness angles on the boolean-indexed points inside the 1.1 sphere; its keep mask rides in the fixture so that the
    same points get the same angles here (tests/util.py: DrawInjector remap)."""
    import numpy as np
    from morpheus_amd import harness
    from bench_support import trainstep
    from tests.util import DrawInjector
    g = load_golden("round4.npz")
    frame, theta, phi, shading, ambient, bg = VIRT72[tag]
    hw, S = 72, 24
    o, d = synth.camera_rays(hw, hw, synth.look_at_pose(theta, phi, 1.5))
    N = o.shape[0]
    smp = of.uniform_samples(o, d, synth.ray_jitter(N), S, 1.01)
    light = of.safe_normalize(o + torch.tensor([0.3, -0.2, 0.5])).to(DEV)
    model = harness.build_model("b", DEV, 0.75).train()
    rend = harness.make_renderer(model, S, samples=tuple(v.to(DEV) for v in smp))
    ts = trainstep.VirtualViewTrainStep(rend, res=hw, guidance=trainstep.InjectedGuidance(hw, hw, DEV, scale=5e-3))
    ts.epoch, ts.keep_outputs = 1000, True                        # progressive level 0.75, as the fixture
    data = dict(H=hw, W=hw, rays_o=o[None].to(DEV), rays_d=d[None].to(DEV), rays_t=torch.full((1, N, 1), frame / 200, device=DEV),
                rays_id=torch.full((1, N, 1), frame, device=DEV, dtype=torch.int64))
    key = "virt72_" + tag
    npts = int(model.config["train"]["trunc"] * 100 + 1)
    keep = np.unpackbits(g[key + "|keep_bits"])[:npts * N].astype(bool)
    assert int(keep.sum()) == int(g[key + "|n_keep"])
    model.zero_grad()
    with DrawInjector(remap={3: keep}) as inj:
        loss = ts(data=data, shading=shading, ambient_ratio=ambient, bg_color=None if bg is None else torch.tensor(bg, device=DEV),
                  light_d=light)
        assert inj.k == int(g[key + "|n_draws"]), "the HIP path must draw what the reference draws, in its order"
    assert abs(model.max_level - 0.75) < 1e-12
    res = ts.last_outputs
    lam = 5e-3                       # colours shaded through FD normals: x250 round-off gain (see the module docstring)
    assert_close(res["image"], g[key + "|image"], lam, "image", floor=FLOOR)
    assert_close(res["depth"], g[key + "|depth"], TOL, "depth", floor=DEPTH_FLOOR)
    assert_close(res["weights_sum"], g[key + "|weights_sum"], TOL, "opacity", floor=FLOOR)
    assert_close(res["sdf"][::16], g[key + "|sdf_s16"], TOL, "sdf", floor=FLOOR)
    assert_close(res["normal"][::16], g[key + "|normal_s16"], 2e-2, "normal (FD)", floor=5e-2)
    assert_close(res["loss_code"], g[key + "|loss_code"], TOL, "loss_code")
    assert_close(res["loss_orient"], g[key + "|loss_orient"], 1e-2, "loss_orient")
    assert_close(res["loss_normal_perturb"], g[key + "|loss_normal_perturb"], 2e-2, "loss_normal_perturb")
    assert_close(res["normal_reg"], g[key + "|normal_reg"], 2e-2, "normal_reg")
    assert_close(loss, g[key + "|loss"], 1e-2, "guidance + regularisation loss")
    loss.backward()
    n_ok = grad_digest_check({k: p.grad for k, p in model.named_parameters() if p.grad is not None}, g, key, 3e-2)
    assert n_ok >= 35, n_ok
    live = ("encoder.embeddings", "deform_code.volumes.2") + (("encoder_c.embeddings", "color_net.net.2.weight_v") if tag == "lam" else ())
    for k in live:                                               # textureless shading does not read the albedo
        assert dict(model.named_parameters())[k].grad.abs().sum() > 0, k
    assert model.pose_array.data.grad is None                    # optimize_pose=False on virtual views


def _delta_digest_check(named, before, golden, key_prefix, tol):
    """grad_digest_check for parameter DELTAS: a delta is the difference of two fp32 parameter values, so every element carries the
    rounding of the parameter it was added to (up to an ulp of |p| on either side) whatever its own size."""
    import numpy as np
    checked = 0
    for k, t in named.items():
        key = key_prefix + k
        if key + "|norm" not in golden:
            continue
        t = t.detach().reshape(-1).double().cpu()
        ulp = 2.0 ** -23 * max(float(before[k].abs().max()), 1e-30)
        gn = float(golden[key + "|norm"])
        assert abs(float(t.norm()) - gn) <= tol * gn + 2 * ulp * np.sqrt(t.numel()), f"{k} norm {float(t.norm())} vs {gn}"
        idx = torch.linspace(0, t.numel() - 1, min(64, t.numel())).long()
        smp = torch.from_numpy(golden[key + "|samples"]).double()
        rms = gn / np.sqrt(t.numel())
        err = float((t[idx] - smp).abs().max())
        assert err <= 4 * tol * max(float(smp.abs().max()), rms) + 2 * ulp, f"{k} samples err {err} (ulp {ulp})"
        checked += 1
    return checked


@pytest.mark.parametrize("variant", ["accum", "freeze"])
def test_cfg4_step_composition_vs_reference_golden(variant):
    """BASELINE configs[3]'s optimiser composition (morpheus.py:1390-1424): one virtual-view backward (x 1 / virtual_freq) and one
    real-view backward into torch.optim.Adam over get_params_all(lr) -- accumulated into ONE step (`accum`: epoch > freeze_epoch), or
    two steps with the deformation groups' learning rates frozen for the first (`freeze`: freeze_lr_deform / reset_lr_deform,
    :504-516) -- against the PARAMETER DELTAS of the reference's own model + optimiser run through the same two steps
    (oracle/make_golden.py:gen_round5; group learning rates from the reference's update_learning_rate at epoch 1000; Adam state
    seeded so that a move is proportional to its gradient).  Here: VirtualViewTrainStep + RealViewTrainStep (fused glue, the background
    colour drawn as train_step draws it) + FlatAdam with the groups matched BY NAME."""
    import numpy as np
    from morpheus_amd import harness
    from bench_support import trainstep
    from morpheus_amd.optim import FlatAdam
    from tests.util import DrawInjector
    g, g4 = load_golden("round5.npz"), load_golden("round4.npz")
    model = harness.build_model("b", DEV, 0.75).train()
    cfg = model.config
    if variant == "freeze":
        # the normal-smoothness term is ill-conditioned on these weights (its gradient changes x 16 for the first step's 1e-3 move,
        # measured on the reference: oracle/make_golden.py:gen_round5); it is pinned at the initial parameters by `accum` and the
        # single-step fixtures, and left out of the variant whose second step runs AFTER a move
        cfg["train"]["normal_smoothness"] = 0.0
    groups = model.get_params_all(cfg["train"]["lr"])
    names = [str(n) for n in g[variant + "|group_names"]]
    assert [gr["name"] for gr in groups] == names                  # the reference's group names, in its order (models/model.py)
    opt = FlatAdam(groups, betas=(0.9, 0.99), eps=1e-15)

    def set_lrs(key):                                              # what update_learning_rate / freeze_lr_deform do: by group NAME
        by_name = dict(zip(names, (float(v) for v in g[variant + "|" + key])))
        for gr in opt.param_groups:
            gr["lr"] = by_name[gr["name"]]

    set_lrs("group_lr")
    assert abs(dict(zip(names, g[variant + "|group_lr"]))["pose"] - 0.1 * dict(zip(names, g[variant + "|group_lr"]))["encoder_sdf"]) < 1e-12
    sd = opt.state_dict()
    for st in sd["state"].values():
        st["step"] = torch.tensor(1000.0)
        st["exp_avg"] = torch.zeros_like(st["exp_avg"])
        st["exp_avg_sq"] = torch.full_like(st["exp_avg_sq"], 1.0)
    opt.load_state_dict(sd)
    set_lrs("group_lr")                                            # (load_state_dict restores the groups' hyper-parameters too)
    before = {k: p.detach().clone() for k, p in model.named_parameters()}
    # the virtual-view step: round4.npz's virt72_lam case
    frame, theta, phi, shading, ambient, bg = VIRT72["lam"]
    hw, S = 72, 24
    o, d = synth.camera_rays(hw, hw, synth.look_at_pose(theta, phi, 1.5))
    N = o.shape[0]
    smp = of.uniform_samples(o, d, synth.ray_jitter(N), S, 1.01)
    rend_v = harness.make_renderer(model, S, samples=tuple(v.to(DEV) for v in smp))
    vs = trainstep.VirtualViewTrainStep(rend_v, res=hw, guidance=trainstep.InjectedGuidance(hw, hw, DEV, scale=5e-3))
    vs.epoch, vs.global_step = 1000, 999
    data_v = dict(H=hw, W=hw, rays_o=o[None].to(DEV), rays_d=d[None].to(DEV), rays_t=torch.full((1, N, 1), frame / 200, device=DEV),
                  rays_id=torch.full((1, N, 1), frame, device=DEV, dtype=torch.int64))
    npts = int(cfg["train"]["trunc"] * 100 + 1)
    keep = np.unpackbits(g4["virt72_lam|keep_bits"])[:npts * N].astype(bool)
    light = of.safe_normalize(o + torch.tensor([0.3, -0.2, 0.5])).to(DEV)
    # the real-view step: extras.npz's realview rays, background drawn inside the step
    sel = torch.from_numpy(g["real|sel"].astype(np.int64))
    hw_r, S_r = 32, 64
    o_r, d_r, t_r, rid_r = [v[:, sel] for v in synth.frame_rays(25, hw_r, hw_r)]
    N_r = o_r.shape[1]
    smp_r = of.uniform_samples(o_r[0], d_r[0], synth.ray_jitter(hw_r * hw_r)[sel], S_r, 1.01)
    rend_r = harness.make_renderer(model, S_r, samples=tuple(v.to(DEV) for v in smp_r))
    frame_r = trainstep.make_frames([25], hw_r, hw_r, DEV)[0]
    ts = trainstep.RealViewTrainStep(rend_r, [frame_r], ray_num=N_r)
    ts.epoch = 1000
    data_r = trainstep.sample_real_view_rays(frame_r, N_r, sel.to(DEV))

    def virtual_backward():
        with DrawInjector(remap={3: keep} if cfg["train"]["normal_smoothness"] > 0 else None):
            lv = vs(data=data_v, shading=shading, ambient_ratio=ambient, bg_color=torch.tensor(bg, device=DEV), light_d=light)
        ((1.0 / cfg["train"]["virtual_freq"]) * lv).backward()
        return lv

    def real_backward():
        ts.apply_level()
        with DrawInjector() as inj, model.operand_scope():
            lr_ = ts._step(data_r, 1000)
            assert inj.k == int(g[variant + "|real_n_draws"]), "background first, then what render_rays draws, in the reference's order"
        lr_.backward()
        return lr_

    opt.zero_grad()
    if variant == "accum":
        lv = virtual_backward()
        lr_ = real_backward()
        opt.step()
    else:
        set_lrs("group_lr_frozen")
        assert all(gr["lr"] == 0.0 for gr in opt.param_groups if gr["name"] in ("code_deform", "decoder_deform", "decoder_topo"))
        lv = virtual_backward()
        opt.step()
        d1 = {k: p.detach() - before[k] for k, p in model.named_parameters()}
        for k, v in d1.items():                                    # frozen groups (and the pose: no gradient on virtual views) stay put
            if k.startswith(("deform_code", "deform_net", "topo_net", "pose_array")):
                assert float(v.abs().max()) == 0.0, k
        assert _delta_digest_check(d1, before, g, variant + "|delta1|", 5e-3) >= 55
        opt.zero_grad()
        set_lrs("group_lr")
        lr_ = real_backward()
        opt.bucket.collect()
        lay = {id(p): (o, k) for p, o, k in opt._views}
        g2 = {k: opt.bucket.flat[lay[id(p)][0]:lay[id(p)][0] + lay[id(p)][1]].view(p.shape) for k, p in model.named_parameters()}
        assert grad_digest_check(g2, {k.replace("|grad2|", "|grad|"): v for k, v in g.items() if "|grad2|" in k}, variant, 3e-3) >= 50
        opt.step()
    assert_close(lv, g[variant + "|loss_virtual"], 1e-2, "virtual-view loss")
    assert_close(lr_, g[variant + "|loss_real"], 2e-3, "real-view loss")
    delta = {k: p.detach() - before[k] for k, p in model.named_parameters()}
    # accum: one step on the SUM of a virtual-view and a real-view gradient, both through finite-difference normals (x 250 round-off
    # gain, the tolerance of the single-step fixtures); freeze: no normal-smoothness term -> the judge's 1e-3 rel-L2 of the deltas
    # (measured: <= 8.8e-4 on every tensor; samples of tensors whose whole move is a few fp32 ulps get the 3e-3 the digest allows)
    n_ok = _delta_digest_check(delta, before, g, variant + "|delta|", 3e-2 if variant == "accum" else 3e-3)
    assert n_ok >= 55, n_ok
    if variant == "freeze":
        for k, v in delta.items():
            gn = float(g["freeze|delta|" + k + "|norm"])
            ulp = 2.0 ** -23 * float(before[k].abs().max())
            if gn > 100 * ulp * v.numel() ** 0.5:                  # (a tensor that moved by a few ulps per element has no 1e-3 to show)
                assert abs(float(v.double().norm()) - gn) <= 1e-3 * gn, (k, float(v.double().norm()), gn)
    moved = [k for k, v in delta.items() if float(v.abs().max()) > 0]
    for k in ("encoder.embeddings", "encoder_c.embeddings", "deform_code.volumes.2", "pose_array.data", "deform_net.net.2.weight_v"):
        assert k in moved, k
    for k in delta:                                                # the background net never gets a gradient: never stepped
        if k.startswith("bg_net"):
            assert k not in moved, k


def test_parked_memory_cap_chunks_the_queries(monkeypatch):
    """MORPHEUS_MAX_PARK_GB (morpheus_amd/chunking.py): over the cap, the main query and the perturbed-normal query of a training
    render run in row chunks, all but the last through checkpointing (nothing parked in forward, re-made one chunk at a time in
    backward).  The 72 x 72 virtual-view step (lambertian shading through FD normals, orientation loss, normal_smooth_3d,
    normal_smoothness, code_reg; the draws injected) with the cap at 0.4 GB -- 16 chunks per query -- against the same step
    without a cap: same outputs, same loss, same gradients (different summation orders in the weight-gradient and table sums only).
    Third run (round-5 advisor finding): the capped step with backward called INSIDE an open `model.operand_scope()` -- what
    INTEGRATION.md invites ("wrap a step") -- where a chunk's re-run used to pick up the step's cached operands and, being a nested
    autograd pass, reset the outer pass's in-place gradient sums; the re-run now takes a scope of its own (fresh_operand_scope)."""
    import contextlib
    import numpy as np
    from morpheus_amd import chunking, harness
    from bench_support import trainstep
    from tests.util import DrawInjector
    g4 = load_golden("round4.npz")
    frame, theta, phi, shading, ambient, bg = VIRT72["lam"]
    hw, S = 72, 24
    o, d = synth.camera_rays(hw, hw, synth.look_at_pose(theta, phi, 1.5))
    N = o.shape[0]
    smp = of.uniform_samples(o, d, synth.ray_jitter(N), S, 1.01)
    light = of.safe_normalize(o + torch.tensor([0.3, -0.2, 0.5])).to(DEV)
    npts = None

    def run(cap_gb, backward_inside_scope=False):
        monkeypatch.setenv("MORPHEUS_MAX_PARK_GB", str(cap_gb))
        model = harness.build_model("b", DEV, 0.75).train()
        rend = harness.make_renderer(model, S, samples=tuple(v.to(DEV) for v in smp))
        ts = trainstep.VirtualViewTrainStep(rend, res=hw, guidance=trainstep.InjectedGuidance(hw, hw, DEV, scale=5e-3))
        ts.epoch, ts.keep_outputs = 1000, True
        data = dict(H=hw, W=hw, rays_o=o[None].to(DEV), rays_d=d[None].to(DEV), rays_t=torch.full((1, N, 1), frame / 200, device=DEV),
                    rays_id=torch.full((1, N, 1), frame, device=DEV, dtype=torch.int64))
        keep = np.unpackbits(g4["virt72_lam|keep_bits"])[:int(model.config["train"]["trunc"] * 100 + 1) * N].astype(bool)
        before = dict(chunking.STATS)
        with (model.operand_scope() if backward_inside_scope else contextlib.nullcontext()):
            with DrawInjector(remap={3: keep}):
                loss = ts(data=data, shading=shading, ambient_ratio=ambient, bg_color=torch.tensor(bg, device=DEV), light_d=light)
            loss.backward()
        stats = {k: chunking.STATS[k] - before[k] for k in before}
        res = ts.last_outputs
        return (float(loss.detach()), {k: res[k].detach().clone() for k in ("image", "depth", "sdf", "normal", "weights")},
                {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}, stats)

    l_ref, out_ref, g_ref, st_ref = run(0)             # no bound
    l_cap, out_cap, g_cap, st_cap = run(0.4)
    assert st_ref["chunked_calls"] == 0 and st_cap["chunked_calls"] == 2 and st_cap["chunks"] >= 2 * 8, (st_ref, st_cap)
    assert st_cap["rerun_rows"] > 1.5 * N * S          # both queries, all chunks but the last
    for k in out_ref:
        assert torch.equal(out_ref[k], out_cap[k]), k  # forward values do not depend on how the rows are batched
    assert abs(l_ref - l_cap) <= 1e-6 * abs(l_ref), (l_ref, l_cap)
    assert set(g_ref) == set(g_cap) and len(g_ref) >= 50
    for k, a in g_ref.items():
        rel = float((a.double() - g_cap[k].double()).norm() / a.double().norm().clamp_min(1e-30))
        assert rel <= 2e-4, (k, rel)
    l_in, out_in, g_in, st_in = run(0.4, backward_inside_scope=True)
    assert st_in["chunked_calls"] == 2 and abs(l_ref - l_in) <= 1e-6 * abs(l_ref)
    assert set(g_in) == set(g_ref)
    for k, a in g_ref.items():
        rel = float((a.double() - g_in[k].double()).norm() / a.double().norm().clamp_min(1e-30))
        assert rel <= 2e-4, ("backward inside the scope", k, rel)


def test_in_place_gradient_sums_under_other_autograd_entry_points():
    """The field queries of a step add their weight / beta / table gradients into shared tensors that the operand pack's backward
    hands over (ops._QueryAccumulator, keyed on the autograd graph task).  Besides loss.backward() that has to hold for
    torch.autograd.grad(inputs=...) (the pack node may be pruned: only some leaves asked for), for two backward passes over one
    retained graph, and for a plain backward AFTER those -- each against the per-query form (ACCUMULATE_IN_PLACE off)."""
    from morpheus_amd import harness, ops
    hw, S = 16, 32
    o, d, t, rid = [v.to(DEV) for v in synth.frame_rays(25, hw, hw)]
    N = o.shape[1]
    timg, tdep = [v.to(DEV) for v in synth.targets(N)]
    light = of.safe_normalize(o[0].cpu() + torch.tensor([0.3, -0.2, 0.5])).to(DEV)

    def build():
        model = harness.build_model("b", DEV, 0.75).train()
        for k in ("normal_smoothness", "normal_smooth_3d"):
            model.config["train"][k] = 0.0
        rend = harness.make_renderer(model, S, jitter=synth.ray_jitter(N).to(DEV))
        res = rend.render_rays(o, d, t, rid, hw, hw, ambient_ratio=0.4, light_d=light, shading="lambertian", real_view=False)
        return model, harness.bench_loss(res, timg, tdep) + res["loss_orient"] + res["loss_code"]

    def named(model, keys):
        p = dict(model.named_parameters())
        return [p[k] for k in keys]

    field_keys = ["encoder.embeddings", "encoder_c.embeddings", "sdf2density.beta", "sdf_net.net.1.weight", "color_net.net.0.weight_v"]
    warp_keys = ["deform_net.net.2.weight_v", "topo_net.net.5.bias", "deform_code.volumes.2"]
    saved = ops.ACCUMULATE_IN_PLACE
    try:
        ops.ACCUMULATE_IN_PLACE = False
        model, loss = build()
        loss.backward()
        want = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
        ops.ACCUMULATE_IN_PLACE = saved
        assert ops.ACCUMULATE_IN_PLACE, "this torch has no graph-task id: nothing to test"

        def close(got, k, scale=1.0):
            rel = float((got.double() - scale * want[k].double()).norm() / (scale * want[k].double().norm()).clamp_min(1e-30))
            assert rel <= 2e-5, (k, rel)

        # (a) autograd.grad for the leaves behind the pack (tables, beta, field weights) and for warp-side leaves only (pack pruned)
        model, loss = build()
        for keys in (field_keys, warp_keys, field_keys[:1] + warp_keys[:1]):
            for k, gk in zip(keys, torch.autograd.grad(loss, named(model, keys), retain_graph=True)):
                close(gk, k)
        # (b) ... then two plain backward passes over the same retained graph: every gradient exactly twice
        loss.backward(retain_graph=True)
        loss.backward()
        for k, p in model.named_parameters():
            if p.grad is not None and k in want:
                close(p.grad, k, 2.0)
        # (c) a fresh step afterwards is untouched by what the earlier passes left behind
        model, loss = build()
        loss.backward()
        for k, p in model.named_parameters():
            if p.grad is not None and k in want:
                close(p.grad, k)
    finally:
        ops.ACCUMULATE_IN_PLACE = saved


def virt24_errors(hw=24, S=24, fixture="round5.npz", tag="virt24"):
    """-> rows (tensor, |grad|, reference-fp32 error, HIP error, HIP error without its worst sample), the errors as max |sample - f64
    sample| / max |f64 sample| over the 64 strided samples of the fixture's digests, plus (loss_f64, loss_ref32, loss_hip).  The 24 x 24 virtual-view training step of
    round5.npz / the 72 x 72 x 32-sample one of round6.npz: the reference ran them in fp32 AND in double
    (oracle/make_golden.py:_virtual_step_in_double)."""
    import numpy as np
    from morpheus_amd import harness
    from bench_support import trainstep
    from tests.util import DrawInjector
    g = load_golden(fixture)
    frame = 140
    o, d = synth.camera_rays(hw, hw, synth.look_at_pose(70.0, 35.0, 1.5))
    N = o.shape[0]
    smp = of.uniform_samples(o, d, synth.ray_jitter(N), S, 1.01)
    light = of.safe_normalize(o + torch.tensor([0.3, -0.2, 0.5])).to(DEV)
    model = harness.build_model("b", DEV, 0.75).train()
    model.config["train"]["normal_smoothness"] = 0.0
    rend = harness.make_renderer(model, S, samples=tuple(v.to(DEV) for v in smp))
    ts = trainstep.VirtualViewTrainStep(rend, res=hw, guidance=trainstep.InjectedGuidance(hw, hw, DEV, scale=5e-3))
    ts.epoch, ts.global_step = 1000, 999
    data = dict(H=hw, W=hw, rays_o=o[None].to(DEV), rays_d=d[None].to(DEV), rays_t=torch.full((1, N, 1), frame / 200, device=DEV),
                rays_id=torch.full((1, N, 1), frame, device=DEV, dtype=torch.int64))
    model.zero_grad()
    with DrawInjector() as inj:
        loss = ts(data=data, shading="lambertian", ambient_ratio=0.55, bg_color=torch.tensor([0.2, 0.5, 0.7], device=DEV), light_d=light)
        assert inj.k == int(g[tag + "|f32|n_draws"]) == int(g[tag + "|f64|n_draws"])
    loss.backward()
    rows = []
    for k, p in model.named_parameters():
        k64 = tag + "|f64|grad|" + k
        if p.grad is None or k64 + "|samples" not in g:
            continue
        s64 = torch.from_numpy(g[k64 + "|samples"]).double()
        s32 = torch.from_numpy(g[tag + "|f32|grad|" + k + "|samples"]).double()
        gr = p.grad.detach().reshape(-1).double().cpu()
        idx = torch.linspace(0, gr.numel() - 1, min(64, gr.numel())).long()
        scale = float(s64.abs().max())
        if scale == 0.0:
            continue
        e_hip = torch.sort((gr[idx] - s64).abs(), descending=True).values / scale
        rows.append((k, float(g[k64 + "|norm"]), float((s32 - s64).abs().max()) / scale, float(e_hip[0]), float(e_hip[min(1, len(e_hip) - 1)])))
    return rows, (float(g[tag + "|f64|loss"]), float(g[tag + "|f32|loss"]), float(loss.detach()))


def test_virtual_view_gradients_against_the_reference_in_double():
    """Round 4's 72 x 72 test holds the virtual-view step's gradients to the reference's fp32 fixture at 3e-2 (finite-difference
    normals amplify round-off x 250) -- loose enough to miss a 1 % error.  Here the allowance is DERIVED: the reference ran a 24 x 24
    virtual-view training step (lambertian shading, orientation loss, normal_smooth_3d, code_reg, guidance gradient injected) in fp32
    and in float64; the HIP path's gradient samples must be as close to the float64 ones as the reference's own fp32 run is, within a
    factor (3 x, or 2e-4 of the tensor's largest sample where the reference's own error is smaller than that)."""
    rows, (l64, l32, lhip) = virt24_errors()
    assert len(rows) >= 45, len(rows)
    assert abs(lhip - l64) <= max(3 * abs(l32 - l64), 1e-5 * abs(l64)), (l64, l32, lhip)
    worst = []
    for k, norm, e_ref, e_hip, _ in rows:
        if e_hip > max(3 * e_ref, 2e-4):
            worst.append((k, e_ref, e_hip))
    assert not worst, worst
    # and the derived allowance is tight: the median of the reference's own error over the tensors is 1e-3 (worst 1.3e-2)
    assert sorted(r[2] for r in rows)[len(rows) // 2] < 2e-3


def test_virtual_view_gradients_72_against_the_reference_in_double():
    """The same derived gate at the size of the 72 x 72 fixture test ("nothing of that kind exists at 72^2"): all 5 184 rays
    of the novel view x 32 samples through lambertian shading on finite-difference normals, orientation loss, normal_smooth_3d,
    code_reg and the injected guidance gradient; the reference's fp32 and float64 runs are tests/golden/round6.npz
    (oracle/make_golden.py:gen_round6).  HIP <= 3 x the reference's own fp32 error per tensor, or 2e-4 of the tensor's largest sample."""
    rows, (l64, l32, lhip) = virt24_errors(72, 32, "round6.npz", "virt72d")
    assert len(rows) >= 45, len(rows)
    assert abs(lhip - l64) <= max(3 * abs(l32 - l64), 1e-5 * abs(l64)), (l64, l32, lhip)
    # A hash TABLE gradient is a sum over the points whose cell touches the row: a tap point whose canonical position differs by an ulp
    # between two fp32 implementations can sit in the neighbouring cell of some level, and its contribution then lands on other rows --
    # a discrete event, not round-off.  Measured (tools/gpu/virt_flip_diag.py, profiles/r06_virt72_gradients_vs_f64.txt): b3 against
    # f32 on this step, 190 811 touched rows: median row difference 9.5e-8 of the largest entry, 104 rows beyond 1e-4 (up to 1.7e-3),
    # one of them among the fixture's 64 strided samples.  So for the two tables ONE sample may be such an event (bounded at 2e-3);
    # every other sample, and every other tensor, is held to the derived allowance.
    worst = []
    for k, norm, e_ref, e_hip, e_hip_2nd in rows:
        allow = max(3 * e_ref, 2e-4)
        if k.endswith("embeddings"):
            if e_hip > 2e-3 or e_hip_2nd > allow:
                worst.append((k, e_ref, e_hip, e_hip_2nd))
        elif e_hip > allow:
            worst.append((k, e_ref, e_hip))
    assert not worst, worst


def test_two_frames_vs_reference_golden():
    """B = 2 frames in one batch against the reference's own render_rays (fixture extras.npz:two|*)."""
    from morpheus_amd import harness
    g = load_golden("extras.npz")
    hw, S = 16, 48
    fr = [synth.frame_rays(fid, hw, hw) for fid in (0, 25)]
    o, d, t, rid = [torch.cat([f[k] for f in fr], 0) for k in range(4)]
    N = o.shape[1]
    smp = of.uniform_samples(o.reshape(-1, 3), d.reshape(-1, 3), synth.ray_jitter(2 * N), S, 1.01)
    light = of.safe_normalize(o.reshape(-1, 3) + torch.tensor([0.3, -0.2, 0.5]))
    model = harness.build_model("b", DEV).eval()
    rend = harness.make_renderer(model, S, samples=tuple(v.to(DEV) for v in smp))
    res = rend.render_rays(o.to(DEV), d.to(DEV), t.to(DEV), rid.to(DEV), hw, hw, ambient_ratio=1.0, light_d=light.to(DEV),
                           shading="albedo")
    assert_close(res["image"], g["two|image"], TOL, "image", floor=FLOOR)
    assert_close(res["depth"], g["two|depth"], TOL, "depth", floor=DEPTH_FLOOR)
    assert_close(res["deform"][::16], g["two|deform_s16"], TOL, "deform", floor=1e-3)
    timg, tdep = [v.to(DEV) for v in synth.targets(2 * N)]
    loss = ((res["image"].reshape(-1, 3) - timg) ** 2).mean() + ((res["depth"].reshape(-1) - tdep) ** 2).mean()
    assert_close(loss, g["two|loss"], TOL, "loss")
    model.zero_grad()
    loss.backward()
    n_ok = grad_digest_check({k: p.grad for k, p in model.named_parameters() if p.grad is not None}, g, "two", 5e-4)
    assert n_ok >= 40, n_ok


def _real_view_setup(capacity=None, seed_jitter=True):
    """2 048-ray real-view training render on a marched occupancy ball (the reference's call shape), jitter pinned."""
    from morpheus_amd import harness
    from morpheus_amd.occgrid import OccupancyGrid
    from morpheus_amd.render import HotPathRenderer
    hw = 32
    o, d, t, rid = [v.to(DEV) for v in synth.frame_rays(25, hw, hw)]
    N = o.shape[1]
    c = (torch.arange(128).float() + 0.5) / 128 * 2.02 - 1.01
    X, Y, Z = torch.meshgrid(c, c, c, indexing="ij")
    ball = ((X ** 2 + Y ** 2 + Z ** 2).sqrt() < 0.6).to(torch.uint8).contiguous()
    model = harness.build_model("b", DEV, 0.75).train()
    grid = OccupancyGrid([-1.01] * 3 + [1.01] * 3, 128).to(DEV)
    grid.set_binary(ball.to(DEV))
    grid.fixed_jitter = synth.ray_jitter(N).to(DEV)
    grid.sample_capacity = capacity
    rend = HotPathRenderer(model, model.config, grid, 200)
    depth = synth.hash_tensor((1, N, 1), 400, 0.3, 1.5).to(DEV)
    mask = (synth.hash_tensor((1, N, 1), 401, 0.5, 0.5) > 0.3).float().to(DEV)
    return model, grid, rend, (o, d, t, rid), depth, mask, N


def test_fixed_capacity_sampling_equals_ragged_sampling(monkeypatch):
    """OccupancyGrid.sample_capacity (constant-shape packed samples + a device-side count: what lets a training step be captured
    in a HIP graph) against the ragged layout on the same rays: the rendered outputs, the per-sample outputs of the real
    samples, every in-render loss (incl. the per-sample means, which must leave the padding out) and every parameter gradient."""
    # the random regularisers draw per SAMPLE (shape [M] vs [capacity]): pin the draws to a function of nothing
    monkeypatch.setattr(torch, "randn_like", lambda t, **kw: torch.full_like(t, 0.37))
    monkeypatch.setattr(torch, "rand_like", lambda t, **kw: torch.full_like(t, 0.61))
    monkeypatch.setattr(torch, "rand", lambda *s, **kw: torch.full(s[0] if len(s) == 1 and isinstance(s[0], (list, tuple, torch.Size)) else s, 0.43,
                                                                   device=kw.get("device")))
    outs = {}
    for tag in ("ragged", "capped"):
        model, grid, rend, (o, d, t, rid), depth, mask, N = _real_view_setup()
        if tag == "capped":
            grid.sample_capacity = outs["ragged"][3] + 9000          # not a multiple of anything: 9 000 padding entries
        res = rend.render_rays(o, d, t, rid, N, 1, ambient_ratio=1.0, shading="albedo_normal", real_view=True, cano=False,
                               rays_depth=depth, rays_mask=mask, optimize_pose=True)
        M = int(grid.n_valid) if tag == "capped" else res["sdf"].shape[0]
        loss = (res["image"] ** 2).mean() + (res["depth"] ** 2).mean() + res["loss_normal_perturb"] + res["normal_reg"] + \
            res["sdf_loss"] + 0.1 * res["fs_loss"] + res["loss_code"]
        model.zero_grad()
        loss.backward()
        outs[tag] = (res, float(loss), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}, M)
    (ra, la, ga, Ma), (rb, lb, gb, Mb) = outs["ragged"], outs["capped"]
    assert Ma == Mb and rb["sdf"].shape[0] == Ma + 9000 and int(rb["valid"].sum()) == Ma
    for k in ("image", "depth", "weights_sum"):
        assert torch.equal(ra[k], rb[k]), k
    for k in ("sdf", "weights", "normal", "deform"):
        assert torch.equal(ra[k], rb[k][:Ma]), k
    assert float(rb["weights"][Ma:].abs().max()) == 0.0
    for k in ("loss_normal_perturb", "normal_reg", "sdf_loss", "fs_loss", "loss_code"):
        assert_close(rb[k], ra[k], 2e-5, k, floor=1e-6)     # sums of ~10^5 floats by atomics / a different block partition (measured 1.3e-6)
    assert abs(la - lb) <= 2e-5 * abs(la)
    assert set(ga) == set(gb)
    for k in ga:
        rel = float((ga[k] - gb[k]).norm() / ga[k].norm().clamp_min(1e-30))
        assert rel <= 1e-4, (k, rel)        # same terms; atomics / partial sums in another order
    # too small a capacity: the tail rays are truncated and the sticky overflow flag says so
    model, grid, rend, (o, d, t, rid), depth, mask, N = _real_view_setup(capacity=4096)
    with torch.no_grad():
        res = rend.render_rays(o, d, t, rid, N, 1, ambient_ratio=1.0, shading="albedo")
    assert int(grid.overflow) == 1 and int(grid.n_valid) == 4096 and res["sdf"].shape[0] == 4096


def test_graph_captured_after_eager_steps_on_the_same_parameters():
    """the optimisation loop's mix (bench.py --workload train_loop --graph): an eager real-view step whose loss keeps only its
    value, then a replayed step whose capacity bucket has NOT been captured yet -- the capture happens in the middle of the run,
    after eager backward passes on the same parameters (a loss that still held its graph made exactly this capture crash in a
    100-iteration soak).  The captured step must produce a finite loss and a gradient, and the optimiser must move on it."""
    from morpheus_amd import harness
    from bench_support import trainstep
    from morpheus_amd.occgrid import OccupancyGrid
    from morpheus_amd.optim import FlatAdam
    from morpheus_amd.render import HotPathRenderer
    model = harness.build_model("b", DEV).train()
    grid = OccupancyGrid([-model.bound] * 3 + [model.bound] * 3, 128).to(DEV)
    rend = HotPathRenderer(model, model.config, grid, 200)
    ts = trainstep.RealViewTrainStep(rend, trainstep.make_frames([25, 33], 64, 64, DEV), ray_num=512)
    ts.epoch = 1000
    opt = FlatAdam(model.get_params_all(model.config["train"]["lr"]), betas=(0.9, 0.99), eps=1e-15)
    with torch.no_grad():
        trainstep.warm_up_occupancy(ts)
    ts.global_step = 4096 + 3
    gs = trainstep.GraphedRealViewStep(ts, opt.bucket)            # no prepare(): nothing captured yet
    for _ in range(2):                                             # eager steps first, as the loop's virtual / first real step
        opt.bucket.zero()
        loss = ts()
        loss.backward()
        loss = loss.detach()
        opt.bucket.allreduce_mean()
        opt.step()
    p0 = opt.flat_p.clone()
    assert gs.n_captures == 0
    lg = gs()                                                      # captures its bucket here, then replays it
    assert gs.n_captures == 1 and bool(torch.isfinite(lg)) and float(opt.bucket.flat.abs().max()) > 0
    opt.step()
    loss = ts()                                                    # and an eager step after it still works on the same renderer
    loss.backward()
    assert bool(torch.isfinite(loss.detach())) and not torch.equal(opt.flat_p, p0)
    gs.release()


def test_graphed_real_view_step_replays_the_eager_step():
    """trainstep.GraphedRealViewStep: the real-view step captured in a HIP graph.  Draw-for-draw equality with the eager step is
    not available (the graph owns its Philox offsets), so: (i) with every random draw pinned the replayed graph's loss and
    gradient bucket equal the eager step's on the same frame; (ii) un-pinned, three replays give finite, different losses,
    the optimiser moves the parameters, the sample count stays under the capacity (the next batch is drawn and counted on a
    side stream while a replay runs); (iii) a progressive-level change captures a graph of its own."""
    from morpheus_amd import harness
    from bench_support import trainstep
    from morpheus_amd.occgrid import OccupancyGrid
    from morpheus_amd.optim import FlatAdam
    from morpheus_amd.render import HotPathRenderer

    def build():
        model = harness.build_model("b", DEV).train()
        grid = OccupancyGrid([-model.bound] * 3 + [model.bound] * 3, 128).to(DEV)
        rend = HotPathRenderer(model, model.config, grid, 200)
        frames = trainstep.make_frames([25, 33], 64, 64, DEV)
        ts = trainstep.RealViewTrainStep(rend, frames, ray_num=512)
        ts.epoch = 1000
        opt = FlatAdam(model.get_params_all(model.config["train"]["lr"]), betas=(0.9, 0.99), eps=1e-15)
        with torch.no_grad():
            trainstep.warm_up_occupancy(ts)
        ts.global_step = 4096 + 3                       # no occupancy refresh in the next steps
        return model, grid, ts, opt

    # (i) pinned draws: four optimiser steps eager vs four replays of the captured step, from the same weights
    saved = (torch.rand, torch.rand_like, torch.randn_like, torch.randint)
    try:
        torch.rand = lambda *s, **kw: torch.full(s[0] if len(s) == 1 and isinstance(s[0], (list, tuple, torch.Size)) else s, 0.43, device=kw.get("device"))
        torch.rand_like = lambda t, **kw: torch.full_like(t, 0.61)
        torch.randn_like = lambda t, **kw: torch.full_like(t, 0.37)
        torch.randint = lambda lo, hi, size, **kw: (torch.arange(size[0], device=kw.get("device")) * 7) % hi
        def eager_run():                            # (the marcher's per-ray jitter is a torch.rand draw: pinned with the rest)
            model, grid, ts, opt = build()
            p0 = opt.flat_p.clone()
            losses, first = [], None
            for k in range(4):
                opt.bucket.zero()
                ts.begin_step()
                fi = ts.frame_of_step()
                with model.operand_scope():
                    le = ts._step(trainstep.sample_real_view_rays(ts.frames[fi], ts.ray_num), ts.global_step)
                le.backward()
                opt.bucket.collect()
                if k == 0:
                    first = opt.bucket.flat.clone()
                opt.step()
                losses.append(float(le))
                del le                                   # no eager autograd graph (AccumulateGrad nodes of the default stream) survives
            return losses, first, p0, opt.flat_p.clone()

        eager_losses, flat_first, p_init, p_eager = eager_run()
        eager_again, _, _, p_again = eager_run()        # the run-to-run spread of the eager step itself (atomics in the scatters)
        model, grid, ts, opt = build()
        gs = trainstep.GraphedRealViewStep(ts, opt.bucket)
        caps = gs.prepare()                              # several capacity buckets captured one after the other: the replays below
        assert len(caps) >= 3 and gs.n_captures == len(caps)       # use graphs that were NOT the first capture of the process
        graph_losses = []
        for k in range(4):
            lg = gs()
            if k == 0:
                rel = float((opt.bucket.flat - flat_first).norm() / flat_first.norm())
                assert rel <= 1e-4, rel                  # the first replay's gradient bucket == the eager step's
            opt.step()
            graph_losses.append(float(lg))
        assert gs.n_captures == len(caps)               # every batch found its bucket among the prepared ones
        assert not gs.check_overflow() and gs.last_samples <= gs.last_capacity < 1.02 * gs.last_samples + 512 + gs.bucket_step
        print("graphed vs eager loss, relative:", ["%.2e" % (abs(a - b) / abs(b)) for a, b in zip(graph_losses, eager_losses)])
        print("eager vs eager loss, relative:  ", ["%.2e" % (abs(a - b) / abs(b)) for a, b in zip(eager_again, eager_losses)])
        dc = (p_again - p_init).double()
        print("eager vs eager: distance / moved %.4f" % float((dc - (p_eager - p_init).double()).norm() / dc.norm()))
        for k, (a, b) in enumerate(zip(graph_losses, eager_losses)):
            # the first two steps agree to round-off (7 digits); Adam with eps = 1e-15 then amplifies the round-off of noise-sized
            # gradients (the padded layout sums the weight gradients in a different order: 5e-9 of the bucket) and the trajectories
            # drift apart -- measured 2e-6 / 6-10e-5 at steps 3 and 4, next to 0-2e-7 / 1e-6-3e-5 between two EAGER runs of the same
            # steps (printed above: atomics in the scatters make the eager step itself run-dependent); the growth is ~30x per step,
            # which is why the comparison stops after four
            assert abs(a - b) <= (1e-5 if k < 2 else (3e-5 if k == 2 else 2e-3)) * abs(b), (graph_losses, eager_losses)
        # Adam steps with eps = 1e-15: every touched entry moves by ~lr per step whatever its gradient's size, so the ~1 % of
        # entries whose gradient is round-off noise step in a run-dependent direction (measured after four steps: distance between
        # the two runs 0.075-0.078 of the distance moved, cosine 0.997).  The displacements must point the same way
        da, db = (opt.flat_p - p_init).double(), (p_eager - p_init).double()
        cos = float((da * db).sum() / (da.norm() * db.norm()))
        print("cosine of the two four-step displacements: %.5f, distance / moved: %.4f" % (cos, float((da - db).norm() / db.norm())))
        assert cos >= 0.99, cos
    finally:
        torch.rand, torch.rand_like, torch.randn_like, torch.randint = saved
    # (ii) un-pinned replays drive the optimiser
    model, grid, ts, opt = build()
    gs = trainstep.GraphedRealViewStep(ts, opt.bucket)
    p0 = opt.flat_p.clone()
    losses = []
    for _ in range(3):
        losses.append(float(gs()))
        opt.step()
    assert all(l == l and abs(l) < 1e6 for l in losses) and len(set(losses)) == 3, losses
    assert float((opt.flat_p - p0).abs().max()) > 0 and not gs.check_overflow()
    # (iii) a new progressive level changes the kernels' band / level counts: the step captures a graph for it by itself
    n0 = len(gs.graphs)
    ts.epoch = 0
    losses.append(float(gs()))
    assert len(gs.graphs) == n0 + 1 and {k[1:] for k in gs.graphs} == {(4, 12), (3, 8)} and losses[-1] == losses[-1]
    # ... but a level change that leaves the band / level counts alone does NOT (progressive_level yields a new float every epoch,
    # morpheus.py:808-813: keyed on the raw float, every epoch would re-capture every bucket and keep the old pools)
    n1, c1 = len(gs.graphs), gs.n_captures
    ts.epoch = 8                                         # max_level 0.502: still 3 bands, 9 levels? -> ceil(0.502 * 16) = 9: a new key
    gs()
    ts.epoch = 9                                         # 0.50225: the same counts as epoch 8 -> at most a new capacity bucket
    gs()
    lv = {k[1:] for k in gs.graphs}
    assert lv == {(4, 12), (3, 8), (3, 9)}, lv
    # the capacity and the static jitter belong to the captured body, not to the renderer's shared occupancy grid: an eager
    # render_rays of another batch size on the same renderer afterwards runs ragged and un-truncated
    assert grid.sample_capacity is None and grid.fixed_jitter is None
    o, d, t, rid = [v.to(DEV) for v in synth.frame_rays(25, 24, 24)]
    with torch.no_grad():
        res = ts.r.render_rays(o, d, t, rid, 24, 24, ambient_ratio=1.0, shading="albedo")
    assert res["image"].shape == (1, 576, 3) and "n_valid" not in res
    # least-recently-used eviction beyond max_graphs
    gs.max_graphs = len(gs.graphs)
    ts.epoch = 2000
    gs()
    assert len(gs.graphs) == gs.max_graphs and gs.n_evicted == 1
    gs.release()


def test_reference_glue_equals_fused_glue():
    """bench.py --workload train_real --glue reference: the reference's own caller-side loss code (trainstep.ReferenceGlue: operator
    chains, in-place masks, the boolean index of morpheus.py:1018) around the swapped-in render_rays gives the loss and the
    gradients of this build's fused glue (ops.real_view_render_loss / masked_mean / weighted_sum) on the same batch and draws."""
    from morpheus_amd import harness
    from bench_support import trainstep
    from morpheus_amd.occgrid import OccupancyGrid
    from morpheus_amd.render import HotPathRenderer
    out = {}
    for glue in ("fused", "reference", "reference_scoped"):
        model = harness.build_model("b", DEV).train()
        grid = OccupancyGrid([-model.bound] * 3 + [model.bound] * 3, 128).to(DEV)
        rend = HotPathRenderer(model, model.config, grid, 200)
        ts = trainstep.RealViewTrainStep(rend, trainstep.make_frames([25], 64, 64, DEV), ray_num=512, glue=glue)
        ts.epoch = 1000
        torch.manual_seed(11)                   # the occupancy warm-up draws one jittered point per cell: same grid for both
        with torch.no_grad():
            trainstep.warm_up_occupancy(ts)
        ts.global_step = 4096 + 3
        loss = ts(frame_index=0)
        model.zero_grad()
        loss.backward()
        out[glue] = (float(loss), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
    (lr_, gr) = out["reference"]
    for other in ("fused", "reference_scoped"):      # (scoped: one `with model.operand_scope():` line around the reference's step)
        lf, gf = out[other]
        assert abs(lf - lr_) <= 2e-5 * abs(lr_), (other, lf, lr_)
        assert set(gf) == set(gr)
        for k in gf:
            rel = float((gf[k] - gr[k]).norm() / gr[k].norm().clamp_min(1e-30))
            assert rel <= 2e-3, (other, k, rel)          # same terms, other summation orders; FD-normal terms amplify round-off


def test_step_cache_lifecycle():
    """model._step_cache (round 6): a training forward keeps its prepared operands in the model, so that the reference's train_step
    -- render_rays, then model.density for the surface points, no operand scope anywhere -- prepares them once.  Checked here:
    (i) two model calls of one training forward share ONE operand pack, and the gradients equal those of a scope per call;
    (ii) the backward pass ends the step: forward / backward pairs accumulated before one optimiser step re-prepare and add up;
    (iii) an optimiser step (torch's Adam through the version counters, FlatAdam through its own bump) ends the step;
    (iv) nothing is kept under no_grad or in eval mode -- a weight swap through `param.data` (torch_ema's copy_to around the
    reference's evaluation, morpheus.py:1297-1301) moves no version counter and must still be seen;
    (v) a new parameter set through train() / eval() ends the step."""
    from morpheus_amd import harness, model as mm
    from morpheus_amd.optim import FlatAdam
    torch.manual_seed(5)
    x = (torch.rand(3000, 3, device=DEV) - 0.5) * 1.6
    t = torch.full((1, 1), 0.125, device=DEV).expand(3000, 1)

    def two_calls(model):
        a = model.density(x[:2000], t[:2000])
        b = model.density(x[1000:], t[1000:])
        return (a["sdf"].square().mean() + a["albedo"].mean() + b["sigma"].mean() * 1e-3 + b["sdf"].mean())

    def grads(model):
        return {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}

    model = harness.build_model("b", DEV).train()
    assert mm.IMPLICIT_OPERANDS
    # (i)
    loss = two_calls(model)
    sc = model._stepcache
    assert sc is not None and not sc.stale and ("warp", True) in sc.entries and ("field", True) in sc.entries
    opnd = sc.entries[("field", True)][0]
    model.zero_grad()
    loss.backward()
    assert sc.stale and not sc.entries                # (ii) the backward pass reached the packs: the step is over, its operands dropped
    g_shared = grads(model)
    mm.IMPLICIT_OPERANDS = False
    try:
        model.zero_grad()
        loss2 = two_calls(model)
        assert model._stepcache is sc                 # untouched: nothing was kept
        loss2.backward()
        g_per_call = grads(model)
    finally:
        mm.IMPLICIT_OPERANDS = True
    assert float(loss) == float(loss2) and set(g_shared) == set(g_per_call)
    for k in g_shared:
        rel = float((g_shared[k] - g_per_call[k]).norm() / g_per_call[k].norm().clamp_min(1e-30))
        assert rel <= 2e-5, (k, rel)                   # the same terms; the shared pack sums the two calls' raw gradients first
    # (ii) two forward / backward pairs before one step: the second forward re-prepares (its own pack), the gradients add up
    model.zero_grad()
    two_calls(model).backward()
    l_again = two_calls(model)
    assert model._stepcache is not sc and model._stepcache.entries[("field", True)][0] is not opnd
    l_again.backward()
    assert model._stepcache.stale and not model._stepcache.entries       # nothing spent lingers in the model
    for k, g in grads(model).items():
        rel = float((g - 2 * g_shared[k]).norm() / g_shared[k].norm().clamp_min(1e-30))
        assert rel <= 2e-5, (k, rel)
    # (iii) optimiser steps
    for make in (lambda: torch.optim.Adam(model.parameters(), lr=1e-3), lambda: FlatAdam(model.get_params_all(1e-3), betas=(0.9, 0.99), eps=1e-15)):
        opt = make()
        opt.zero_grad()
        l0 = two_calls(model)
        l0.backward()
        opt.step()
        l1 = two_calls(model)                          # a forward that is never run backward: its operands stay in the model ...
        before = model._stepcache
        assert not before.stale
        mm.IMPLICIT_OPERANDS = False
        try:
            opt.zero_grad()
            two_calls(model).backward()                # (gradients from a graph of its own)
        finally:
            mm.IMPLICIT_OPERANDS = True
        assert model._stepcache is before and not before.stale
        opt.step()                                    # ... until the parameters move under them
        l2 = two_calls(model)
        assert model._stepcache is not before and float(l2) != float(l1) and float(l1) != float(l0)
        model.zero_grad()
    # (iv) no_grad / eval: nothing kept across calls; a `.data` swap is seen
    model.eval()
    assert model._stepcache is None
    with torch.no_grad():
        s0 = model.density(x, t)["sdf"].clone()
        assert model._stepcache is None
        saved = [p.data.clone() for p in model.sdf_net.parameters()]
        for p in model.sdf_net.parameters():
            p.data.copy_(p.data * 1.01)
        s1 = model.density(x, t)["sdf"].clone()
        for p, q in zip(model.sdf_net.parameters(), saved):
            p.data.copy_(q)
        s2 = model.density(x, t)["sdf"]
    assert float((s1 - s0).abs().max()) > 1e-4 and torch.equal(s2, s0)
    # (v)
    model.train()
    two_calls(model)
    assert model._stepcache is not None
    model.eval()
    assert model._stepcache is None


def test_field_queries_accumulate_gradients_in_place():
    """ops._QueryAccumulator: the six field queries of a real-view step add their weight / beta / table gradients into ONE tensor
    each inside their kernels (first query returns it to autograd, later ones return None) -- the same gradients as one tensor per
    query summed by autograd, with fewer launches."""
    from torch.utils._python_dispatch import TorchDispatchMode
    from morpheus_amd import harness, ops
    from bench_support import trainstep
    from morpheus_amd.occgrid import OccupancyGrid
    from morpheus_amd.render import HotPathRenderer

    class CountAdds(TorchDispatchMode):
        def __init__(self):
            super().__init__()
            self.big = 0

        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = getattr(func, "__name__", "")
            if name.startswith(("add", "zeros_like", "zero_")) and args and isinstance(args[0], torch.Tensor) and args[0].numel() >= 24928:
                self.big += 1
            return func(*args, **(kwargs or {}))

    out, seen = {}, {}
    for in_place in (True, False):
        ops.ACCUMULATE_IN_PLACE = in_place
        try:
            model = harness.build_model("b", DEV).train()
            grid = OccupancyGrid([-model.bound] * 3 + [model.bound] * 3, 128).to(DEV)
            rend = HotPathRenderer(model, model.config, grid, 200)
            ts = trainstep.RealViewTrainStep(rend, trainstep.make_frames([25], 64, 64, DEV), ray_num=512)
            ts.epoch = 1000
            torch.manual_seed(11)
            with torch.no_grad():
                trainstep.warm_up_occupancy(ts)
            ts.global_step = 4096 + 3
            for p in model.parameters():
                p.grad = None
            loss = ts(frame_index=0)
            cnt = CountAdds()
            with cnt:
                loss.backward()
            out[in_place] = (float(loss), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
            seen[in_place] = cnt.big
        finally:
            ops.ACCUMULATE_IN_PLACE = True
    (l1, g1), (l0, g0) = out[True], out[False]
    assert abs(l1 - l0) <= 1e-6 * abs(l0) and set(g1) == set(g0)     # (the forward's atomic loss sums may differ in the last bit)
    for k in g0:
        rel = float((g1[k] - g0[k]).norm() / g0[k].norm().clamp_min(1e-30))
        assert rel <= 2e-5, (k, rel)                      # same terms, summed in another order
    assert torch.count_nonzero(g1["encoder.embeddings"]) > 0 and torch.count_nonzero(g1["sdf2density.beta"]) == 1
    assert seen[True] + 15 <= seen[False], seen           # table fills + table adds + token adds that no longer exist


def test_two_models_with_their_own_arithmetic_in_one_process():
    """`scene_representation.mlp_mode` binds the arithmetic form to a model's operand packs: two models of one process run
    different forms side by side, and changing the process default between a forward and its backward changes nothing."""
    from morpheus_amd import harness, ops
    hw, S = 16, 32
    o, d, t, rid = [v.to(DEV) for v in synth.frame_rays(25, hw, hw)]
    N = o.shape[1]
    timg, tdep = [v.to(DEV) for v in synth.targets(N)]
    res, grads = {}, {}
    models = {m: harness.build_model("b", DEV).eval() for m in ("b3", "f32")}
    for m, model in models.items():
        model.mlp_mode = m
    rends = {m: harness.make_renderer(model, S, jitter=synth.ray_jitter(N).to(DEV)) for m, model in models.items()}
    outs = {m: rends[m].render_rays(o, d, t, rid, hw, hw, ambient_ratio=1.0, shading="albedo") for m in models}   # both graphs alive
    for m, model in models.items():                      # a changed default must not reach the packs already prepared
        prev = ops.set_mlp_mode("f32" if m == "b3" else "b3")
        try:
            harness.bench_loss(outs[m], timg, tdep).backward()
            grads[m] = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
        finally:
            ops.set_mlp_mode(prev)
    assert not torch.equal(outs["b3"]["sdf"], outs["f32"]["sdf"])          # two arithmetic forms really ran ...
    assert_close(outs["b3"]["image"], outs["f32"]["image"], 1e-5, "image b3 vs f32", floor=FLOOR)   # ... and agree to fp32 round-off
    for k in grads["f32"]:
        rel = float((grads["b3"][k] - grads["f32"][k]).norm() / grads["f32"][k].norm().clamp_min(1e-30))
        assert rel <= 1e-3, (k, rel)


@pytest.mark.parametrize("tag", ["use_t", "no_joint", "use_t_no_joint", "use_app", "encode_topo", "no_color_grid",
                                 "app_topo_freqcolor_no_joint"])
def test_model_switch_variants_vs_reference_goldens(tag):
    """The constructor switches of models/model.py:36-53 that no shipped YAML sets but a caller may: use_t=True (13 time-encoding
    columns between position encoding and deform code: folded into the per-frame first-layer bias) and use_joint=False (raw x in
    front of sdf_net: the encoding's sin / cos columns get zero weights) on the fused kernels; use_app=True, encode_topo=True and
    color_grid=False (the field nets read other per-point inputs) on the composed field path.  forward() / warp() -- and for the
    composed path normal() and a canonical density() -- on 1024 probe points at two frame times against fixtures from the
    reference's own model built with the same switches (oracle/make_golden.py:gen_variants)."""
    from morpheus_amd import harness
    from morpheus_amd.model import scene_representation
    sw = {"use_t": dict(use_t=True, use_joint=True), "no_joint": dict(use_t=False, use_joint=False),
          "use_t_no_joint": dict(use_t=True, use_joint=False), "use_app": dict(use_app=True), "encode_topo": dict(encode_topo=True),
          "no_color_grid": dict(color_grid=False),
          "app_topo_freqcolor_no_joint": dict(use_app=True, encode_topo=True, color_grid=False, use_joint=False)}[tag]
    full = dict(use_t=False, use_joint=True, use_app=False, encode_topo=False, color_grid=True)
    full.update(sw)
    composed = full["use_app"] or full["encode_topo"] or not full["color_grid"]
    g = load_golden("variants.npz")
    n = 1024
    x = synth.hash_tensor((n, 3), 360, 1.15).to(DEV)
    t = torch.where(torch.arange(n)[:, None] % 2 == 0, torch.tensor(37 / 200), torch.tensor(0.615)).to(DEV)
    cfg = harness.load_config()
    for kind in ("a", "b"):
        for ml_tag, ml in (("full", None), ("half", 0.5)):
            model = scene_representation(cfg, 1.01, num_frames=200, deform_dim=16, amb_dim=2, **full)
            model.load_state_dict(synth.variant_state(kind, 200, **sw), strict=True)
            model.max_level = ml
            model = model.to(DEV).eval()
            assert model.composed_field == composed
            model.zero_grad()
            sdf, sig, col, _, dfm, _ = model(x, t, None, ratio=1.0, shading="albedo", cano=False)
            key = f"{tag}_{kind}_{ml_tag}"
            assert_close(sdf, g[key + "|sdf"], TOL, key + " sdf", floor=FLOOR)
            assert_close(sig, g[key + "|sigma"], 1e-3, key + " sigma (x10 gain on sdf round-off)", floor=FLOOR)
            assert_close(col, g[key + "|color"], TOL, key + " color", floor=FLOOR)
            assert_close(dfm, g[key + "|deform"], TOL, key + " deform", floor=1e-3)
            assert_close(model.warp(x, t)[1], g[key + "|topo"], TOL, key + " topo", floor=1e-3)
            if composed:
                with torch.no_grad():
                    assert_close(model.normal(x, t)[1], g[key + "|normal_raw"], 5e-3, key + " normal_raw (FD, x250 gain)", floor=5e-2)
                    dc = model.density(x, cano=True)
                assert_close(dc["sdf"], g[key + "|cano_sdf"], TOL, key + " canonical sdf", floor=FLOOR)
                assert_close(dc["albedo"], g[key + "|cano_albedo"], TOL, key + " canonical albedo", floor=FLOOR)
            if ml is None:
                ((col ** 2).sum() + 0.01 * (sig ** 2).mean() + (sdf ** 2).sum() + (dfm ** 2).sum()).backward()
                n_ok = grad_digest_check({k: p.grad for k, p in model.named_parameters() if p.grad is not None}, g, key, 5e-3)
                assert n_ok >= 10, n_ok


def test_composed_field_path_equals_fused_path_through_a_training_render():
    """The composed field path (use_app / encode_topo / color_grid=False: model._sigma_albedo_composed) inside render_rays'
    training block -- FD normals, normal_smooth_3d, normal_smoothness, code_reg, pose optimisation.  A model with use_app=True
    and encode_topo=True whose extra first-layer columns are ZERO computes the shipped model's function (same weights otherwise),
    so its real-view training render, loss terms and gradients must equal the fused kernels' -- through rocBLAS instead of the
    MFMA kernels -- and the appearance code gets an exactly-zero gradient."""
    from morpheus_amd import harness
    from bench_support import trainstep
    from morpheus_amd.model import scene_representation
    from tests.util import DrawInjector
    hw, S, N = 16, 32, 96
    o, d, t, rid = [v[:, :N] for v in synth.frame_rays(25, hw, hw)]
    smp = of.uniform_samples(o[0], d[0], synth.ray_jitter(hw * hw)[:N], S, 1.01)
    frame = trainstep.make_frames([25], hw, hw, DEV)[0]
    data = trainstep.sample_real_view_rays(frame, N, torch.arange(N, device=DEV))
    cfg = harness.load_config()
    base = synth.make_state("b", 200)
    out = {}
    for tag in ("fused", "composed"):
        if tag == "fused":
            model = harness.build_model("b", DEV, 0.75)
        else:
            model = scene_representation(cfg, 1.01, num_frames=200, deform_dim=16, amb_dim=2, use_t=False, use_joint=True,
                                         color_grid=True, use_app=True, encode_topo=True)
            sd = synth.variant_state("b", 200, use_app=True, encode_topo=True)
            w0 = torch.zeros(64, 39 + 32 + 18)
            w0[:, :39 + 32 + 2] = base["sdf_net.net.0.weight"]                 # [enc(x) | hash | topo(2) | its 16 sin / cos: 0]
            sd["sdf_net.net.0.weight"] = w0
            v = torch.zeros(64, 64 + 48)
            v[:, :64] = base["color_net.net.0.weight_v"]                       # [hash_c | geo | app code: 0]
            sd["color_net.net.0.weight_v"], sd["color_net.net.0.weight_g"] = v, base["color_net.net.0.weight_g"]
            model.load_state_dict(sd, strict=True)
            model.max_level = 0.75
            model = model.to(DEV)
            assert model.composed_field
        model.train()
        rend = harness.make_renderer(model, S, samples=tuple(v.to(DEV) for v in smp))
        with DrawInjector():
            res = rend.render_rays(o.to(DEV), d.to(DEV), t.to(DEV), rid.to(DEV), N, 1, ambient_ratio=1.0,
                                   light_d=of.safe_normalize(o[0] + 0.3).to(DEV), shading="albedo_normal", real_view=True,
                                   cano=False, rays_depth=data["depth"].view(1, -1, 1), rays_mask=data["mask"].view(1, -1, 1),
                                   optimize_pose=True)
        terms = {k: res[k] for k in ("loss_code", "sdf_loss", "fs_loss", "loss_normal_perturb", "normal_reg")}
        loss = (res["image"] ** 2).mean() + res["depth"].mean() + sum(terms.values())
        model.zero_grad()
        loss.backward()
        out[tag] = (res, terms, {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
    (rf, tf, gf), (rc, tc, gc) = out["fused"], out["composed"]
    assert_close(rc["image"], rf["image"], TOL, "image", floor=FLOOR)
    assert_close(rc["depth"], rf["depth"], TOL, "depth", floor=DEPTH_FLOOR)
    assert_close(rc["sdf"], rf["sdf"], TOL, "sdf", floor=FLOOR)
    for k in tf:
        assert_close(tc[k], tf[k], 2e-2 if "normal" in k else 1e-3, k)
    assert float(gc["app_code.volumes.0"].abs().max()) == 0.0 and set(gf) <= set(gc)
    for k in gf:
        a, b = gc[k], gf[k]
        if k == "sdf_net.net.0.weight":
            a = a[:, :73]
        if k == "color_net.net.0.weight_v":
            a = a[:, :64]
        rel = float((a - b).norm() / b.norm().clamp_min(1e-30))
        assert rel <= 5e-3, (k, rel)          # FD-normal terms amplify the two arithmetic forms' round-off
