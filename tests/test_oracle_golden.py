"""CPU: the oracle restatement (oracle/field.py) against reference-generated goldens.

The goldens hold OUTPUTS of the reference's own Python (models.model.scene_representation and
the unmodified MorpheuS.render_rays) run in the build container by oracle/make_golden.py; inputs
are regenerated here from morpheus_amd.synth.  This pins the oracle before it is trusted as the
checker for the HIP path.
"""
import numpy as np
import pytest
import torch

from morpheus_amd import synth
from oracle import field as of
from tests.util import assert_close, grad_digest_check, load_golden, probe_points

TOL = 5e-5   # torch-vs-torch on CPU, floor 1e-3: round-off only (different GEMM summation orders)


def leaf_state(kind):
    st = synth.make_state(kind)
    return {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in st.items()}


def test_operators():
    g = load_golden("operators.npz")
    x = probe_points(64, 310, 1.5)
    for tag, ml in (("none", None), ("050", 0.5), ("075", 0.75)):
        assert_close(of.freq_encode(x, 6, ml), g[f"freq_{tag}"], 1e-6, f"freq {tag}")
    assert_close(of.freq_encode(x[:, :1], 6, 0.5), g["freq1_050"], 1e-6, "freq1")
    st = synth.make_state("b")
    vols = [st[f"deform_code.volumes.{k}"] for k in range(3)]
    tt = torch.tensor([[0.0], [7 / 200], [0.5], [199 / 200], [1.3], [-0.2], [0.123456]])
    assert_close(of.multicode_sample(vols, tt), g["multicode"], 1e-5, "multicode")
    assert_close(of.multicode_sample(vols, tt[1:2]), g["multicode_single"], 1e-5, "multicode n=1")
    p = leaf_state("b")
    xin = synth.hash_tensor((32, 87), 320, 1.0)
    y = of.mlp_apply(xin, p, "deform_net", 6, True)
    (y ** 2).sum().backward()
    assert_close(y, g["mlp_wn_out"], 5e-5, "mlp out")
    assert_close(p["deform_net.net.0.weight_v"].grad, g["mlp_wn_grad_v0"], 1e-4, "mlp dv0", floor=1e-4)
    assert_close(p["deform_net.net.5.weight_g"].grad, g["mlp_wn_grad_g5"], 1e-4, "mlp dg5")
    assert_close(p["deform_net.net.2.bias"].grad, g["mlp_wn_grad_b2"], 1e-4, "mlp db2")
    s = torch.linspace(-1, 1, 41).requires_grad_(True)
    beta = torch.tensor(0.1, requires_grad=True)
    sig = of.laplace_density(s, beta)
    sig.sum().backward()
    assert_close(sig, g["laplace_sigma"], 1e-6, "laplace")
    assert_close(s.grad, g["laplace_dsdf"], 1e-6, "laplace ds")
    assert_close(beta.grad, g["laplace_dbeta"], 1e-5, "laplace dbeta")
    ids = torch.tensor([0, 3, 17, 199])
    assert_close(of.pose_rotation(st["pose_array.data"], ids), g["pose_R"], 1e-6, "pose R")
    assert_close(st["pose_array.data"][:, 3:6][ids], g["pose_t"], 0, "pose t")
    assert_close(of.safe_normalize(torch.cat([x[:8], torch.zeros(1, 3)])), g["safe_normalize"], 1e-6, "safe_norm")


@pytest.mark.parametrize("kind", ["a", "b"])
def test_model_forward_modes(kind):
    g = load_golden("model.npz")
    n = 2048
    x = probe_points(n, 330)
    t = torch.full((n, 1), 37 / 200)
    light = of.safe_normalize(synth.hash_tensor((n, 3), 331, 1.0))
    for ml_tag, ml in (("full", None), ("half", 0.5)):
        for shading in ("albedo", "lambertian", "textureless", "normal"):
            for cano in (False, True):
                if ml is not None and shading in ("textureless", "normal"):
                    continue
                p = leaf_state(kind)
                f = of.OracleField(p, 1.01, ml)
                sdf, sig, col, nrm, dfm, raw = f.forward(x, t, light, ratio=0.3, shading=shading, cano=cano)
                key = f"{kind}_{ml_tag}_{shading}_{'cano' if cano else 'deform'}"
                assert_close(sdf, g[key + "|sdf"], TOL, key + " sdf", floor=1e-2)
                assert_close(sig, g[key + "|sigma"], 5e-4, key + " sigma (exp(-sdf/beta): x10 gain on sdf round-off)")
                assert_close(col, g[key + "|color"], TOL if shading == "albedo" else 3e-3, key + " color")
                if nrm is not None:
                    assert_close(raw, g[key + "|normal_raw"], 3e-3, key + " normal_raw (FD: x250 round-off gain)", floor=5e-2)
                if dfm is not None:
                    assert_close(dfm, g[key + "|deform"], TOL, key + " deform")
                if shading in ("albedo", "lambertian") and ml is None:
                    probe = (col ** 2).sum() + 0.01 * (sig ** 2).mean() + (sdf ** 2).sum()
                    probe.backward()
                    n_ok = grad_digest_check({k: v.grad for k, v in p.items() if v.is_floating_point()
                                              and v.grad is not None}, g, key, 2e-4)
                    assert n_ok >= 10
        p = leaf_state(kind)
        f = of.OracleField(p, 1.01, ml)
        d = f.density(x, t)
        assert_close(d["sdf"], g[f"{kind}_{ml_tag}_density|sdf"], TOL, "density sdf", floor=1e-2)
        assert_close(d["albedo"], g[f"{kind}_{ml_tag}_density|albedo"], TOL, "density albedo")
        assert_close(f.normal(x, t)[1], g[f"{kind}_{ml_tag}_normal_warped|raw"], 3e-3, "normal raw", floor=5e-2)
        assert_close(f.warp(x, t)[1], g[f"{kind}_{ml_tag}_warp|topo"], TOL, "topo")


@pytest.mark.parametrize("kind", ["a", "b"])
@pytest.mark.parametrize("case", ["cfg1", "cfg3head"])
def test_render_rays(kind, case):
    g = load_golden("render.npz")
    hw, S, nray = {"cfg1": (32, 64, None), "cfg3head": (128, 128, 256)}[case]
    o, d, t, rid = synth.frame_rays(25, hw, hw)
    if nray is not None:
        o, d, t, rid = o[:, :nray], d[:, :nray], t[:, :nray], rid[:, :nray]
    N = o.shape[1]
    samples = of.uniform_samples(o[0], d[0], synth.ray_jitter(N), S, 1.01)
    light = of.safe_normalize(o[0] + torch.tensor([0.3, -0.2, 0.5]))
    cfg_train = dict(ori_weight=0.01, code_reg=0.5, trunc=0.1)
    for mode in ("eval_albedo_deform", "eval_albedo_cano", "eval_lambertian_deform", "train_albedo_deform_pose"):
        p = leaf_state(kind)
        f = of.OracleField(p, 1.01, None)
        train = mode.startswith("train")
        kw = {}
        if train:
            dep = synth.hash_tensor((1, N, 1), 400, 0.3, 1.5)
            msk = (synth.hash_tensor((1, N, 1), 401, 0.5, 0.5) > 0.3).float()
            kw = dict(rays_depth=dep, rays_mask=msk, optimize_pose=True, real_view=False)
        res = of.render_rays(f, o, d, t, rid, samples, ambient_ratio=0.3, light_d=light,
                             shading="lambertian" if "lambertian" in mode else "albedo",
                             cano="cano" in mode, training=train, cfg_train=cfg_train, **kw)
        key = f"{kind}_{case}_{mode}"
        assert_close(res["image"], g[key + "|image"], 5e-5, key + " image")
        assert_close(res["depth"], g[key + "|depth"], 5e-5, key + " depth")
        assert_close(res["weights_sum"], g[key + "|weights_sum"], 5e-5, key + " opacity")
        assert_close(res["sdf"][::16], g[key + "|sdf_s16"], 1e-4, key + " sdf (abs err ~1e-6 near the zero crossing)", floor=1e-2)
        assert_close(res["weights"][::16], g[key + "|weights_s16"], 5e-4, key + " weights", floor=1e-3)
        for lk in ("loss_code", "sdf_loss", "fs_loss"):
            if key + "|" + lk in g.files:
                assert_close(res[lk], g[key + "|" + lk], 5e-5, key + " " + lk)
        timg, tdep = synth.targets(N)
        loss = ((res["image"][0] - timg) ** 2).mean() + ((res["depth"][0] - tdep) ** 2).mean()
        if train:
            loss = loss + res["loss_code"] + res["sdf_loss"] + 0.1 * res["fs_loss"]
        loss.backward()
        assert_close(loss, g[key + "|loss"], 5e-5, key + " loss")
        n_ok = grad_digest_check({k: v.grad for k, v in p.items() if v.is_floating_point() and v.grad is not None},
                                 g, key, 3e-4)
        assert n_ok >= 10


def test_reference_fp32_is_itself_off_the_float64_value_by_more_than_1e4():
    """The fact behind the counted parity gate: run in DOUBLE on the same inputs (tests/golden/round4.npz,
    oracle/make_golden.py:gen_round4), the imported reference differs from its OWN fp32 result by more than 1e-4 at SURVEY 8(d)'s
    floor on a few SDF samples next to the zero crossing -- no fp32 implementation can be held to 1e-4 against those fp32 values.
    Rendered RGB / depth / opacity of the reference's fp32 run stay within 6e-5 of the double values."""
    import numpy as np
    from tests.util import load_golden, rel_err
    r, g4 = load_golden("render.npz"), load_golden("round4.npz")
    worst, n_over = 0.0, 0
    for kind in "ab":
        for case in ("cfg1", "cfg3head"):
            for mode in ("eval_albedo_deform", "eval_albedo_cano"):
                key = f"{kind}_{case}_{mode}"
                e = rel_err(r[key + "|sdf_s16"], g4[key + "|f64|sdf_s16"])
                worst, n_over = max(worst, float(e.max())), n_over + int((e > 1e-4).sum())
                for q in ("image", "depth", "weights_sum"):
                    assert float(rel_err(r[key + "|" + q], g4[key + "|f64|" + q]).max()) < 1e-4, (key, q)
                assert g4[key + "|f64|sdf_s16"].dtype == np.float64
    assert 1e-4 < worst < 6e-4 and 3 <= n_over <= 20, (worst, n_over)


def test_round5_fixture_facts():
    """tests/golden/round5.npz (oracle/make_golden.py:gen_round5), the reference-side facts the GPU tests lean on, on the CPU:
    * the step-composition fixture: group names and learning rates as the reference's own update_learning_rate / freeze_lr_deform set
      them (pose at a tenth; the three deformation groups at 0 while frozen), frozen groups' deltas exactly 0 after the first step;
    * the 24 x 24 virtual-view training step that the reference ran in fp32 AND in float64: the oracle's lambertian image agrees
      with the reference's fp32 one, and the reference's own fp32 gradients are up to 1.3e-2 (median ~1e-3) off its double run --
      the allowance the HIP-side gate is derived from."""
    g = load_golden("round5.npz")
    for variant in ("accum", "freeze"):
        names = [str(n) for n in g[variant + "|group_names"]]
        lr = dict(zip(names, g[variant + "|group_lr"]))
        assert {"pose", "encoder_sdf", "code_deform", "decoder_deform", "decoder_topo"} <= set(names)
        assert abs(lr["pose"] - 0.1 * lr["encoder_sdf"]) < 1e-15 and lr["decoder_deform"] == lr["encoder_sdf"] > 0
    frozen = dict(zip([str(n) for n in g["freeze|group_names"]], g["freeze|group_lr_frozen"]))
    assert all(frozen[k] == 0.0 for k in ("code_deform", "decoder_deform", "decoder_topo")) and frozen["encoder_sdf"] > 0
    for k in g.files:
        if k.startswith("freeze|delta1|") and k.endswith("|norm") and any(s in k for s in ("deform_", "topo_net", "pose_array")):
            assert float(g[k]) == 0.0, k
    assert float(g["freeze|delta|deform_net.net.2.weight_v|norm"]) > 0 and float(g["accum|delta|pose_array.data|norm"]) > 0
    # virt24: oracle image vs the reference's fp32 image; reference fp32 vs float64 gradients
    hw, S = 24, 24
    o, d = synth.camera_rays(hw, hw, synth.look_at_pose(70.0, 35.0, 1.5))
    N = o.shape[0]
    samples = of.uniform_samples(o, d, synth.ray_jitter(N), S, 1.01)
    light = of.safe_normalize(o + torch.tensor([0.3, -0.2, 0.5]))
    f = of.OracleField(synth.make_state("b"), 1.01, 0.75)
    with torch.no_grad():
        res = of.render_rays(f, o[None], d[None], torch.full((1, N, 1), 140 / 200), torch.full((1, N, 1), 140, dtype=torch.int64), samples,
                             ambient_ratio=0.55, light_d=light, shading="lambertian", bg_color=torch.tensor([0.2, 0.5, 0.7]))
    assert_close(res["image"], g["virt24|f32|image"], 5e-3, "oracle lambertian image vs the reference's (FD normals)", floor=1e-2)
    assert g["virt24|f64|image"].dtype == np.float64
    errs = []
    for k in g.files:
        if k.startswith("virt24|f64|grad|") and k.endswith("|samples"):
            s64, s32 = g[k].astype(np.float64), g[k.replace("|f64|", "|f32|")].astype(np.float64)
            if np.abs(s64).max() > 0:
                errs.append(float(np.abs(s32 - s64).max() / np.abs(s64).max()))
    assert len(errs) >= 45 and 5e-3 < max(errs) < 3e-2 and 2e-4 < sorted(errs)[len(errs) // 2] < 3e-3, (max(errs), sorted(errs)[len(errs) // 2])


def test_round6_fixture_facts():
    """tests/golden/round6.npz (oracle/make_golden.py:gen_round6): the 72 x 72 x 32-sample virtual-view training step that the reference
    ran in fp32 AND in float64 -- the yardstick of test_virtual_view_gradients_72_against_the_reference_in_double.  On the CPU: both
    precisions took the same number of injected draws, the float64 arrays are float64, the two losses agree to fp32 round-off, and the
    reference's own fp32 gradients are up to 5.5e-3 (median 7e-4) off its double run on the 57 parameter tensors with a gradient."""
    g = load_golden("round6.npz")
    assert int(g["virt72d|f32|n_draws"]) == int(g["virt72d|f64|n_draws"]) >= 1
    assert g["virt72d|f64|image"].dtype == np.float64 and g["virt72d|f64|image"].reshape(-1, 3).shape[0] == 72 * 72
    l64, l32 = float(g["virt72d|f64|loss"]), float(g["virt72d|f32|loss"])
    assert abs(l32 - l64) < 1e-5 * abs(l64), (l32, l64)
    errs = []
    for k in g.files:
        if k.startswith("virt72d|f64|grad|") and k.endswith("|samples"):
            s64, s32 = g[k].astype(np.float64), g[k.replace("|f64|", "|f32|")].astype(np.float64)
            if np.abs(s64).max() > 0:
                errs.append(float(np.abs(s32 - s64).max() / np.abs(s64).max()))
    assert len(errs) >= 50 and 1e-3 < max(errs) < 2e-2 and 1e-4 < sorted(errs)[len(errs) // 2] < 3e-3, (max(errs), sorted(errs)[len(errs) // 2])
