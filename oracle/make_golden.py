"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.npz.

Runs ONLY in the build container (needs /root/reference).  It imports the
reference's own Python -- models.model.scene_representation and the unmodified
morpheus.MorpheuS.render_rays -- on CPU behind the shim set of SURVEY.md 8(c),
feeds it closed-form inputs/weights from morpheus_amd.synth and stores the
reference's OUTPUTS as small fixtures.  Inputs are never stored: tests
regenerate them bit-identically from the same closed-form generators.

No reference source travels: the fixtures are numeric arrays only.

Shims (harness side only; reference files untouched, nothing written there):
  1. sys.dont_write_bytecode
  2. stub modules for absent imports (cv2, mcubes, imageio, trimesh, open3d,
     torchmetrics, clip, pyrender, torch_ema, omegaconf ...)
  3. `datasets` pre-registered as a namespace pointing at the reference dir
     (the installed HuggingFace `datasets` would otherwise win)
  4. external.encoders.gridencoder.grid.GridEncoder := oracle.hashgrid.OracleGridEncoder
     (the real one is CUDA-only and its JIT build writes into the source tree)
  5. a pure-torch `nerfacc` exposing render_weight_from_density /
     accumulate_along_rays / OccGridEstimator(.sampling returns preset samples)
  6. torch.Tensor.cuda -> identity (models/density.py:20 calls .cuda())
  7. a fake `self` for the unbound MorpheuS.render_rays

Usage:  python -m oracle.make_golden        (from the repo root)
"""
from __future__ import annotations

import os
import sys
import types

sys.dont_write_bytecode = True

import numpy as np
import torch

REF = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

from morpheus_amd import synth  # noqa: E402
from oracle import field as ofield  # noqa: E402
from oracle.hashgrid import OracleGridEncoder  # noqa: E402


# ----------------------------------------------------------------------------- shims
class _Anything(types.ModuleType):
    """Module stub: any attribute is another permissive stub (callable, subclassable)."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        v = type(name, (), {"__init__": lambda self, *a, **k: None, "__call__": lambda self, *a, **k: None})
        setattr(self, name, v)
        return v


def _stub(name):
    m = _Anything(name)
    m.__path__ = []
    sys.modules[name] = m
    return m


class _PresetSampler:
    """Stands in for nerfacc.OccGridEstimator: .sampling() returns the samples handed to it."""

    def __init__(self, *a, **k):
        self.samples = None

    def sampling(self, rays_o, rays_d, **kw):
        ri, ts, te = self.samples
        return ri, ts, te


def install_shims():
    for n in ["cv2", "mcubes", "imageio", "trimesh", "open3d", "torchmetrics", "clip", "pyrender",
              "torch_ema", "omegaconf", "kornia", "pytorch_lightning", "taming", "diffusers",
              "tools", "tools.culling", "tools.vis", "tools.pose_utils"]:
        _stub(n)
    # nerfacc shim (semantics: SURVEY C.8) -- deliberately the cumsum formulation of oracle.field
    nf = types.ModuleType("nerfacc")
    nf.OccGridEstimator = _PresetSampler

    def render_weight_from_density(t_starts, t_ends, sigmas, ray_indices=None, n_rays=None, **kw):
        return ofield.render_weights(t_starts, t_ends, sigmas, ray_indices, n_rays)

    def accumulate_along_rays(weights, values=None, ray_indices=None, n_rays=None):
        return ofield.accumulate(weights, values, ray_indices, n_rays)

    nf.render_weight_from_density = render_weight_from_density
    nf.accumulate_along_rays = accumulate_along_rays
    sys.modules["nerfacc"] = nf
    # datasets namespace -> reference dir
    ds = types.ModuleType("datasets")
    ds.__path__ = [os.path.join(REF, "datasets")]
    sys.modules["datasets"] = ds
    # hash-grid stub
    for n in ["external", "external.encoders", "external.encoders.gridencoder"]:
        m = types.ModuleType(n)
        m.__path__ = []
        sys.modules[n] = m
    g = types.ModuleType("external.encoders.gridencoder.grid")
    g.GridEncoder = OracleGridEncoder
    sys.modules["external.encoders.gridencoder.grid"] = g
    torch.Tensor.cuda = lambda self, *a, **k: self
    if REF not in sys.path:
        sys.path.insert(0, REF)


def ref_config():
    import yaml
    with open(os.path.join(REF, "configs", "snoopy.yaml")) as f:
        return yaml.safe_load(f)


def build_ref_model(state, max_level=None, **switches):
    """switches: overrides of the YAML's model switches (use_t, use_joint: gen_variants)"""
    from models.model import scene_representation
    cfg = ref_config()
    sw = {k: cfg["model"][k] for k in ("use_app", "use_t", "color_grid", "use_joint", "encode_topo")}
    sw.update(switches)
    m = scene_representation(cfg, 1.01, num_frames=200, deform_dim=cfg["model"]["deform_dim"], amb_dim=cfg["model"]["amb_dim"],
                             **sw)
    missing, unexpected = m.load_state_dict(state, strict=False)
    assert not unexpected, unexpected
    assert all("res_tab" in k for k in missing), missing
    m.max_level = max_level
    return m, cfg


# ----------------------------------------------------------------------------- helpers
def probe_points(n, stream=300, scale=1.15):
    """Points in [-scale, scale]^3: mostly inside the +-1.01 box, some outside (OOB semantics)."""
    return synth.hash_tensor((n, 3), stream, scale)


def grad_digest(named_grads):
    """Compact gradient record: per-tensor L2 norm, sum, and 64 strided samples."""
    out = {}
    for k, g in named_grads.items():
        g = g.detach().reshape(-1).double()
        idx = torch.linspace(0, g.numel() - 1, min(64, g.numel())).long()
        out[k + "|norm"] = np.float64(g.norm().item())
        out[k + "|sum"] = np.float64(g.sum().item())
        out[k + "|samples"] = g[idx].float().numpy()
    return out


def npf(t):
    return None if t is None else t.detach().float().numpy()


# ----------------------------------------------------------------------------- generators
def gen_operators():
    from models.encodings import FreqEncoder_torch
    from models.deform_code import MultiCode
    from models.decoders import MLP
    from models.density import LaplaceDensity
    from models.pose import PoseArray
    from utils import safe_normalize
    g = {}
    x = probe_points(64, 310, 1.5)
    enc = FreqEncoder_torch(input_dim=3, max_freq_log2=5, N_freqs=6)
    for tag, ml in (("none", None), ("050", 0.5), ("075", 0.75)):
        g[f"freq_{tag}"] = npf(enc(x, max_level=ml))
    enc1 = FreqEncoder_torch(input_dim=1, max_freq_log2=5, N_freqs=6)
    g["freq1_050"] = npf(enc1(x[:, :1], max_level=0.5))

    st = synth.make_state("b")
    mc = MultiCode([25, 50, 200], 16)
    for k in range(3):
        mc.volumes[k].data.copy_(st[f"deform_code.volumes.{k}"])
    tt = torch.tensor([[0.0], [7 / 200], [0.5], [199 / 200], [1.3], [-0.2], [0.123456]])
    g["multicode"] = npf(mc.sample(tt))
    g["multicode_single"] = npf(mc.sample(tt[1:2]))

    mlp = MLP(87, 3, 128, 6, bias=True)
    sd = {k[len("deform_net."):]: v for k, v in st.items() if k.startswith("deform_net.")}
    mlp.load_state_dict(sd)
    xin = synth.hash_tensor((32, 87), 320, 1.0)
    y = mlp(xin)
    (y ** 2).sum().backward()
    g["mlp_wn_out"] = npf(y)
    g["mlp_wn_grad_v0"] = npf(mlp.net[0].weight_v.grad)
    g["mlp_wn_grad_g5"] = npf(mlp.net[5].weight_g.grad)
    g["mlp_wn_grad_b2"] = npf(mlp.net[2].bias.grad)

    dens = LaplaceDensity({"beta": 0.1})
    s = torch.linspace(-1, 1, 41).requires_grad_(True)
    sig = dens(s)
    sig.sum().backward()
    g["laplace_sigma"] = npf(sig)
    g["laplace_dsdf"] = npf(s.grad)
    g["laplace_dbeta"] = npf(dens.beta.grad)

    pa = PoseArray(200)
    pa.data.data.copy_(st["pose_array.data"])
    ids = torch.tensor([0, 3, 17, 199])
    g["pose_R"] = npf(pa.get_rotation_matrices(ids))
    g["pose_t"] = npf(pa.get_translations(ids))
    g["safe_normalize"] = npf(safe_normalize(torch.cat([x[:8], torch.zeros(1, 3)])))
    np.savez_compressed(os.path.join(OUT, "operators.npz"), **g)
    print("operators.npz", len(g), "arrays")


def gen_model():
    """forward() in every shading mode x cano x max_level on 2048 probe points, both weight states,
    plus gradient digests of a scalar probe."""
    g = {}
    n = 2048
    x = probe_points(n, 330)
    t = torch.full((n, 1), 37 / 200)
    light = ofield.safe_normalize(synth.hash_tensor((n, 3), 331, 1.0))
    for kind in ("a", "b"):
        st = synth.make_state(kind)
        for ml_tag, ml in (("full", None), ("half", 0.5)):
            m, _ = build_ref_model(st, ml)
            m.eval()
            for shading in ("albedo", "lambertian", "textureless", "normal"):
                for cano in (False, True):
                    if ml is not None and shading in ("textureless", "normal"):
                        continue
                    m.zero_grad()
                    sdf, sig, col, nrm, dfm, raw = m(x, t, light, ratio=0.3, shading=shading, cano=cano)
                    key = f"{kind}_{ml_tag}_{shading}_{'cano' if cano else 'deform'}"
                    g[key + "|sdf"], g[key + "|sigma"], g[key + "|color"] = npf(sdf), npf(sig), npf(col)
                    if nrm is not None:
                        g[key + "|normal"], g[key + "|normal_raw"] = npf(nrm), npf(raw)
                    if dfm is not None:
                        g[key + "|deform"] = npf(dfm)
                    if shading in ("albedo", "lambertian") and ml is None:
                        probe = (col ** 2).sum() + 0.01 * (sig ** 2).mean() + (sdf ** 2).sum()
                        probe.backward()
                        gd = grad_digest({k: p.grad for k, p in m.named_parameters() if p.grad is not None})
                        for kk, v in gd.items():
                            g[key + "|grad|" + kk] = v
            # density() / normal() / warp() entry points
            d = m.density(x, t)
            g[f"{kind}_{ml_tag}_density|sdf"], g[f"{kind}_{ml_tag}_density|albedo"] = npf(d["sdf"]), npf(d["albedo"])
            nn_, raw = m.normal(x, t)
            g[f"{kind}_{ml_tag}_normal_warped|raw"] = npf(raw)
            dfm, topo, _ = m.warp(x, t)
            g[f"{kind}_{ml_tag}_warp|topo"] = npf(topo)
    np.savez_compressed(os.path.join(OUT, "model.npz"), **g)
    print("model.npz", len(g), "arrays")


def gen_render():
    """Unmodified MorpheuS.render_rays (eval + deterministic training extras) on cfg1 =
    1 frame, 32x32 rays, S=64 fixed samples; and the first 256 rays of the cfg2/3 set (S=128)."""
    import morpheus as ref_morpheus
    g = {}
    for kind in ("a", "b"):
        st = synth.make_state(kind)
        for case, (hw, S, nray) in (("cfg1", (32, 64, None)), ("cfg3head", (128, 128, 256))):
            o, d, t, rid = synth.frame_rays(25, hw, hw)
            if nray is not None:
                o, d, t, rid = o[:, :nray], d[:, :nray], t[:, :nray], rid[:, :nray]
            N = o.shape[1]
            samples = ofield.uniform_samples(o[0], d[0], synth.ray_jitter(N), S, 1.01)
            light = ofield.safe_normalize(o[0] + torch.tensor([0.3, -0.2, 0.5]))
            for mode in ("eval_albedo_deform", "eval_albedo_cano", "eval_lambertian_deform",
                         "train_albedo_deform_pose"):
                m, cfg = build_ref_model(st, None)
                train = mode.startswith("train")
                m.train(train)
                cfg["train"]["normal_smooth_3d"] = 0.0     # randomised regularisers: next tier
                cfg["train"]["normal_smoothness"] = 0.0
                sampler = _PresetSampler()
                sampler.samples = samples
                fake = types.SimpleNamespace(model=m, occupancy_grid=sampler, config=cfg,
                                             dataset=types.SimpleNamespace(num_frames=200))
                shading = "lambertian" if "lambertian" in mode else "albedo"
                cano = "cano" in mode
                kw = {}
                if train:
                    dep = synth.hash_tensor((1, N, 1), 400, 0.3, 1.5)
                    msk = (synth.hash_tensor((1, N, 1), 401, 0.5, 0.5) > 0.3).float()
                    kw = dict(rays_depth=dep, rays_mask=msk, optimize_pose=True, real_view=False)
                res = ref_morpheus.MorpheuS.render_rays(fake, o, d, t, rid, hw, hw, ambient_ratio=0.3,
                                                        light_d=light, shading=shading, cano=cano, **kw)
                key = f"{kind}_{case}_{mode}"
                g[key + "|image"] = npf(res["image"])
                g[key + "|depth"] = npf(res["depth"])
                g[key + "|weights_sum"] = npf(res["weights_sum"])
                g[key + "|sdf_s16"] = npf(res["sdf"][::16])
                g[key + "|weights_s16"] = npf(res["weights"][::16])
                if res["deform"] is not None:
                    g[key + "|deform_s16"] = npf(res["deform"][::16])
                if res["normal"] is not None:
                    g[key + "|normal_s16"] = npf(res["normal"][::16])
                for lk in ("loss_code", "sdf_loss", "fs_loss", "loss_orient"):
                    if lk in res:
                        g[key + "|" + lk] = npf(res[lk])
                # fwd+bwd digest with the benchmark loss: MSE(image) + MSE(depth)
                timg, tdep = synth.targets(N)
                loss = ((res["image"][0] - timg) ** 2).mean() + ((res["depth"][0] - tdep) ** 2).mean()
                if train:
                    loss = loss + res["loss_code"] + res["sdf_loss"] + 0.1 * res["fs_loss"]
                m.zero_grad()
                loss.backward()
                g[key + "|loss"] = npf(loss)
                gd = grad_digest({k: p.grad for k, p in m.named_parameters() if p.grad is not None})
                for kk, v in gd.items():
                    g[key + "|grad|" + kk] = v
    np.savez_compressed(os.path.join(OUT, "render.npz"), **g)
    print("render.npz", len(g), "arrays")


# ----------------------------------------------------------------------------- round-3 fixtures (variants.npz)
VARIANTS = {"use_t": dict(use_t=True, use_joint=True), "no_joint": dict(use_t=False, use_joint=False),
            "use_t_no_joint": dict(use_t=True, use_joint=False),
            # round 4: the switches that change the field nets' per-point inputs (composed field path of morpheus_amd/model.py)
            "use_app": dict(use_app=True), "encode_topo": dict(encode_topo=True), "no_color_grid": dict(color_grid=False),
            "app_topo_freqcolor_no_joint": dict(use_app=True, encode_topo=True, color_grid=False, use_joint=False)}
ROUND4_VARIANTS = ("use_app", "encode_topo", "no_color_grid", "app_topo_freqcolor_no_joint")


def gen_variants():
    """The model switches no shipped YAML sets but the reference's constructor takes (models/model.py:36-53): use_t=True
    (time encoding next to the deform code), use_joint=False (raw x in front of sdf_net), use_app=True (appearance code behind
    the colour net's input), encode_topo=True (18-column topology encoding) and color_grid=False (frequency-encoded colour
    input), through the reference's own forward() / density() / warp() / normal() on 1024 probe points at two frame times,
    with gradient digests."""
    g = {}
    n = 1024
    x = probe_points(n, 360)
    t = torch.where(torch.arange(n)[:, None] % 2 == 0, torch.tensor(37 / 200), torch.tensor(0.615))
    for tag, sw in VARIANTS.items():
        for kind in ("a", "b"):
            for ml_tag, ml in (("full", None), ("half", 0.5)):
                m, _ = build_ref_model(synth.variant_state(kind, 200, **sw), ml, **sw)
                m.eval()
                m.zero_grad()
                sdf, sig, col, _, dfm, _ = m(x, t, None, ratio=1.0, shading="albedo", cano=False)
                key = f"{tag}_{kind}_{ml_tag}"
                g[key + "|sdf"], g[key + "|sigma"], g[key + "|color"], g[key + "|deform"] = npf(sdf), npf(sig), npf(col), npf(dfm)
                g[key + "|topo"] = npf(m.warp(x, t)[1])
                if tag in ROUND4_VARIANTS:       # + the finite-difference normals through the same field, and a canonical query
                    g[key + "|normal_raw"] = npf(m.normal(x, t)[1])
                    dc = m.density(x, cano=True)
                    g[key + "|cano_sdf"], g[key + "|cano_albedo"] = npf(dc["sdf"]), npf(dc["albedo"])
                if ml is None:
                    probe = (col ** 2).sum() + 0.01 * (sig ** 2).mean() + (sdf ** 2).sum() + (dfm ** 2).sum()
                    probe.backward()
                    for kk, v in grad_digest({k: p.grad for k, p in m.named_parameters() if p.grad is not None}).items():
                        g[key + "|grad|" + kk] = v
    np.savez_compressed(os.path.join(OUT, "variants.npz"), **g)
    print("variants.npz", len(g), "arrays")


# ----------------------------------------------------------------------------- round-2 fixtures (extras.npz)
class DrawInjector:
    """Closed-form stand-ins for torch.rand / rand_like / randn_like while a reference (or HIP) function runs: the k-th
    draw of shape `shape` is synth.hash_tensor(shape, 8000 + k) (uniform: +0.5; normal: x 3.4 ~ unit variance), so the
    reference here and the HIP path on the GPU box see IDENTICAL random perturbations as long as they draw in the same
    order with the same shapes (tests/util.py carries the same class)."""

    def __init__(self, base=8000):
        self.k, self.base = 0, base

    def _next(self, shape, normal, device=None, dtype=None):
        self.k += 1
        v = synth.hash_tensor(tuple(shape), self.base + self.k, 0.5)          # [-0.5, 0.5)
        v = v * 3.4 if normal else v + 0.5
        return v.to(device) if device is not None else v

    def __enter__(self):
        self._saved = (torch.rand, torch.rand_like, torch.randn_like)
        inj = self

        def rand(*size, device=None, dtype=None, **kw):
            size = size[0] if len(size) == 1 and isinstance(size[0], (list, tuple, torch.Size)) else size
            return inj._next(size, False, device)

        torch.rand = rand
        torch.rand_like = lambda t, **kw: inj._next(t.shape, False, t.device)
        torch.randn_like = lambda t, **kw: inj._next(t.shape, True, t.device)
        return self

    def __exit__(self, *a):
        torch.rand, torch.rand_like, torch.randn_like = self._saved


def real_view_case(kind="b", hw=32, S=64, n_keep=96):
    """A real-view training call whose surface-band points all lie inside the 1.1 sphere (so that the reference's
    boolean drop `surf_pts[norm < 1.1]` keeps everything and the shapes of its random draws are data-independent):
    the first n_keep rays of frame 25 whose oracle-rendered opacity exceeds 0.99."""
    o, d, t, rid = synth.frame_rays(25, hw, hw)
    N = o.shape[1]
    st = synth.make_state(kind)
    f = ofield.OracleField({k: v for k, v in st.items()}, 1.01, 0.75)
    smp = ofield.uniform_samples(o[0], d[0], synth.ray_jitter(N), S, 1.01)
    with torch.no_grad():
        r = ofield.render_rays(f, o, d, t, rid, smp, ambient_ratio=1.0, shading="albedo",
                               light_d=ofield.safe_normalize(o[0] + 0.3))
    ok = torch.ones(N, dtype=torch.bool)
    for off in (-0.06, 0.0, 0.07):        # the band is depth + [-trunc/2, trunc/2 + 0.01): keep a margin
        ok &= torch.linalg.norm(o[0] + d[0] * (r["depth"][0][:, None] + off), dim=-1) < 1.08
    sel = torch.nonzero(ok)[:n_keep, 0]
    assert sel.numel() == n_keep, sel.numel()
    return sel


def gen_extras():
    import morpheus as ref_morpheus
    from datasets.utils import get_camera_rays
    from bench_support import trainstep
    g = {}
    # ---- a1: ray generation = get_camera_rays (datasets/utils.py:28-65) + the c2w application of dataset.py:355-366
    for tag, (H, W) in (("sq", (24, 24)), ("rect", (20, 28))):
        fx = fy = torch.tensor(1.2 * W)
        cx, cy = 0.5 * W, 0.5 * H
        rays_d_cam = get_camera_rays(H, W, fx, fy, cx, cy)
        pose = torch.from_numpy(np.stack([synth.look_at_pose(60.0, -40.0), synth.look_at_pose(75.0, 130.0, 1.3)]))
        rays_o = pose[..., None, None, :3, -1].repeat(1, H, W, 1)
        rays_d = torch.sum(rays_d_cam[None, ...].repeat(2, 1, 1, 1)[..., None, :] * pose[:, None, None, :3, :3], -1)
        g[f"raygen_{tag}|rays_o"], g[f"raygen_{tag}|rays_d"] = npf(rays_o), npf(rays_d)
    # ---- a13: background net (models/model.py:400-410)
    dirs = ofield.safe_normalize(synth.hash_tensor((256, 3), 340, 1.0))
    tt = synth.hash_tensor((256, 1), 341, 0.5, 0.5)
    for kind in ("a", "b"):
        for ml_tag, ml in (("full", None), ("half", 0.5)):
            m, _ = build_ref_model(synth.make_state(kind), ml)
            m.zero_grad()
            c = m.background(dirs, tt)
            (c ** 2).sum().backward()
            g[f"bg_{kind}_{ml_tag}|color"] = npf(c)
            g[f"bg_{kind}_{ml_tag}|grad_w0"] = npf(m.bg_net.net[0].weight_v.grad)
            g[f"bg_{kind}_{ml_tag}|grad_b1"] = npf(m.bg_net.net[1].bias.grad)
    # ---- f-2: the in-render regularisers with the random draws INJECTED, and the three caller-side loss groups of the
    #      real-view step (morpheus.py:946-1029, 1090-1145) evaluated by the reference's own methods
    kind, hw, S = "b", 32, 64
    sel = real_view_case(kind, hw, S)
    g["realview|sel"] = sel.numpy().astype(np.int32)
    o, d, t, rid = [v[:, sel] for v in synth.frame_rays(25, hw, hw)]
    N = o.shape[1]
    samples = ofield.uniform_samples(o[0], d[0], synth.ray_jitter(hw * hw)[sel], S, 1.01)
    m, cfg = build_ref_model(synth.make_state(kind), 0.75)
    m.train()
    sampler = _PresetSampler()
    sampler.samples = samples
    fake = types.SimpleNamespace(model=m, occupancy_grid=sampler, config=cfg, dataset=types.SimpleNamespace(num_frames=200),
                                 global_step=1000)
    fake.get_ortho_normal_dir = types.MethodType(ref_morpheus.MorpheuS.get_ortho_normal_dir, fake)
    fake.get_normal_smoothness_loss = types.MethodType(ref_morpheus.MorpheuS.get_normal_smoothness_loss, fake)
    frame = trainstep.make_frames([25], hw, hw, "cpu")[0]
    data = trainstep.sample_real_view_rays(frame, N, sel)
    bg = synth.hash_tensor((N, 3), 350, 0.5, 0.5)
    with DrawInjector() as inj:
        res = ref_morpheus.MorpheuS.render_rays(fake, o, d, t, rid, N, 1, bg_color=bg, ambient_ratio=1.0,
                                                light_d=ofield.safe_normalize(o[0] + 0.3), shading="albedo_normal", real_view=True, cano=False,
                                                rays_depth=data["depth"].view(1, -1, 1), rays_mask=data["mask"].view(1, -1, 1),
                                                optimize_pose=True)
        g["realview|n_draws"] = np.int32(inj.k)
    for lk in ("loss_normal_perturb", "normal_reg", "loss_code", "sdf_loss", "fs_loss"):
        g["realview|" + lk] = npf(res[lk])
    g["realview|image"], g["realview|depth"] = npf(res["image"]), npf(res["depth"])
    g["realview|weights_sum"], g["realview|sdf_s8"] = npf(res["weights_sum"]), npf(res["sdf"][::8])
    g["realview|beta"] = npf(m.sdf2density.get_beta())
    g["realview|normal_s8"] = npf(res["normal"][::8])
    B, H, W = 1, N, 1
    pred_rgb, pred_depth, pred_mask, pred_normal, pred_sdf = ref_morpheus.MorpheuS.get_pred_from_outputs(fake, res, B, H, W)
    fake.device = "cpu"
    gt_rgb, gt_depth, gt_mask = ref_morpheus.MorpheuS.get_gt_from_data(fake, {k: (v.clone() if torch.is_tensor(v) else v)
                                                                          for k, v in data.items()}, bg, B, H, W)
    l_render = ref_morpheus.MorpheuS.get_real_view_render_loss(fake, pred_rgb, pred_depth, pred_mask, gt_rgb, gt_depth,
                                                                gt_mask, data["rays_o"], data["rays_d"])
    l_point = ref_morpheus.MorpheuS.get_real_view_point_loss(fake, gt_rgb, gt_depth, gt_mask, data["rays_o"], data["rays_d"],
                                                              data["rays_t"], res)
    l_reg = ref_morpheus.MorpheuS.get_regularization_loss(fake, res, pred_normal, cano=False)
    g["realview|loss_render"], g["realview|loss_point"], g["realview|loss_reg"] = npf(l_render), npf(l_point), npf(l_reg)
    g["realview|gt_rgb"] = npf(gt_rgb)
    total = l_render + l_point + l_reg
    m.zero_grad()
    total.backward()
    g["realview|loss"] = npf(total)
    for kk, v in grad_digest({k: p.grad for k, p in m.named_parameters() if p.grad is not None}).items():
        g["realview|grad|" + kk] = v
    # ---- virtual-view training call: lambertian shading, orientation loss and the 2-D normal image (accumulated with the
    #      LIVE weights, morpheus.py:775) -- no random draws on this path
    hw2, S2 = 16, 32
    o, d, t, rid = synth.frame_rays(25, hw2, hw2)
    N2 = o.shape[1]
    smp2 = ofield.uniform_samples(o[0], d[0], synth.ray_jitter(N2), S2, 1.01)
    light2 = ofield.safe_normalize(o[0] + torch.tensor([0.3, -0.2, 0.5]))
    m, cfg = build_ref_model(synth.make_state("b"), None)
    m.train()
    cfg["train"].update(normal_smooth_2d=0.1, normal_smooth_3d=0.0, normal_smoothness=0.0)
    sampler = _PresetSampler()
    sampler.samples = smp2
    fake = types.SimpleNamespace(model=m, occupancy_grid=sampler, config=cfg, dataset=types.SimpleNamespace(num_frames=200))
    res = ref_morpheus.MorpheuS.render_rays(fake, o, d, t, rid, hw2, hw2, bg_color=torch.tensor([0.2, 0.5, 0.7]), ambient_ratio=0.3,
                                            light_d=light2, shading="lambertian", real_view=False, cano=False)
    g["virt|image"], g["virt|normal_image"] = npf(res["image"]), npf(res["normal_image"])
    g["virt|loss_orient"], g["virt|loss_code"] = npf(res["loss_orient"]), npf(res["loss_code"])
    wimg = synth.hash_tensor((N2, 3), 360, 1.0)
    total = (res["normal_image"] * wimg).sum() + res["loss_orient"] + res["loss_code"] + (res["image"] ** 2).mean()
    m.zero_grad()
    total.backward()
    g["virt|loss"] = npf(total)
    for kk, v in grad_digest({k: p.grad for k, p in m.named_parameters() if p.grad is not None}).items():
        g["virt|grad|" + kk] = v
    # ---- two frames in one batch (B = 2 rows, one frame per row: how cfg5 / multi-frame batches reach render_rays)
    fr = [synth.frame_rays(fid, hw2, hw2) for fid in (0, 25)]
    o, d, t, rid = [torch.cat([f[k] for f in fr], 0) for k in range(4)]
    smpb = ofield.uniform_samples(o.reshape(-1, 3), d.reshape(-1, 3), synth.ray_jitter(2 * N2), 48, 1.01)
    lightb = ofield.safe_normalize(o.reshape(-1, 3) + torch.tensor([0.3, -0.2, 0.5]))
    m, cfg = build_ref_model(synth.make_state("b"), None)
    m.eval()
    sampler = _PresetSampler()
    sampler.samples = smpb
    fake = types.SimpleNamespace(model=m, occupancy_grid=sampler, config=cfg, dataset=types.SimpleNamespace(num_frames=200))
    res = ref_morpheus.MorpheuS.render_rays(fake, o, d, t, rid, hw2, hw2, ambient_ratio=1.0, light_d=lightb, shading="albedo")
    g["two|image"], g["two|depth"], g["two|deform_s16"] = npf(res["image"]), npf(res["depth"]), npf(res["deform"][::16])
    timg, tdep = synth.targets(2 * N2)
    lossb = ((res["image"].reshape(-1, 3) - timg) ** 2).mean() + ((res["depth"].reshape(-1) - tdep) ** 2).mean()
    m.zero_grad()
    lossb.backward()
    g["two|loss"] = npf(lossb)
    for kk, v in grad_digest({k: p.grad for k, p in m.named_parameters() if p.grad is not None}).items():
        g["two|grad|" + kk] = v
    np.savez_compressed(os.path.join(OUT, "extras.npz"), **g)
    print("extras.npz", len(g), "arrays")


# ----------------------------------------------------------------------------- round-4 fixtures (round4.npz)
def build_ref_model_f64(state, max_level=None):
    """The imported reference model in DOUBLE: same constructor, same weights (fp32 values, exactly representable), the hash
    grid evaluated by oracle/hashgrid_f64.py.  Forward only."""
    from oracle.hashgrid_f64 import OracleGridEncoderF64
    grid_mod = sys.modules["external.encoders.gridencoder.grid"]
    saved = grid_mod.GridEncoder
    grid_mod.GridEncoder = OracleGridEncoderF64
    try:
        m, cfg = build_ref_model(state, max_level)
    finally:
        grid_mod.GridEncoder = saved
    return m.double().eval(), cfg


def keep_mask_of_smoothness_points(res_depth, rays_o, rays_d, trunc, offsets_draw):
    """which of the npts x N surface-band points of get_normal_smoothness_loss (morpheus.py:530-556) the reference keeps
    (inside the 1.1 sphere), from its own rendered depth and its first draw -- stored so that the HIP-side test can hand the
    reference's [n_kept, 1] angle draw to the same points."""
    npts = int(trunc * 100 + 1)
    off = torch.linspace(-0.5 * trunc, 0.5 * trunc, npts) + 0.01 * offsets_draw
    pts = (res_depth.reshape(1, -1) + off[:, None])[..., None] * rays_d[None] + rays_o[None]
    return (torch.linalg.norm(pts.view(-1, 3), ord=2, dim=-1) < 1.1)


def gen_round4():
    import morpheus as ref_morpheus
    from bench_support import trainstep
    g = {}
    # ---- (1) float64 yardstick of the counted gate: the six evaluation renders of the parity table (2 weight states x
    #      {cfg1 deform, cfg1 canonical, cfg3-head deform}) by the imported reference in double
    for kind in ("a", "b"):
        st = synth.make_state(kind)
        for case, (hw, S, nray) in (("cfg1", (32, 64, None)), ("cfg3head", (128, 128, 256))):
            o, d, t, rid = synth.frame_rays(25, hw, hw)
            if nray is not None:
                o, d, t, rid = o[:, :nray], d[:, :nray], t[:, :nray], rid[:, :nray]
            N = o.shape[1]
            ri, ts, te = ofield.uniform_samples(o[0], d[0], synth.ray_jitter(N), S, 1.01)
            light = ofield.safe_normalize(o[0] + torch.tensor([0.3, -0.2, 0.5]))
            for mode in ("eval_albedo_deform", "eval_albedo_cano"):
                m, cfg = build_ref_model_f64(st, None)
                sampler = _PresetSampler()
                sampler.samples = (ri, ts.double(), te.double())
                fake = types.SimpleNamespace(model=m, occupancy_grid=sampler, config=cfg,
                                             dataset=types.SimpleNamespace(num_frames=200))
                with torch.no_grad():
                    res = ref_morpheus.MorpheuS.render_rays(fake, o.double(), d.double(), t.double(), rid, hw, hw, ambient_ratio=0.3,
                                                            light_d=light.double(), shading="albedo", cano="cano" in mode)
                key = f"{kind}_{case}_{mode}|f64"
                assert res["image"].dtype == torch.float64 and res["sdf"].dtype == torch.float64
                g[key + "|image"] = res["image"].numpy()
                g[key + "|depth"] = res["depth"].numpy()
                g[key + "|weights_sum"] = res["weights_sum"].numpy()
                g[key + "|sdf_s16"] = res["sdf"][::16].numpy()
    print("round4: f64 yardstick done", len(g))
    # ---- (2) the virtual-view step at 72 x 72 (datasets/dataset.py:503-578 at novel_view_scale 0.2 of 360): ALL rays of one
    #      novel view, shipped regularisers on (orientation loss, normal_smooth_3d, normal_smoothness, code_reg), random draws
    #      injected; the SDS guidance replaced by its interface (trainstep.InjectedGuidance: a fixed gradient on pred_rgb);
    #      the reference's own get_regularization_loss; backward.  Two of get_shading's outcomes (morpheus.py:864-885).
    hw, S = 72, 24
    for tag, (frame, theta, phi, shading, ambient, bg) in (
            ("lam", (140, 70.0, 35.0, "lambertian", 0.55, torch.tensor([0.2, 0.5, 0.7]))),
            ("tex", (31, 95.0, -120.0, "textureless", 0.3, None))):
        o, d = synth.camera_rays(hw, hw, synth.look_at_pose(theta, phi, 1.5))
        N = o.shape[0]
        o, d = o[None], d[None]
        t = torch.full((1, N, 1), frame / 200)
        rid = torch.full((1, N, 1), frame, dtype=torch.int64)
        samples = ofield.uniform_samples(o[0], d[0], synth.ray_jitter(N), S, 1.01)
        light = ofield.safe_normalize(o[0] + torch.tensor([0.3, -0.2, 0.5]))
        m, cfg = build_ref_model(synth.make_state("b"), 0.75)
        m.train()
        sampler = _PresetSampler()
        sampler.samples = samples
        fake = types.SimpleNamespace(model=m, occupancy_grid=sampler, config=cfg, dataset=types.SimpleNamespace(num_frames=200),
                                     global_step=1000)
        fake.get_ortho_normal_dir = types.MethodType(ref_morpheus.MorpheuS.get_ortho_normal_dir, fake)
        fake.get_normal_smoothness_loss = types.MethodType(ref_morpheus.MorpheuS.get_normal_smoothness_loss, fake)
        with DrawInjector() as inj:
            res = ref_morpheus.MorpheuS.render_rays(fake, o, d, t, rid, hw, hw, bg_color=bg, ambient_ratio=ambient, light_d=light,
                                                    shading=shading, real_view=False, cano=False)
            n_draws = inj.k
        key = "virt72_" + tag
        g[key + "|n_draws"] = np.int32(n_draws)
        assert n_draws == 3, n_draws          # randn_like(xyzs); rand_like(trunc offsets); rand([n_kept, 1])
        draw2 = synth.hash_tensor((int(cfg["train"]["trunc"] * 100 + 1),), 8000 + 2, 0.5) + 0.5
        keep = keep_mask_of_smoothness_points(res["depth"].detach(), o[0], d[0], cfg["train"]["trunc"], draw2)
        g[key + "|keep_bits"] = np.packbits(keep.numpy())
        g[key + "|n_keep"] = np.int32(int(keep.sum()))
        for lk in ("loss_orient", "loss_normal_perturb", "normal_reg", "loss_code"):
            g[key + "|" + lk] = npf(res[lk])
        g[key + "|image"], g[key + "|depth"] = npf(res["image"]), npf(res["depth"])
        g[key + "|weights_sum"], g[key + "|sdf_s16"] = npf(res["weights_sum"]), npf(res["sdf"][::16])
        g[key + "|normal_s16"] = npf(res["normal"][::16])
        pred_rgb, pred_depth, pred_mask, pred_normal, pred_sdf = ref_morpheus.MorpheuS.get_pred_from_outputs(fake, res, 1, hw, hw)
        l_guid = trainstep.InjectedGuidance(hw, hw, "cpu", scale=5e-3)(pred_rgb)
        l_reg = ref_morpheus.MorpheuS.get_regularization_loss(fake, res, pred_normal, cano=False)
        total = l_guid + l_reg
        m.zero_grad()
        total.backward()
        g[key + "|loss_guidance"], g[key + "|loss_reg"], g[key + "|loss"] = npf(l_guid), npf(l_reg), npf(total)
        for kk, v in grad_digest({k: p.grad for k, p in m.named_parameters() if p.grad is not None}).items():
            g[key + "|grad|" + kk] = v
        print("round4:", key, "kept", int(keep.sum()), "of", keep.numel(), "loss", float(total))
    np.savez_compressed(os.path.join(OUT, "round4.npz"), **g)
    print("round4.npz", len(g), "arrays")


# ----------------------------------------------------------------------------- round-5 fixture (round5.npz)
def _virtual_step_in_double(g, hw, S, tag):
    """One virtual-view TRAINING step of the imported reference, run in fp32 AND in float64 on the same inputs with its backward
    (hw x hw rays x S samples; lambertian through finite-difference normals, orientation loss, normal_smooth_3d with its draws
    injected, code_reg; normal_smoothness off: its angle draw depends on a boolean index), so that the HIP path's gradients can be held
    to "within k x the reference's own fp32 error" instead of a fitted 1-3e-2.  Writes `tag|f32|...` and `tag|f64|...` into g."""
    import morpheus as ref_morpheus
    from bench_support import trainstep
    o, d = synth.camera_rays(hw, hw, synth.look_at_pose(70.0, 35.0, 1.5))
    N = o.shape[0]
    o, d = o[None], d[None]
    t = torch.full((1, N, 1), 140 / 200)
    rid = torch.full((1, N, 1), 140, dtype=torch.int64)
    smp = ofield.uniform_samples(o[0], d[0], synth.ray_jitter(N), S, 1.01)
    light = ofield.safe_normalize(o[0] + torch.tensor([0.3, -0.2, 0.5]))
    bg = torch.tensor([0.2, 0.5, 0.7])
    for prec in ("f32", "f64"):
        if prec == "f32":
            m, cfg = build_ref_model(synth.make_state("b"), 0.75)
        else:
            m, cfg = build_ref_model_f64(synth.make_state("b"), 0.75)
            m.encoder.differentiable = m.encoder_c.differentiable = True
        m.train()
        cfg["train"]["normal_smoothness"] = 0.0
        cast = (lambda v: v.double()) if prec == "f64" else (lambda v: v)
        sampler = _PresetSampler()
        sampler.samples = (smp[0], cast(smp[1]), cast(smp[2]))
        fake = types.SimpleNamespace(model=m, occupancy_grid=sampler, config=cfg, dataset=types.SimpleNamespace(num_frames=200),
                                     global_step=1000)
        fake.get_ortho_normal_dir = types.MethodType(ref_morpheus.MorpheuS.get_ortho_normal_dir, fake)
        with DrawInjector() as inj:
            res = ref_morpheus.MorpheuS.render_rays(fake, cast(o), cast(d), cast(t), rid, hw, hw, bg_color=cast(bg), ambient_ratio=0.55,
                                                    light_d=cast(light), shading="lambertian", real_view=False, cano=False)
            n_draws = inj.k
        pred_rgb, _, _, pred_normal, _ = ref_morpheus.MorpheuS.get_pred_from_outputs(fake, res, 1, hw, hw)
        G = trainstep.InjectedGuidance(hw, hw, "cpu", scale=5e-3)
        l_guid = (pred_rgb * cast(G.grad)).sum()
        l_reg = ref_morpheus.MorpheuS.get_regularization_loss(fake, res, pred_normal, cano=False)
        total = l_guid + l_reg
        m.zero_grad()
        total.backward()
        key = tag + "|" + prec
        g[key + "|n_draws"] = np.int32(n_draws)
        for lk in ("loss_orient", "loss_normal_perturb", "loss_code"):
            g[key + "|" + lk] = np.float64(float(res[lk]))
        g[key + "|loss"] = np.float64(float(total))
        g[key + "|image"] = res["image"].detach().double().numpy()
        for kk, v in grad_digest({k: p.grad for k, p in m.named_parameters() if p.grad is not None}).items():
            g[key + "|grad|" + kk] = v if not isinstance(v, np.ndarray) else v.astype(np.float64) if prec == "f64" else v
        print("virtual step in double:", tag, prec, "loss", float(total), "draws", n_draws)


def gen_round5():
    """cfg4's STEP COMPOSITION (morpheus.py:1390-1424): one virtual-view backward and one real-view backward feeding torch.optim.Adam
    over model.get_params_all(lr) (:154-155), learning rates set by the reference's own update_learning_rate (:472-503) and, in the
    `freeze` variant, freeze_lr_deform / reset_lr_deform (:504-516):
        accum   (epoch > freeze_epoch)  zero_grad; (1/virtual_freq * virtual loss).backward(); real loss.backward(); step()
        freeze  (epoch <= freeze_epoch) freeze_lr_deform; virtual backward; step(); zero_grad; reset_lr_deform; real backward; step()
    The two steps are the fixtures already pinned one by one (round4.npz virt72_lam, extras.npz realview), here on ONE model, with
    the real-view background colour drawn as train_step draws it (torch.rand before render_rays, :893-894).  Adam's state is seeded
    (step 1000, exp_avg 0, exp_avg_sq 1: a move is -0.1 lr g / sqrt(0.99 + 0.01 g^2), proportional to the gradient up to |g| ~ 10) -- the first step of a fresh
    Adam with eps = 1e-15 moves every touched element by exactly +-lr, which would pin signs only.  Stored: the group learning rates,
    the two losses, and per-tensor digests of the parameter DELTAS."""
    import morpheus as ref_morpheus
    from bench_support import trainstep
    g = {}
    hw_v, S_v = 72, 24
    frame_v, theta, phi, shading_v, ambient_v, bg_v = 140, 70.0, 35.0, "lambertian", 0.55, torch.tensor([0.2, 0.5, 0.7])
    o_v, d_v = synth.camera_rays(hw_v, hw_v, synth.look_at_pose(theta, phi, 1.5))
    N_v = o_v.shape[0]
    o_v, d_v = o_v[None], d_v[None]
    t_v = torch.full((1, N_v, 1), frame_v / 200)
    rid_v = torch.full((1, N_v, 1), frame_v, dtype=torch.int64)
    smp_v = ofield.uniform_samples(o_v[0], d_v[0], synth.ray_jitter(N_v), S_v, 1.01)
    light_v = ofield.safe_normalize(o_v[0] + torch.tensor([0.3, -0.2, 0.5]))
    hw_r, S_r = 32, 64
    sel = real_view_case("b", hw_r, S_r)
    o_r, d_r, t_r, rid_r = [v[:, sel] for v in synth.frame_rays(25, hw_r, hw_r)]
    N_r = o_r.shape[1]
    smp_r = ofield.uniform_samples(o_r[0], d_r[0], synth.ray_jitter(hw_r * hw_r)[sel], S_r, 1.01)
    frame = trainstep.make_frames([25], hw_r, hw_r, "cpu")[0]
    data_r = trainstep.sample_real_view_rays(frame, N_r, sel)

    def fake_of(m, cfg, samples):
        sampler = _PresetSampler()
        sampler.samples = samples
        fake = types.SimpleNamespace(model=m, occupancy_grid=sampler, config=cfg, dataset=types.SimpleNamespace(num_frames=200),
                                     global_step=1000, device="cpu")
        fake.get_ortho_normal_dir = types.MethodType(ref_morpheus.MorpheuS.get_ortho_normal_dir, fake)
        fake.get_normal_smoothness_loss = types.MethodType(ref_morpheus.MorpheuS.get_normal_smoothness_loss, fake)
        return fake

    def virtual_loss(m, cfg):
        fake = fake_of(m, cfg, smp_v)
        with DrawInjector():
            res = ref_morpheus.MorpheuS.render_rays(fake, o_v, d_v, t_v, rid_v, hw_v, hw_v, bg_color=bg_v, ambient_ratio=ambient_v,
                                                    light_d=light_v, shading=shading_v, real_view=False, cano=False)
        pred_rgb, _, _, pred_normal, _ = ref_morpheus.MorpheuS.get_pred_from_outputs(fake, res, 1, hw_v, hw_v)
        return (trainstep.InjectedGuidance(hw_v, hw_v, "cpu", scale=5e-3)(pred_rgb) +
                ref_morpheus.MorpheuS.get_regularization_loss(fake, res, pred_normal, cano=False)), res

    def real_loss(m, cfg):
        fake = fake_of(m, cfg, smp_r)
        rec = {}
        with DrawInjector() as inj:
            inner = fake.get_normal_smoothness_loss

            def recording(rays_o, rays_d, rays_t, depth):
                # which of the npts x N surface-band points the reference keeps (inside the 1.1 sphere, morpheus.py:543-549): it
                # draws the perturbation angles on the KEPT points only; the HIP-side test hands them to the same points
                off_draw = synth.hash_tensor((int(cfg["train"]["trunc"] * 100 + 1),), inj.base + inj.k + 1, 0.5) + 0.5
                rec["keep"] = keep_mask_of_smoothness_points(depth.detach(), rays_o.detach(), rays_d.detach(), cfg["train"]["trunc"], off_draw)
                rec["angle_draw"] = inj.k + 2
                return inner(rays_o, rays_d, rays_t, depth)

            fake.get_normal_smoothness_loss = recording
            bg = torch.rand((N_r, 3))                                     # get_bg_color, real view (morpheus.py:893-894): draw 1
            res = ref_morpheus.MorpheuS.render_rays(fake, o_r, d_r, t_r, rid_r, N_r, 1, bg_color=bg, ambient_ratio=1.0,
                                                    shading="albedo_normal", real_view=True, cano=False,
                                                    rays_depth=data_r["depth"].view(1, -1, 1), rays_mask=data_r["mask"].view(1, -1, 1),
                                                    optimize_pose=True)
            n_draws = inj.k
        B, H, W = 1, N_r, 1
        pred_rgb, pred_depth, pred_mask, pred_normal, _ = ref_morpheus.MorpheuS.get_pred_from_outputs(fake, res, B, H, W)
        gt_rgb, gt_depth, gt_mask = ref_morpheus.MorpheuS.get_gt_from_data(
            fake, {k: (v.clone() if torch.is_tensor(v) else v) for k, v in data_r.items()}, bg, B, H, W)
        loss = ref_morpheus.MorpheuS.get_real_view_render_loss(fake, pred_rgb, pred_depth, pred_mask, gt_rgb, gt_depth, gt_mask,
                                                               data_r["rays_o"], data_r["rays_d"])
        loss = loss + ref_morpheus.MorpheuS.get_real_view_point_loss(fake, gt_rgb, gt_depth, gt_mask, data_r["rays_o"], data_r["rays_d"],
                                                                     data_r["rays_t"], res)
        loss = loss + ref_morpheus.MorpheuS.get_regularization_loss(fake, res, pred_normal, cano=False)
        return loss, n_draws, rec

    g["real|sel"] = sel.numpy().astype(np.int32)
    for variant in ("accum", "freeze"):
        m, cfg = build_ref_model(synth.make_state("b"), 0.75)
        m.train()
        if variant == "freeze":
            # The normal-smoothness term (get_normal_smoothness_loss: differences of NORMALISED finite-difference gradients) is
            # ill-conditioned on these closed-form weights: measured on the reference itself, its gradient to the SDF net's last bias
            # grows 0.70 -> 10.9 for the <= 1e-3 relative parameter move of the first step.  It is pinned at the initial parameters
            # (extras.npz realview, round4.npz virt72, the `accum` variant here); a second step evaluated AFTER a move would compare
            # round-off, not the composition -- so this variant runs both steps without it (one draw pair less per render).
            cfg["train"]["normal_smoothness"] = 0.0
        opt = torch.optim.Adam(m.get_params_all(cfg["train"]["lr"]), betas=(0.9, 0.99), eps=1e-15)
        host = types.SimpleNamespace(epoch=1000, config=cfg, optimizer=opt)
        ref_morpheus.MorpheuS.update_learning_rate(host)                   # the cosine factor at epoch 1000 of n_epochs; pose x 0.1
        g[variant + "|group_names"] = np.array([gr["name"] for gr in opt.param_groups])
        g[variant + "|group_lr"] = np.array([gr["lr"] for gr in opt.param_groups], dtype=np.float64)
        for gr in opt.param_groups:
            for p in gr["params"]:
                opt.state[p] = dict(step=torch.tensor(1000.0), exp_avg=torch.zeros_like(p), exp_avg_sq=torch.full_like(p, 1.0))
        before = {k: p.detach().clone() for k, p in m.named_parameters()}
        inv_vf = 1.0 / cfg["train"]["virtual_freq"]
        opt.zero_grad()
        if variant == "accum":
            lv, _ = virtual_loss(m, cfg)
            (inv_vf * lv).backward()
            lr_, n_draws, rec = real_loss(m, cfg)
            lr_.backward()
            opt.step()
        else:
            ref_morpheus.MorpheuS.freeze_lr_deform(host)
            g[variant + "|group_lr_frozen"] = np.array([gr["lr"] for gr in opt.param_groups], dtype=np.float64)
            lv, _ = virtual_loss(m, cfg)
            (inv_vf * lv).backward()
            opt.step()
            for kk, v in grad_digest({k: p.detach() - before[k] for k, p in m.named_parameters()}).items():
                g[variant + "|delta1|" + kk] = v           # after the virtual-view step alone (frozen groups: exactly 0)
            opt.zero_grad()
            m.zero_grad()
            ref_morpheus.MorpheuS.reset_lr_deform(host)
            lr_, n_draws, rec = real_loss(m, cfg)
            lr_.backward()
            for kk, v in grad_digest({k: p.grad for k, p in m.named_parameters() if p.grad is not None}).items():
                g[variant + "|grad2|" + kk] = v            # the real-view gradients at the once-stepped parameters
            opt.step()
        g[variant + "|loss_virtual"], g[variant + "|loss_real"] = npf(lv), npf(lr_)
        g[variant + "|real_n_draws"] = np.int32(n_draws)
        if rec:
            g[variant + "|real_keep_bits"] = np.packbits(rec["keep"].numpy())
            g[variant + "|real_n_keep"] = np.int32(int(rec["keep"].sum()))
            g[variant + "|real_angle_draw"] = np.int32(rec["angle_draw"])
            print("round5:", variant, "real-view smoothness points kept", int(rec["keep"].sum()), "of", rec["keep"].numel(), "angle draw", rec["angle_draw"])
        delta = {k: p.detach() - before[k] for k, p in m.named_parameters()}
        for kk, v in grad_digest(delta).items():
            g[variant + "|delta|" + kk] = v
        moved = sum(int((v != 0).any()) for v in delta.values())
        print("round5:", variant, "virtual", float(lv), "real", float(lr_), "tensors moved", moved, "of", len(delta))
    _virtual_step_in_double(g, 24, 24, "virt24")
    np.savez_compressed(os.path.join(OUT, "round5.npz"), **g)
    print("round5.npz", len(g), "arrays")


def gen_round6():
    """Round 6: the double-precision yardstick of the virtual-view step's gradients at the size of the 72 x 72 fixture test (round 5
    had it at 24 x 24 only; VERDICT r5 "nothing of that kind exists at 72^2"): 5 184 rays x 32 samples, fp32 and float64."""
    g = {}
    _virtual_step_in_double(g, 72, 32, "virt72d")
    np.savez_compressed(os.path.join(OUT, "round6.npz"), **g)
    print("round6.npz", len(g), "arrays")


def main():
    assert os.path.isdir(REF), "make_golden.py needs /root/reference (build container only)"
    os.makedirs(OUT, exist_ok=True)
    install_shims()
    torch.set_num_threads(8)
    if "--variants-only" in sys.argv:      # round 3: only the model-switch fixtures (the others are unchanged)
        gen_variants()
        return
    if "--round4-only" in sys.argv:        # round 4: float64 yardstick + the 72 x 72 virtual-view step (the others are unchanged)
        gen_round4()
        return
    if "--round5-only" in sys.argv:        # round 5: cfg4's step composition (two backwards -> Adam), the others are unchanged
        gen_round5()
        return
    if "--round6-only" in sys.argv:        # round 6: the 72 x 72 virtual-view step in double (the others are unchanged)
        gen_round6()
        return
    if "--extras-only" not in sys.argv:
        gen_operators()
        gen_model()
        gen_render()
    gen_extras()
    gen_variants()
    gen_round4()
    gen_round5()
    gen_round6()


if __name__ == "__main__":
    main()
