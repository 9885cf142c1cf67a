"""`render_rays` with the reference's signature and result keys (morpheus.py:558-794).

`HotPathRenderer` plays the role of the `self` that MorpheuS.render_rays closes over: it owns
`model` (scene_representation), `config` (the YAML dict), `occupancy_grid` (anything with a
nerfacc-style `.sampling(rays_o, rays_d, ...) -> (ray_indices, t_starts, t_ends)`) and
`num_frames`.  A maintainer wires it into morpheus.py by delegating MorpheuS.render_rays to
`HotPathRenderer.render_rays`.

Covered: the eval outputs (image, depth, sdf, weights, weights_sum, normal, deform, normal_raw) and
the deterministic training extras (loss_orient, loss_code, sdf_loss, fs_loss, normal_image).  The
randomised regularisers of morpheus.py:714-783 (normal perturbation / smoothness) reuse the same
model entry points and are implemented here on top of them.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import chunking, ops
from .model import safe_normalize, scene_representation


class UniformSampler:
    """Benchmark sampler of SURVEY 8(d) behind nerfacc's `.sampling` call shape: S equal bins per
    ray inside the AABB with one stratified jitter per ray (csrc/sampler.hip)."""

    def __init__(self, n_samples: int, bound: float, jitter: Optional[torch.Tensor] = None):
        self.n_samples, self.bound, self.jitter = n_samples, float(bound), jitter
        self.packed = None
        self.xyz = None      # sample positions o + d (ts+te)/2 of the last call (same arithmetic as morpheus.py:647)

    def sampling(self, rays_o, rays_d, sigma_fn=None, render_step_size=None, alpha_thre=0, stratified=True,
                 cone_angle=0.0, early_stop_eps=0):
        n = rays_o.shape[0]
        if self.jitter is not None:
            u = self.jitter
        elif stratified:
            u = torch.rand(n, device=rays_o.device)
        else:
            u = torch.full((n,), 0.5, device=rays_o.device)
        ri, ts, te, xyz, rs, rc = ops.sample_uniform(rays_o, rays_d, u, self.n_samples, self.bound, with_xyz=True)
        self.packed, self.xyz = (rs, rc), xyz
        return ri, ts, te


class PresetSampler:
    """Feeds fixed packed samples (parity runs treat samples as inputs)."""

    def __init__(self, ray_indices, t_starts, t_ends):
        self.samples = (ray_indices, t_starts, t_ends)
        self.packed = None

    def sampling(self, rays_o, rays_d, **kw):
        return self.samples


class HotPathRenderer:
    def __init__(self, model: scene_representation, config: dict, occupancy_grid, num_frames: int,
                 frame_batched: bool = True):
        self.model, self.config, self.occupancy_grid, self.num_frames = model, config, occupancy_grid, num_frames
        # frame_batched: each row of the [B, N, .] ray tensors is ONE frame (how the reference's dataset
        # builds every batch) -> per-frame deform-code bias without a host sync.
        self.frame_batched = frame_batched

    # -- helpers of morpheus.py:518-556
    def _const(self, key, build, device):
        """small device constants built once per device (a torch.tensor(list, device=...) is a host->device copy: a sync
        point of the eager step and illegal inside a captured HIP graph)"""
        c = self.__dict__.setdefault("_consts", {})
        k = (key, str(device))
        if k not in c:
            c[k] = build().to(device)
        return c[k]

    def get_ortho_normal_dir(self, normals, phi=None):
        """morpheus.py:518-528; `phi` injects the random angle (parity tests)."""
        n = torch.nn.functional.normalize(normals, dim=-1)
        # n[..., [1, 0, 2]] * (1, -1, 0) without host-built index / constant tensors (each is a host->device copy: a sync in the
        # eager step, illegal inside a captured HIP graph); the same three values
        u = torch.nn.functional.normalize(torch.stack([n[..., 1], -n[..., 0], n[..., 2] * 0.0], -1), dim=-1)
        v = torch.cross(n, u, dim=-1)
        if phi is None:
            phi = self._ortho_angle(normals)
        return torch.cos(phi) * u + torch.sin(phi) * v

    @staticmethod
    def _ortho_angle(normals):
        """the random angle of get_ortho_normal_dir (morpheus.py:525): one uniform draw per normal, times 2 pi"""
        # (fp32(2 pi) == 2 fp32(pi): one product gives the bits of the reference's `rand() * 2.0 * np.pi`)
        return torch.rand(list(normals.shape[:-1]) + [1], device=normals.device) * (2.0 * np.pi)

    def get_normal_smoothness_loss(self, rays_o, rays_d, rays_t, depth, offsets=None, phi=None, ray_slots=None,
                                   single_frame=None):
        """morpheus.py:530-556: normals at npts = trunc*100+1 points around the rendered depth of every ray vs normals at
        points displaced by smoothness_std along a random direction orthogonal to the normal.  The reference drops the
        points outside the 1.1 sphere with a boolean index (a device->host sync and a data-dependent shape); here they
        stay in the batch and leave the mean through a 0/1 weight -- the same value, no sync.
        `offsets` [npts] / `phi` [npts*N, 1] inject the two random draws (parity tests); `ray_slots` = (t_rows, slot per
        ray) for multi-frame batches; `single_frame`: every ray carries rays_t[0] (one batch row = one frame).  With neither,
        the points take their own ray's time, as the reference does (per-ray rays_t)."""
        trunc = self.config["train"]["trunc"]
        npts = int(trunc * 100 + 1)
        if offsets is None:
            off = self._const(("smooth_offsets", trunc, npts), lambda: torch.linspace(-0.5 * trunc, 0.5 * trunc, npts), depth.device)
            off = torch.add(off, torch.rand_like(off), alpha=0.01)
        else:
            off = offsets
        # pts = (depth + off[:, None])[..., None] * rays_d[None] + rays_o[None] and the 1.1-sphere test in one launch each way
        pts, keep = ops.smooth_points(depth, off.to(depth), rays_o, rays_d)
        n_rays = rays_t.shape[0]
        if ray_slots is not None:
            tt, fs = rays_t[None].repeat(npts, 1, 1).view(-1, 1), (ray_slots[0], ray_slots[1].repeat(npts))
        elif single_frame:         # one frame: the time is an expanded scalar (model._slots sees a single slot)
            tt, fs = rays_t[:1].expand(npts * n_rays, 1), None
        else:                      # rays of several times without a row structure: per-sample times (model._slots)
            tt, fs = rays_t[None].repeat(npts, 1, 1).view(-1, 1), None
        n1, _ = self.model.normal(pts, t=tt, frame_slots=fs)
        # pts + get_ortho_normal_dir(n1) * smoothness_std, then sum(square(n1 - n2) * keep) / max(3 sum(keep), 1): one launch each
        pts_p = ops.ortho_perturb(pts, n1, self._ortho_angle(n1) if phi is None else phi, self.config["train"]["smoothness_std"])
        n2, _ = self.model.normal(pts_p, t=tt, frame_slots=fs)
        return ops.masked_mean("sqdiff", n1, n2, row_weight=keep)

    # -- forward-only consumer (f-4)
    def eval_step(self, data, cano=False, optimize_pose=False, max_chunk=300 * 300):
        """MorpheuS.eval_step (morpheus.py:1238-1269; the same loop is in visualizer.py:69-91): a whole view rendered in
        chunks of <= max_chunk rays, forward only (callers wrap it in no_grad), RGB and depth images re-assembled."""
        rays_o, rays_d, rays_t, rays_id = data["rays_o"], data["rays_d"], data["rays_t"], data["rays_id"]
        B, N = rays_o.shape[:2]
        H, W = data["H"], data["W"]
        shading = data["shading"] if "shading" in data else "albedo"
        ambient_ratio = data["ambient_ratio"] if "ambient_ratio" in data else 1.0
        scale_factor = int(N // max_chunk + 1)
        if N > max_chunk:
            max_chunk = N // scale_factor + 1
        pred_rgb, pred_depth = [], []
        for i in range(0, N, max_chunk):
            out = self.render_rays(rays_o[:, i:i + max_chunk], rays_d[:, i:i + max_chunk], rays_t[:, i:i + max_chunk],
                                   rays_id[:, i:i + max_chunk], H, W, perturb=True, ambient_ratio=ambient_ratio,
                                   shading=shading, cano=cano, optimize_pose=optimize_pose)
            pred_rgb.append(out["image"])
            pred_depth.append(out["depth"])
        return torch.cat(pred_rgb, dim=1).reshape(B, H, W, 3), torch.cat(pred_depth, dim=1).reshape(B, H, W)

    # -- the hot path
    def render_rays(self, rays_o, rays_d, rays_t, rays_id, H, W, perturb=True, bg_color=None, ambient_ratio=1.0,
                    light_d=None, shading="albedo", real_view=True, cano=False, rays_depth=None, rays_mask=None,
                    optimize_pose=False):
        with self.model.operand_scope():      # weight operands prepared once for every field query of this call
            return self._render_rays(rays_o, rays_d, rays_t, rays_id, H, W, perturb, bg_color, ambient_ratio, light_d,
                                     shading, real_view, cano, rays_depth, rays_mask, optimize_pose)

    def _render_rays(self, rays_o, rays_d, rays_t, rays_id, H, W, perturb, bg_color, ambient_ratio, light_d, shading,
                     real_view, cano, rays_depth, rays_mask, optimize_pose):
        model, cfg = self.model, self.config
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        rays_t = rays_t.contiguous().view(-1, 1)
        rays_id = rays_id.contiguous().view(-1, 1)
        if not cano and optimize_pose:
            rows = tuple(prefix) if (self.frame_batched and len(prefix) == 2) else None
            rays_o, rays_d = model.pose_optimisation(rays_o, rays_d, rays_id, rows=rows)
        if rays_depth is not None:
            rays_depth = rays_depth.contiguous().view(-1, 1)
        if rays_mask is not None:
            rays_mask = rays_mask.contiguous().view(-1, 1)
        N = rays_o.shape[0]
        results = {}

        with torch.no_grad():
            ray_indices, t_starts_, t_ends_ = self.occupancy_grid.sampling(
                rays_o, rays_d, sigma_fn=None, render_step_size=cfg["render"]["step_size"], alpha_thre=0,
                stratified=True, cone_angle=0.0, early_stop_eps=0)
        # per-sample light directions are only read by the shaded modes (model.py:515-531), and not at ambient_ratio = 1
        # (real-view steps), where the lambertian factor is exactly 1
        lit = shading != "albedo" and (ambient_ratio != 1 or shading in ("textureless", "normal"))
        if light_d is None and lit:
            light_d = safe_normalize(rays_o + torch.randn(3, device=rays_o.device))      # morpheus.py:641
        M_samples = ray_indices.shape[0]
        # fixed-capacity sampling (occgrid.OccupancyGrid.sample_capacity): only the first n_valid packed entries are samples
        n_valid = getattr(self.occupancy_grid, "n_valid", None)
        valid = None if n_valid is None else (torch.arange(M_samples, device=rays_o.device) < n_valid)

        def l1_mean(a, b):
            """mean |a - b| over the SAMPLES of two per-sample tensors [M, ...] (the reference's `(a - b).abs().mean()`), padding
            entries excluded: one launch each way"""
            return ops.masked_mean("absdiff", a, b, n_valid=n_valid)

        single_frame = (not cano) and self.frame_batched and len(prefix) == 2 and prefix[0] == 1
        ray_idx32 = ray_indices
        _long = []

        def ri_long():          # int64 indices for torch gathers, built only if something asks for them
            if not _long:
                _long.append(ray_idx32.long())
            return _long[0]

        _i32 = []

        def ri32():             # int32 indices for the HIP kernels (the marcher / uniform sampler already produce int32)
            if not _i32:
                _i32.append(ray_idx32 if ray_idx32.dtype == torch.int32 else ray_idx32.to(torch.int32))
            return _i32[0]

        xyzs = getattr(self.occupancy_grid, "xyz", None)
        if xyzs is not None:
            self.occupancy_grid.xyz = None             # one use: it belongs to the sampling call above
        packed = getattr(self.occupancy_grid, "packed", None)    # (ray_start, ray_cnt) of the sampling call above
        ray_start, ray_cnt = packed if packed is not None else ops.packed_info(ri_long(), N)
        if xyzs is None or rays_o.requires_grad or rays_d.requires_grad:   # pose optimisation: positions carry gradients
            # xyz = o[ri] + d[ri] * (ts + te) / 2 (morpheus.py:644-647) as one launch; its backward is a per-ray segment sum
            xyzs = ops.sample_positions(rays_o, rays_d, ri32(), t_starts_, t_ends_, ray_start, ray_cnt)
        # a batch row is one frame: with a single row every sample shares rays_t[0] -- no gather needed
        time_step = rays_t[:1].expand(M_samples, 1) if single_frame else rays_t[ri_long()]

        if M_samples == 0:
            # the reference falls into a NameError here; return the white image it intended
            results.update(image=torch.ones([*prefix, 3], device=rays_o.device),
                           depth=torch.zeros([*prefix], device=rays_o.device), sdf=None, weights=None,
                           weights_sum=None, normal=None, deform=None, normal_raw=None)
            return results

        # per-frame deform-code slots without a device->host sync.  One frame: `time_step` is an expanded scalar, which
        # the model recognises as a single slot by itself.  Several frames (a batch row is one frame): slot per ray row.
        frame_slots = ray_slots = None
        if not single_frame and not cano and self.frame_batched and len(prefix) == 2:
            B, n_per = prefix
            slot_ray = torch.arange(B, device=rays_o.device, dtype=torch.int32).repeat_interleave(n_per)
            ray_slots = (rays_t.view(B, n_per)[:, 0].contiguous(), slot_ray)
            frame_slots = (ray_slots[0], slot_ray[ri_long()].contiguous())
        t_light = light_d[ri_long()] if lit else None
        # a bound on what the call parks (chunking.py): the main query and the perturbed-normal query of the regulariser in row
        # chunks when their estimate passes MORPHEUS_MAX_PARK_GB -- chunks but the last re-made in backward
        tr_ = cfg["train"]
        with_normals = shading != "albedo"
        perturb_query = model.training and tr_["normal_smooth_3d"] > 0 and with_normals
        row_bytes = chunking.query_bytes_per_row(not cano, 1 + (6 if with_normals else 0)) + \
            (chunking.query_bytes_per_row((not tr_["topo_none"]) and not cano, 6) if perturb_query else 0)
        chunk_rows = chunking.rows_under_cap(row_bytes, device=rays_o.device, rows=M_samples) if torch.is_grad_enabled() else M_samples
        slot_rows = None if frame_slots is None else frame_slots[1]

        def main_query(x_, t_, l_, s_):
            return model(x_, t_, l_, ratio=ambient_ratio, shading=shading, cano=cano,
                         frame_slots=None if s_ is None else (frame_slots[0], s_))

        sdf, sigmas, rgbs, normals, deform, normal_raw = chunking.chunked_query(main_query, (xyzs, time_step, t_light, slot_rows),
                                                                                 chunk_rows, model=model)

        weights, opacity, depth, rgb_acc = ops.composite(sigmas, t_starts_.contiguous(), t_ends_.contiguous(), rgbs,
                                                         ray_start, ray_cnt, padded=valid is not None)
        opacity, depth = opacity[:, None], depth[:, None]
        if bg_color is None:
            if cfg["model"]["bg_radius"] > 0 and cano and (not real_view):
                bg_color = model.background(rays_d, rays_t)
            else:
                bg_color = 1
        if isinstance(bg_color, (int, float)) and rgb_acc.is_cuda:
            # a constant background (the reference's white, `bg_color = 1`): one cached [N, 3] constant, the same blend launch
            n_rays_ = rgb_acc.shape[0]
            bg_color = self._const(("bg", float(bg_color), n_rays_), lambda: torch.full((n_rays_, 3), float(bg_color)), rgb_acc.device)
        elif (torch.is_tensor(bg_color) and bg_color.is_cuda and not bg_color.requires_grad and bg_color.numel() == 3
              and bg_color.shape != rgb_acc.shape):
            bg_color = bg_color.reshape(1, 3).to(rgb_acc.dtype).expand(rgb_acc.shape[0], 3).contiguous()     # one colour for every ray
        if torch.is_tensor(bg_color) and bg_color.shape == rgb_acc.shape and bg_color.is_cuda:
            image = ops.bg_blend(rgb_acc, opacity, bg_color).view(*prefix, 3)       # the same three rounded operations, one launch
        else:
            image = (rgb_acc + (1 - opacity) * bg_color).view(*prefix, 3)
        depth = depth.view(*prefix)
        results.update(image=image, depth=depth, sdf=sdf, weights=weights, weights_sum=opacity, normal=normals,
                       deform=deform, normal_raw=normal_raw)
        if valid is not None:
            results.update(valid=valid, n_valid=n_valid)      # for the caller's own per-sample means (trainstep.sample_mean)

        if model.training:
            tr = cfg["train"]
            if tr["ori_weight"] > 0 and normals is not None and (not real_view):
                t_dirs = safe_normalize(rays_d[ri_long()])
                lo = weights.detach() * (normals * t_dirs).sum(-1).clamp(min=0) ** 2
                results["loss_orient"] = lo.sum(-1).mean()
            if tr["normal_smooth_3d"] > 0 and normals is not None:
                if tr["normal_dir"]:
                    xyzs_p = ops.ortho_perturb(xyzs, normals, self._ortho_angle(normals), tr["smoothness_std"])
                else:
                    xyzs_p = torch.add(xyzs, torch.randn_like(xyzs), alpha=tr["smoothness_std"])
                def perturbed_normals(x_, t_, s_):
                    if tr["topo_none"]:
                        return model.normal(x_, topo=None, cano=cano)[:1]
                    fs_ = None if s_ is None else (frame_slots[0], s_)
                    return model.normal(x_, topo=model.get_topo(x_, t=t_, frame_slots=fs_), cano=cano)[:1]

                normals_p, = chunking.chunked_query(perturbed_normals, (xyzs_p, time_step, slot_rows), chunk_rows, model=model)
                results["loss_normal_perturb"] = l1_mean(normals, normals_p)
                if tr["normal_smooth_3d_t"] > 0:
                    tt = time_step + torch.rand_like(time_step) * 1 / self.num_frames
                    normals_pt, _ = model.normal(xyzs, topo=model.get_topo(xyzs, t=tt), cano=cano)
                    results["loss_normal_perturb_t"] = l1_mean(normals, normals_pt)
                if tr["deform_smooth"] > 0 and not cano:
                    deform_p, _, _ = model.warp(xyzs_p, t=time_step, frame_slots=frame_slots)
                    results["loss_deform_perturb"] = l1_mean(deform, deform_p)
            if (tr["deform_smooth_t"] > 0 or tr["topo_smooth_t"] > 0) and not cano:
                tt = time_step + torch.rand_like(time_step) * 1 / self.num_frames
                deform_pt, topo_pt, _ = model.warp(xyzs, t=tt)
                topo_now = model.get_topo(xyzs, t=time_step, frame_slots=frame_slots)   # the reference reads an undefined `topo` here
                results["loss_deform_perturb_t"] = l1_mean(deform, deform_pt)
                results["loss_topo_perturb_t"] = l1_mean(topo_now, topo_pt)
            if tr["code_reg"] > 0 and not cano:
                # the three code samples (t0, t0 - 1/F, t0 + 1/F: morpheus.py:783-787) from ONE launch each way: the offsets are a
                # cached device constant (t0 + (-d) is t0 - d bit for bit), the rows come apart with unbind (one stack backward)
                d = 1.0 / self.num_frames
                offs = self._const(("code_reg_offsets", d), lambda: torch.tensor([[0.0], [-d], [d]]), time_step.device)
                code, cp, cn = model.get_deform_code(time_step[:1] + offs).unbind(0)
                # square(2 code - cp - cn).mean() with the chain's rounding ((2 code - cp) - cn): the difference and the mean in one launch
                results["loss_code"] = ops.masked_mean("sqdiff", (2 * code - cp)[None], cn[None])
            if tr["normal_smooth_2d"] > 0 and normals is not None and (not real_view):
                # accumulate_along_rays(weights, (normals+1)/2) with the LIVE weights (morpheus.py:775): the density
                # gradient of the normal image is part of the normal_smooth_2d loss
                _, _, _, nimg = ops.composite(sigmas, t_starts_.contiguous(), t_ends_.contiguous(), (normals + 1) / 2,
                                              ray_start, ray_cnt, padded=valid is not None)
                results["normal_image"] = nimg
            if tr["normal_smoothness"] > 0:
                # the normals are taken at the rays' own times even in a canonical render (morpheus.py:548-553 warps always)
                one_t = self.frame_batched and len(prefix) == 2 and prefix[0] == 1
                results["normal_reg"] = self.get_normal_smoothness_loss(rays_o, rays_d, rays_t, depth.reshape(1, -1),
                                                                        ray_slots=ray_slots, single_frame=one_t)
            if rays_depth is not None:
                # get_sdf_loss (utils.py:91-113, morpheus.py:789) on the packed samples: one launch each way
                fs_loss, sdf_loss = ops.sdf_losses(sdf, t_starts_, t_ends_, ri32(), rays_depth, rays_mask, tr["trunc"], n_valid)
                results["sdf_loss"], results["fs_loss"] = sdf_loss, fs_loss
        return results
